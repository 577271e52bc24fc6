#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06q
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_field.py tests/test_gpu_gridencoder.py tests/test_gpu_dp_two_ranks.py tests/test_gpu_sd_editing.py -q -x --timeout=600 > gpurun_out/r06q/pytest.log 2>&1; grep -E "passed|failed|error" gpurun_out/r06q/pytest.log | tail -3
