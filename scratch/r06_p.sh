#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r06p
mkdir -p $out
timeout 1800 python -m pytest tests/test_gpu_gridencoder.py tests/test_gpu_fullsize.py tests/test_gpu_render.py tests/test_gpu_train.py tests/test_gpu_uninitialised.py -q --timeout=900 > $out/pytest_sel.log 2>&1; grep -E "passed|failed|FAILED" $out/pytest_sel.log | tail -6
