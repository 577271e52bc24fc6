#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r06q
mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_gridencoder.py tests/test_gpu_fullsize.py -q -x --timeout=600 > $out/pytest.log 2>&1; grep -E "passed|failed|error" $out/pytest.log | tail -3; grep -E "^FAILED|^ERROR" $out/pytest.log | head
bash scratch/ab_recon.sh r06q/ab "" "-" "-" 2>&1
bash scratch/ab_recon.sh r06q/ab "--prefit 300" "-" 2>&1
bash scratch/ab_recon.sh r06q/ab "--grid bear" "-" 2>&1
