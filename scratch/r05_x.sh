#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r05x
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_sd_ops.py tests/test_gpu_sd_nets.py tests/test_gpu_sd_editing.py -q -x --timeout=400 > $out/pytest_sel.log 2>&1; grep -E "passed|failed" $out/pytest_sel.log
bash scratch/edit_step_kernels.sh r05x > $out/esk.log 2>&1; head -2 $out/esk.log; grep -E "k_sd_gemm<.*true, true|splitk" $out/edit_step_kernels.txt | head
timeout 300 python scratch/edit_host.py 2>&1 | grep -E "wall|isolated"
