#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r06h
mkdir -p $out
timeout 1800 python -m pytest tests -q -m gpu --timeout=900 > $out/pytest_gpu.log 2>&1; tail -6 $out/pytest_gpu.log
timeout 900 python scratch/edit_ab.py 2 20 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tee $out/edit_ab.txt
bash scratch/edit_step_kernels.sh r06h > $out/edit_step_kernels.log 2>&1; head -2 $out/edit_step_kernels.log; grep -n "epilogue\|concat\|gn_" $out/edit_step_kernels.txt
