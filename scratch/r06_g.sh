#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r06g
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_uninitialised.py -q --timeout=900 > $out/pytest_sel.log 2>&1; tail -5 $out/pytest_sel.log
timeout 1500 python scratch/edit_ab.py 3 20 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tee $out/edit_ab.txt
