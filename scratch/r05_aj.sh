#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05aj
timeout 600 python scratch/soak_early_term.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tee gpurun_out/r05aj/soak.log
