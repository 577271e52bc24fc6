#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05n
timeout 300 python scratch/x2_dead.py gpurun_out/r05n/x2_dead.json 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl"
