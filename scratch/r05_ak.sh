#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_train.py -q -x -k "trajectory or reproducible" 2>&1 | tail -5 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl"
