#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_gridencoder.py -q -x -k "fewer_active" 2>&1 | tail -12 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl"
