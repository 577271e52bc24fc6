#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_dp_two_ranks.py -q -x --timeout=600 2>&1 | tail -15 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl"
