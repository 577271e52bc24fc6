#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/${1:-suite}
timeout 1500 python -m pytest tests -q -m gpu --timeout=400 -p no:cacheprovider > gpurun_out/${1:-suite}/pytest_gpu.log 2>&1; grep -E "passed|failed|FAILED|ERROR" gpurun_out/${1:-suite}/pytest_gpu.log | head
