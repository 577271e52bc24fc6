#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r06p
mkdir -p $out
timeout 1800 python -m pytest tests -q -m gpu --timeout=900 > $out/pytest_gpu.log 2>&1; tail -4 $out/pytest_gpu.log
