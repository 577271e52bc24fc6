#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r06f
mkdir -p $out
timeout 1800 python -m pytest tests/test_gpu_uninitialised.py tests/test_gpu_gridencoder.py -q --timeout=900 > $out/pytest_sel.log 2>&1; tail -12 $out/pytest_sel.log
