#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r05dp
mkdir -p $out
timeout 700 python -m pytest tests/test_gpu_dp_two_ranks.py -q -x > $out/pytest.log 2>&1; tail -15 $out/pytest.log | cut -c1-300
