#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r05b
mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_train.py -q -x -k "graphed or scaler or adam" > $out/pytest_sel.log 2>&1; tail -3 $out/pytest_sel.log
timeout 300 scratch/fused_fwd_lab > $out/fused_fwd_lab.log 2>&1; grep -E "^----|^C |production" $out/fused_fwd_lab.log
timeout 300 python scratch/stale_plan.py > $out/stale_plan.json 2> $out/stale_plan.err; cat $out/stale_plan.json; tail -2 $out/stale_plan.err
timeout 600 python scratch/edit_aten.py > $out/edit_aten.log 2>&1; tail -90 $out/edit_aten.log
