#!/bin/bash
# round-5 first GPU call: changed-code tests, baseline bench, the three "measure first" experiments
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r05a
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_sd_ops.py tests/test_gpu_gridencoder.py tests/test_gpu_field.py -q -x > $out/pytest_sel.log 2>&1; tail -3 $out/pytest_sel.log
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; tail -c 400 $out/bench.json; echo
timeout 300 scratch/fused_fwd_lab > $out/fused_fwd_lab.log 2>&1; cat $out/fused_fwd_lab.log
timeout 600 python scratch/zero_rows.py > $out/zero_rows.json 2> $out/zero_rows.err; grep -E "state|mean_zero" $out/zero_rows.json; tail -2 $out/zero_rows.err
timeout 300 python scratch/stale_plan.py > $out/stale_plan.json 2> $out/stale_plan.err; cat $out/stale_plan.json; tail -2 $out/stale_plan.err
