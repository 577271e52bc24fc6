#!/bin/bash
# round-6 first GPU call: changed-code tests, bench (with small_batch), batch sweep, tuning-build A/B of the gather / scatter variants
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r06a
mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_gridencoder.py tests/test_gpu_train.py tests/test_gpu_render.py tests/test_gpu_field.py -q -x -s --timeout=600 > $out/pytest_sel.log 2>&1; tail -4 $out/pytest_sel.log; grep "headline parity" $out/pytest_sel.log
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; python3 -c "
import json; d=json.load(open('$out/bench.json')); print('recon', d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('fine_traversal')); print({k:(round(v.get('ms_per_step',0),3), v.get('stage_ms'), v.get('implied_strong_scaling_bound_8gpu'), (v.get('roofline') or {}).get('fine_traversal')) for k,v in d.get('variants',{}).items()}); s=d['secondary']; print('edit', s['ms_per_step'])"
bash scratch/batch_sweep.sh r06a/sweep
make -s -C customnerf_amd/csrc -B -j64 TUNING=1 > $out/make_tuning.log 2>&1; tail -2 $out/make_tuning.log
bash scratch/ab_recon.sh r06a/ab_init "--no-tune-traversal" "-" "CNERF_GRID_K=2" "CNERF_GRID_K=4" "CNERF_GRID_K=8" "CNERF_GRID_TRAV=1 CNERF_GRID_SPT=4" "CNERF_GRID_TRAV=1 CNERF_GRID_SPT=8" "CNERF_GRID_TRAV=1 CNERF_GRID_SPT=16" "CNERF_B3_WALK_MODE=1" | tee $out/ab_init.txt
bash scratch/ab_recon.sh r06a/ab_fit "--no-tune-traversal --prefit 300" "-" "CNERF_GRID_K=4" "CNERF_GRID_TRAV=1 CNERF_GRID_SPT=4" "CNERF_GRID_TRAV=1 CNERF_GRID_SPT=8" "CNERF_GRID_TRAV=1 CNERF_GRID_SPT=16" "CNERF_B3_WALK_MODE=1" | tee $out/ab_fit.txt
bash scratch/ab_recon.sh r06a/ab_tuner "--prefit 300" "-" | tee $out/ab_tuner.txt
