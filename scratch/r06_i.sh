#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r06i
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_train.py -q --timeout=600 > $out/pytest_sel.log 2>&1; tail -3 $out/pytest_sel.log
timeout 300 python bench.py --task recon --steps 20 --warmup 5 --no-cpu-baseline --no-variants --stage-events > $out/bench_recon.json 2> $out/bench_recon.err; python3 -c "
import json; d=json.load(open('$out/bench_recon.json')); print('recon', d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('fine_traversal'), d['config'].get('stage_ms'))"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python3 bench.py --task recon --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-roofline > $out/prof.log 2>&1
f=$(find $out/prof -name bench_kernel_stats.csv | head -1); python3 - <<P
import csv
for r in list(csv.DictReader(open('$f')))[:24]: print('%-60s %5s %9.1f' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
P
rm -rf $out/prof
make -s -C customnerf_amd/csrc -B -j64 TUNING=1 > $out/make_tuning.log 2>&1; tail -2 $out/make_tuning.log
bash scratch/ab_recon.sh r06i/ab_spt "--no-tune-traversal" "-" "CNERF_GRID_TRAV=1 CNERF_GRID_SPT=12" "CNERF_GRID_TRAV=1 CNERF_GRID_SPT=16" "CNERF_GRID_TRAV=1 CNERF_GRID_SPT=20" "CNERF_GRID_TRAV=1 CNERF_GRID_SPT=24" "CNERF_GRID_DENSE_W=1" "CNERF_GRID_DENSE_W=0" | tee $out/ab_spt.txt
