#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r06j
mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_gridencoder.py tests/test_gpu_fullsize.py tests/test_gpu_uninitialised.py tests/test_gpu_render.py tests/test_gpu_train.py -q --timeout=900 -x > $out/pytest_sel.log 2>&1; tail -8 $out/pytest_sel.log
timeout 300 python bench.py --task recon --grid bear --steps 20 --warmup 5 --no-cpu-baseline --no-variants --stage-events > $out/bench_bear.json 2> $out/bench_bear.err; python3 -c "
import json; d=json.load(open('$out/bench_bear.json')); print('bear', d['ms_per_step'], d['value'], d['roofline'].get('coarse'), d['roofline'].get('fine'), d['config'].get('stage_ms'))"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_bear -o bench -- python3 bench.py --task recon --grid bear --steps 20 --warmup 5 --no-cpu-baseline --no-variants > $out/prof_bear.log 2>&1
rm -f $out/prof_bear/bench_kernel_trace.csv; f=$(find $out/prof_bear -name bench_kernel_stats.csv | head -1); cp $f $out/bear_kernel_stats.csv; rm -rf $out/prof_bear
python3 - <<P
import csv
rows=list(csv.DictReader(open('$out/bear_kernel_stats.csv')))
for r in rows[:22]: print('%-60s %5s %9.1f' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
P
make -s -C customnerf_amd/csrc -B -j64 TUNING=1 > $out/make_tuning.log 2>&1; tail -2 $out/make_tuning.log
bash scratch/ab_recon.sh r06j/ab_bear "--grid bear" "CNERF_B3_WIDE=0" "-" "CNERF_B3_WIDE=0" "-" | tee $out/ab_bear.txt
bash scratch/ab_recon.sh r06j/ab_bear_fit "--grid bear --prefit 300" "CNERF_B3_WIDE=0" "-" | tee $out/ab_bear_fit.txt
