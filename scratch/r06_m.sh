#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r06m
mkdir -p $out
timeout 1800 python -m pytest tests/test_gpu_train.py tests/test_gpu_gridencoder.py tests/test_gpu_fullsize.py tests/test_gpu_sd_editing.py tests/test_gpu_uninitialised.py tests/test_gpu_render.py tests/test_gpu_field.py -q --timeout=900 > $out/pytest_sel.log 2>&1; tail -6 $out/pytest_sel.log
timeout 300 python bench.py --task recon --steps 20 --warmup 5 --no-cpu-baseline --stage-events > $out/bench_recon.json 2> $out/bench_recon.err; python3 -c "
import json; d=json.load(open('$out/bench_recon.json')); print('recon', d['ms_per_step'], d['roofline']['frac'], d['config'].get('stage_ms')); print({k:(round(v.get('ms_per_step',0),3)) for k,v in d.get('variants',{}).items()})"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python3 bench.py --task recon --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-roofline > $out/prof.log 2>&1
f=$(find $out/prof -name bench_kernel_stats.csv | head -1); python3 - <<P
import csv
for r in list(csv.DictReader(open('$f')))[:22]: print('%-60s %5s %9.1f' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
P
rm -rf $out/prof
