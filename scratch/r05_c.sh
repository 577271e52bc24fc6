#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r05c
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_gridencoder.py tests/test_gpu_fullsize.py tests/test_gpu_train.py tests/test_gpu_render.py -q -x > $out/pytest_sel.log 2>&1; tail -5 $out/pytest_sel.log
timeout 600 python bench.py --task recon --steps 20 --warmup 5 --no-cpu-baseline > $out/bench.json 2> $out/bench.err; python3 -c "
import json; d=json.load(open('$out/bench.json')); print(d['ms_per_step'], d['value']); print({k:(v.get('ms_per_step'), v.get('value')) for k,v in d.get('variants',{}).items()})"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_recon -o bench -- python3 bench.py --task recon --steps 20 --warmup 5 --no-cpu-baseline --no-variants > $out/prof_recon.log 2>&1
rm -f $out/prof_recon/*/bench_kernel_trace.csv $out/prof_recon/bench_kernel_trace.csv
python3 - <<E
import csv, glob
f = glob.glob('$out/prof_recon/**/bench_kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:22]:
    print(f"{int(r['Calls']):5d} {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:90]}")
E
