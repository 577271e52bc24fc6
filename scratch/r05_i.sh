#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r05j
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_field.py tests/test_gpu_train.py -q -x > $out/pytest_sel.log 2>&1; tail -3 $out/pytest_sel.log
timeout 300 python bench.py --task recon --steps 20 --warmup 5 --no-cpu-baseline > $out/bench.json 2> $out/bench.err; python3 -c "
import json; d=json.load(open('$out/bench.json')); print(d['ms_per_step'], d['value']); print({k:(v.get('ms_per_step'), v.get('value'), v.get('exchange_ms')) for k,v in d.get('variants',{}).items()})"
prof() {
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$1 -o bench -- python3 bench.py --task recon --steps 10 --warmup 3 --no-cpu-baseline --no-variants --no-roofline $2 > $out/prof_$1.log 2>&1
  rm -f $out/prof_$1/*/bench_kernel_trace.csv $out/prof_$1/bench_kernel_trace.csv
  python3 - <<E
import csv, glob
f = glob.glob('$out/prof_$1/**/bench_kernel_stats.csv', recursive=True)[0]
print('--- $1: ' + ', '.join(f"{r['Name'][:14]} {float(r['AverageNs'])/1e3:.0f}" for r in csv.DictReader(open(f)) if 'k_bin' in r['Name'] or 'field_bwd' in r['Name']))
E
}
prof release ""
prof release_fit "--prefit 300"
