#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r05f
mkdir -p $out
prof() { # name, extra env / args
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$1 -o bench -- python3 bench.py --task recon --steps 20 --warmup 5 --no-cpu-baseline --no-variants $2 > $out/prof_$1.log 2>&1
  rm -f $out/prof_$1/*/bench_kernel_trace.csv $out/prof_$1/bench_kernel_trace.csv
  python3 - <<E
import csv, glob
f = glob.glob('$out/prof_$1/**/bench_kernel_stats.csv', recursive=True)[0]
print('--- $1')
for r in list(csv.DictReader(open(f)))[:$3]:
    print(f"{int(r['Calls']):5d} {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:90]}")
E
}
timeout 900 python -m pytest tests/test_gpu_gridencoder.py tests/test_gpu_fullsize.py -q > $out/pytest_sel.log 2>&1; tail -3 $out/pytest_sel.log
timeout 600 python bench.py --task recon --steps 20 --warmup 5 --no-cpu-baseline > $out/bench.json 2> $out/bench.err; python3 -c "
import json; d=json.load(open('$out/bench.json')); print(d['ms_per_step'], d['value']); print({k:(v.get('ms_per_step'), v.get('value'), v.get('exchange_ms')) for k,v in d.get('variants',{}).items()})"
prof recon "" 14
prof fit "--prefit 300" 8
make -s -C customnerf_amd/csrc -B -j48 TUNING=1 > $out/make_tuning.log 2>&1
export CNERF_B3_ONLY=1; prof hashed_only "" 4
export CNERF_B3_ONLY=2; prof dense_only "" 4
unset CNERF_B3_ONLY
export CNERF_B3=0; prof v2 "" 12
