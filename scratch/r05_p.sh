#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r05p
mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_field.py tests/test_gpu_train.py -q -x > $out/pytest_sel.log 2>&1; grep -E "passed|failed" $out/pytest_sel.log
timeout 300 python scratch/x2_dead.py $out/x2_dead.json 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl"
timeout 600 python -m pytest tests/test_gpu_sd_nets.py -q -s -k "sds_train_step or clip_text_encoder" 2>&1 | grep -E "\[sds|\[clip|passed|failed" > $out/tolerances.log; cat $out/tolerances.log
prof() {
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $out/prof_$1 -o bench -- python3 bench.py --task recon --steps 10 --warmup 3 --no-cpu-baseline --no-variants --no-roofline $2 > $out/prof_$1.log 2>&1
  python3 - <<E
import csv, glob, collections
f = glob.glob('$out/prof_$1/**/bench_kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
rows = sorted(((sum(v[-10:]) / len(v[-10:]) / 1e3, k, len(v)) for k, v in d.items()), reverse=True)
print('--- $1 (mean of the last 10 launches, us)')
for t, k, n in rows[:4]: print(f'   {t:8.1f}  {n:5d}  {k[:60]}')
E
  rm -rf $out/prof_$1
}
prof release ""
prof release_fit "--prefit 300"
timeout 300 python bench.py --task recon --steps 20 --warmup 5 --no-cpu-baseline > $out/bench.json 2> $out/bench.err; python3 -c "
import json; d=json.load(open('$out/bench.json')); print(d['ms_per_step'], d['value']); print({k:(v.get('ms_per_step'), v.get('value'), v.get('exchange_ms')) for k,v in d.get('variants',{}).items()})"
