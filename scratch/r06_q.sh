#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r06q
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_field.py tests/test_gpu_train.py -q -x --timeout=600 > $out/pytest_sel.log 2>&1; tail -5 $out/pytest_sel.log
for rep in 1 2; do
for v in "" "--no-packed-weights"; do
for r in 16384 2048; do
timeout 200 python3 bench.py --task recon --rays $r --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-roofline $v > $out/b.json 2> $out/b.err
python3 -c "
import json; d=json.load(open('$out/b.json')); print('rays $r [$v] step %.4f ms' % d['ms_per_step'])"
done; done; done 2>&1 | tee $out/ab_packed.txt
for v in "" "--no-packed-weights"; do
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python3 bench.py --task recon --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-roofline $v > $out/prof.log 2>&1
f=$(find $out/prof -name bench_kernel_stats.csv | head -1); echo "== [$v]"; python3 - <<P
import csv
for r in list(csv.DictReader(open('$f')))[:30]:
    if 'field' in r['Name']: print('%-60s %5s %9.1f' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
P
rm -rf $out/prof
done 2>&1 | tee $out/ab_packed_kernels.txt
