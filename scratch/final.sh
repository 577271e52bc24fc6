#!/bin/bash
# end-of-session measurement set: scratch/final.sh <tag>  (every step under its own timeout)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=$1
out=gpurun_out/$tag
mkdir -p $out
timeout 600 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1; tail -2 $out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
timeout 600 python bench.py > $out/bench_recon.json 2> $out/bench_recon.err; tail -c 600 $out/bench_recon.json; echo
timeout 900 python bench.py --task edit > $out/bench_edit.json 2> $out/bench_edit.err; tail -c 400 $out/bench_edit.json; echo
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_recon -o bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/prof_recon.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_edit -o bench -- python3 bench.py --task edit --steps 10 --warmup 3 --no-cpu-baseline > $out/prof_edit.log 2>&1
rm -f $out/prof_recon/bench_kernel_trace.csv $out/prof_edit/bench_kernel_trace.csv
for c in FETCH_SIZE WRITE_SIZE TCP_TCC_READ_REQ_sum; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -o b -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $out/pmc_$c.log 2>&1
  python3 - <<E
import csv, collections
try:
    rows = list(csv.DictReader(open('$out/pmc_$c/b_counter_collection.csv')))
    acc = collections.defaultdict(list)
    for r in rows: acc[r['Kernel_Name'][:34]].append(float(r['Counter_Value']))
    print('$c', {k: round(sum(v) / len(v), 1) for k, v in acc.items() if 'k_grid_fwd' in k or 'k_bin2' in k})
except Exception as e:
    print('$c', 'no data', e)
E
  rm -f $out/pmc_$c/b_kernel_trace.csv
done
