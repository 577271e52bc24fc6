#!/bin/bash
# per-kernel PMC averages of the recon bench, one counter per pass: scratch/pmc.sh <tag> <counter>...
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out/$tag
for c in "$@"; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/$tag/$c -o b -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/$c.log 2>&1
  python3 - <<E
import csv, collections
try:
    rows = list(csv.DictReader(open('gpurun_out/$tag/$c/b_counter_collection.csv')))
except Exception as e:
    print('$c', 'no data', e); raise SystemExit
acc = collections.defaultdict(list)
for r in rows:
    acc[r['Kernel_Name'][:40]].append(float(r['Counter_Value']))
keys = [k for k in acc if any(s in k for s in ("k_bin2", "k_grid_fwd", "k_field_bwd_mma", "k_field_fwd"))]
print('$c', {k: round(sum(acc[k]) / len(acc[k]), 3) for k in keys})
E
  rm -f gpurun_out/$tag/$c/b_kernel_trace.csv
done
