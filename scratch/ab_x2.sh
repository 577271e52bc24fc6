#!/bin/bash
# time of k_field_bwd_x2 under CNERF_X2_ABLATE masks
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for m in "$@"; do
  mkdir -p gpurun_out/abx4_$m
  CNERF_X2_ABLATE=$m rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abx4_$m -o b -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
  python3 - <<E
import csv
rows=list(csv.DictReader(open('gpurun_out/abx4_$m/b_kernel_stats.csv')))
for r in rows:
    if 'k_field_bwd' in r['Name']: print('ablate=$m', r['Name'][:30], float(r['AverageNs'])/1e3)
E
  rm -rf gpurun_out/abx4_$m
done
