#!/bin/bash
# PMC passes on any python script: scratch/pmc_script.sh <script.py> <kernel substring> "C1 C2 ..." ["..."]  -> per-kernel-name averages
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
script=$1; k=$2; shift; shift
i=0
for set in "$@"; do
  i=$((i+1))
  rm -rf gpurun_out/pmcs_$i
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmcs_$i -o b -- python3 $script > gpurun_out/pmcs_$i.log 2>&1
  python3 - <<E
import csv, collections
rows = list(csv.DictReader(open('gpurun_out/pmcs_$i/b_counter_collection.csv')))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if '$k' in r['Kernel_Name']:
        acc[r['Kernel_Name'][:52] + ' grid=' + r.get('Grid_Size', '?')][r['Counter_Name']].append(float(r['Counter_Value']))
for kk, v in acc.items():
    print(kk, {c: round(sum(x) / len(x), 1) for c, x in v.items()})
E
  rm -rf gpurun_out/pmcs_$i
done
