#!/bin/bash
# PMC passes on the standalone field-backward timing script: scratch/pmc_w8.sh "C1 C2 ..." ["..."]  (release library)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  rm -rf gpurun_out/pmcw8_$i
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmcw8_$i -o b -- python3 scratch/field_bwd_timing.py > gpurun_out/pmcw8_$i.log 2>&1
  python3 - <<E
import csv, collections
rows = list(csv.DictReader(open('gpurun_out/pmcw8_$i/b_counter_collection.csv')))
acc = collections.defaultdict(list)
for r in rows:
    if 'k_field_bwd' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
print({c: round(sum(v) / len(v), 1) for c, v in acc.items()})
E
  rm -rf gpurun_out/pmcw8_$i
done
