#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for w in 0 1; do
  mkdir -p gpurun_out/l2w
  WARM=$w rocprofv3 --kernel-trace --output-format csv -d gpurun_out/l2w -o t -- python3 scratch/l2warm_test.py > gpurun_out/l2w/log 2>&1
  python3 - <<E
import csv
rows=[r for r in csv.DictReader(open('gpurun_out/l2w/t_kernel_trace.csv')) if 'k_grid_fwd_fast' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
print('WARM=$w gather durations (us):', [round(x) for x in d[-12:]])
E
  rm -rf gpurun_out/l2w
done
