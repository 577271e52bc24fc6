#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/bear
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bear -o b -- python3 bench.py --grid bear --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/bear/log 2>&1
grep -o '"ms_per_step": [0-9.]*' gpurun_out/bear/log
python3 - <<E
import csv
rows=list(csv.DictReader(open('gpurun_out/bear/b_kernel_stats.csv')))
for r in rows[:14]: print('%-60s calls %5s avg %8.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
E
rm -f gpurun_out/bear/b_kernel_trace.csv
