#!/bin/bash
# kernel trace of the recon bench: scratch/prof.sh <tag> [bench args]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -o bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > gpurun_out/$tag/bench.log 2>&1
grep -o '"ms_per_step": [0-9.]*' gpurun_out/$tag/bench.log
python3 - <<E
import csv
rows=list(csv.DictReader(open('gpurun_out/$tag/bench_kernel_stats.csv')))
for r in rows[:16]: print('%-62s %5s %10.1f %6s' % (r['Name'][:62], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
E
rm -f gpurun_out/$tag/bench_kernel_trace.csv
