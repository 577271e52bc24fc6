#!/bin/bash
# per-kernel averages of the recon bench under an environment setting: scratch/ab_env_all.sh VAR=val
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/abenv
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abenv -o b -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/abenv/log 2>&1
grep -o '"ms_per_step": [0-9.]*' gpurun_out/abenv/log
python3 - <<E
import csv
rows=list(csv.DictReader(open('gpurun_out/abenv/b_kernel_stats.csv')))
for r in rows[:16]: print('%-52s calls %4s avg %8.1f' % (r['Name'][:52], r['Calls'], float(r['AverageNs'])/1e3))
E
rm -rf gpurun_out/abenv
