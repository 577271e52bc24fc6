#!/bin/bash
# A/B of an environment variable on the recon bench kernels: scratch/ab_env.sh VAR kernel_substring v1 v2 ...
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
var=$1; pat=$2; shift; shift
for v in "$@"; do
  mkdir -p gpurun_out/abenv
  env $var=$v rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abenv -o b -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/abenv/log 2>&1
  python3 - <<E
import csv, re
rows=list(csv.DictReader(open('gpurun_out/abenv/b_kernel_stats.csv')))
ms=re.findall(r'"ms_per_step": ([0-9.]+)', open('gpurun_out/abenv/log').read())
for r in rows:
    if '$pat' in r['Name']: print('$var=$v', r['Name'][:28], round(float(r['AverageNs'])/1e3,1), 'ms/step', ms)
E
  rm -rf gpurun_out/abenv
done
