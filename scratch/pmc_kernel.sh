#!/bin/bash
# scratch/pmc_kernel.sh <kernel substring> <counter> [<counter> ...] : one rocprofv3 --pmc pass per counter on the recon bench, average per launch
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
k=$1; shift
for c in "$@"; do
  rm -rf gpurun_out/pk_$c; mkdir -p gpurun_out/pk_$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pk_$c -o b -- python3 bench.py --task recon --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/pk_$c/log 2>&1
  python3 - <<E
import csv
try:
    rows = [r for r in csv.DictReader(open('gpurun_out/pk_$c/b_counter_collection.csv')) if '$k' in r['Kernel_Name']]
    v = [float(r['Counter_Value']) for r in rows]
    print('$c', round(sum(v) / len(v), 1), 'launches', len(v))
except Exception as e:
    print('$c', 'no data', e)
E
  rm -rf gpurun_out/pk_$c
done
