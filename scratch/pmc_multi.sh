#!/bin/bash
# several counters per pass (<= 8 SQ / 4 TCC slots): scratch/pmc_multi.sh <tag> "C1 C2 ..." ["C9 ..."] -> per-kernel averages
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out/$tag
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/$tag/p$i -o b -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline $BENCH_ARGS > gpurun_out/$tag/p$i.log 2>&1
  python3 - <<E
import csv, collections
try:
    rows = list(csv.DictReader(open('gpurun_out/$tag/p$i/b_counter_collection.csv')))
except Exception as e:
    print('$set', 'no data', e); raise SystemExit
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    acc[r['Kernel_Name'][:44]][r['Counter_Name']].append(float(r['Counter_Value']))
sel = ("k_bin2", "k_grid_fwd", "k_field_bwd", "k_field_fwd", "k_sd_gemm", "k_sd_attention", "k_composite", "k_adam")
for k in acc:
    if any(s in k for s in sel):
        print(k, {c: round(sum(v) / len(v), 1) for c, v in acc[k].items()}, 'n=%d' % len(next(iter(acc[k].values()))))
E
  rm -f gpurun_out/$tag/p$i/b_kernel_trace.csv
done
