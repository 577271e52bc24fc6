#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/lvl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lvl -o t -- python3 scratch/level_cost.py > gpurun_out/lvl/out.log 2>&1
cat gpurun_out/lvl/out.log | tail -8
python3 - <<'E'
import csv, collections
rows = list(csv.DictReader(open('gpurun_out/lvl/t_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = ['k_bin2_hist', 'k_bin_scan_blocks', 'k_bin2_emit', 'k_bin2_accum']
per = collections.defaultdict(list)
cnt = collections.Counter()
for r in rows:
    n = r['Kernel_Name']
    for k in names:
        if k in n:
            per[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
labels = [f"{m} ml={ml}" for m in ("ray", "shuf") for ml in (5, 16, 11)]
for k in names:
    v = per[k]
    g = len(v) // 6
    print(k, {labels[i]: round(sum(v[i * g + 3:(i + 1) * g]) / max(1, g - 3), 1) for i in range(6)})
E
rm -f gpurun_out/lvl/t_kernel_trace.csv
