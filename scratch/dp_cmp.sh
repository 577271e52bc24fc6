#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for m in plain sharded; do
  sed "s/for mode in (\"plain\", \"sharded\"):/for mode in (\"$m\",):/" scratch/edit_dp_world1.py > scratch/_edw_$m.py
  mkdir -p gpurun_out/dpcmp_$m
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dpcmp_$m -o e -- python3 scratch/_edw_$m.py > gpurun_out/dpcmp_$m/log 2>&1
  rm -f gpurun_out/dpcmp_$m/e_kernel_trace.csv scratch/_edw_$m.py
done
python3 - <<'E'
import csv
def load(m):
    return {r['Name']: (int(r['Calls']), float(r['TotalDurationNs'])) for r in csv.DictReader(open(f'gpurun_out/dpcmp_{m}/e_kernel_stats.csv'))}
a, b = load('plain'), load('sharded')
steps = 38
rows = []
for k in set(a) | set(b):
    ca, ta = a.get(k, (0, 0.0)); cb, tb = b.get(k, (0, 0.0))
    rows.append(((tb - ta) / steps / 1e3, k[:80], ca, cb, ta / steps / 1e3, tb / steps / 1e3))
rows.sort(reverse=True)
print("per-step kernel time, sharded - plain (us):")
for r in rows[:14]: print(f"{r[0]:8.1f}  {r[1]}  calls {r[2]}/{r[3]}  {r[4]:.1f} -> {r[5]:.1f}")
print("...")
for r in rows[-6:]: print(f"{r[0]:8.1f}  {r[1]}  calls {r[2]}/{r[3]}  {r[4]:.1f} -> {r[5]:.1f}")
print("sum", sum(r[0] for r in rows))
E
