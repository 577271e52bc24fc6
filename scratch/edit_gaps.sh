#!/bin/bash
# where the idle time of the edit leg sits: scratch/edit_gaps.sh <tag>  -> per preceding-kernel name, the gap (start of next - end of this) summed over 9 step periods
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -o e -- python3 bench.py --task edit --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-variants "$@" > $out/bench.log 2>&1
grep -o '"ms_per_step": [0-9.]*' $out/bench.log | head -1
python3 - <<E
import csv, collections
rows = list(csv.DictReader(open('$out/e_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'k_sds_grad' in r['Kernel_Name']]
a, b = marks[-10], marks[-1]
win = rows[a:b + 1]
gap = collections.defaultdict(lambda: [0, 0.0])
end = int(win[0]['End_Timestamp'])
for prev, cur in zip(win, win[1:]):
    g = int(cur['Start_Timestamp']) - end
    if g > 0:
        k = prev['Kernel_Name'][:70] + ' -> ' + cur['Kernel_Name'][:40]
        gap[k][0] += 1; gap[k][1] += g / 1e3
    end = max(end, int(cur['End_Timestamp']))
tot = sum(v[1] for v in gap.values()) / 9
print('idle us/step %.1f' % tot)
for k, v in sorted(gap.items(), key=lambda kv: -kv[1][1])[:14]:
    print('%-115s %6.1f /step %8.1f us/step %7.1f us each' % (k, v[0] / 9, v[1] / 9, v[1] / v[0]))
E
rm -f $out/e_kernel_trace.csv
