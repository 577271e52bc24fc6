#!/bin/bash
# where the idle time of the recon leg sits: scratch/recon_gaps.sh <tag> [bench args] -> per boundary (kernel -> next kernel) the gap, averaged over 9 step periods
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -o e -- python3 bench.py --task recon --steps 12 --warmup 5 --no-cpu-baseline --no-roofline --no-variants "$@" > $out/bench.log 2>&1
grep -o '"ms_per_step": [0-9.]*' $out/bench.log | head -1
python3 - <<P
import csv, collections
rows = list(csv.DictReader(open('$out/e_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'k_adam_scaled_multi' in r['Kernel_Name']]
a, b = marks[-10], marks[-1]
win = rows[a:b + 1]
gap = collections.defaultdict(lambda: [0, 0.0])
busy = 0.0
end = int(win[0]['End_Timestamp'])
for prev, cur in zip(win, win[1:]):
    g = int(cur['Start_Timestamp']) - end
    busy += (int(cur['End_Timestamp']) - int(cur['Start_Timestamp'])) / 1e3
    k = prev['Kernel_Name'].split('(')[0][-34:] + ' -> ' + cur['Kernel_Name'].split('(')[0][-34:]
    gap[k][0] += 1; gap[k][1] += g / 1e3
    end = max(end, int(cur['End_Timestamp']))
span = (int(win[-1]['End_Timestamp']) - int(win[0]['End_Timestamp'])) / 1e3 / 9
print('step period %.1f us, kernel time %.1f us, idle %.1f us/step over %d boundaries' % (span, busy / 9, sum(v[1] for v in gap.values()) / 9, len(win) // 9))
for k, v in sorted(gap.items(), key=lambda kv: -kv[1][1]):
    print('%-76s %5.1f /step %7.1f us/step %6.1f us each' % (k, v[0] / 9, v[1] / 9, v[1] / v[0]))
P
rm -f $out/e_kernel_trace.csv
