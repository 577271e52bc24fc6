#!/bin/bash
# per-step kernel table of the edit leg: scratch/edit_breakdown.sh <tag> [bench args]   (window = 9 step periods between k_sds_grad launches)
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
win = rows[a:b]
n = 9
acc = collections.defaultdict(lambda: [0, 0.0])
for r in win:
    k = r['Kernel_Name']
    acc[k][0] += 1; acc[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
span = (int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3 / n
tot = sum(v[1] for v in acc.values()) / n
print('launches/step %.1f  kernel us/step %.1f  span us/step %.1f' % (len(win) / n, tot, span))
with open('$out/edit_per_step.txt', 'w') as fh:
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        line = '%-100s %7.1f %9.1f %8.1f' % (k[:100], v[0] / n, v[1] / n, v[1] / v[0])
        fh.write(line + '\n')
for l in open('$out/edit_per_step.txt').read().splitlines()[:45]: print(l)
E
rm -f $out/e_kernel_trace.csv
