#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=gaps; mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$tag -o bench -- python3 bench.py --task edit --steps 6 --warmup 3 --no-cpu-baseline --no-roofline "$@" > gpurun_out/$tag/bench.log 2>&1
grep -o '"ms_per_step": [0-9.]*' gpurun_out/$tag/bench.log
python3 - <<E
import csv, re
rows=list(csv.DictReader(open('gpurun_out/$tag/bench_kernel_trace.csv')))
ev=sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
def short(k): return re.sub(r'\(.*','',k).replace('void ','')[:34]
# find step boundaries: the Adam kernel
idx=[i for i,e in enumerate(ev) if 'adam' in e[2].lower()]
print('adam launches', len(idx))
# take the second-to-last step
names=[short(e[2]) for e in ev]
ad=[i for i in idx]
# group consecutive adam launches
bounds=[ad[0]]
for a,b in zip(ad,ad[1:]):
    if b-a>50: bounds.append(b)
s0,s1=bounds[-3],bounds[-2]
seg=ev[s0:s1]
t0=seg[0][0]
print('step kernels',len(seg),'span us',(seg[-1][1]-t0)/1e3)
cur=seg[0][1]; run_busy=0; last_print=0
for i,(s,e,k) in enumerate(seg[1:],1):
    g=s-cur
    if g>8000:
        print('%9.1f us  gap %6.1f us  after %-34s before %-34s (idx %d)' % ((cur-t0)/1e3, g/1e3, short(seg[i-1][2]), short(k), i))
    cur=max(cur,e)
E
rm -f gpurun_out/$tag/bench_kernel_trace.csv
