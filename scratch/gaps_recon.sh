#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=gapsr; mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$tag -o bench -- python3 bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-roofline > gpurun_out/$tag/bench.log 2>&1
grep -o '"ms_per_step": [0-9.]*' gpurun_out/$tag/bench.log
python3 - <<E
import csv, re
rows=list(csv.DictReader(open('gpurun_out/$tag/bench_kernel_trace.csv')))
ev=sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Stream_Id','')) for r in rows)
def short(k): return re.sub(r'\(.*','',k).replace('void ','')[:30]
idx=[i for i,e in enumerate(ev) if 'adam' in e[2].lower()]
bounds=[idx[0]]
for a,b in zip(idx,idx[1:]):
    if b-a>10: bounds.append(b)
s0,s1=bounds[-3],bounds[-2]
seg=ev[s0:s1]; t0=seg[0][0]
print('kernels in step', len(seg), 'span us', (seg[-1][1]-t0)/1e3)
cur=seg[0][1]; idle=0
for i,(s,e,k,st) in enumerate(seg[1:],1):
    g=s-cur
    if g>0: idle+=g
    if g>3000: print('%8.1f us gap %5.1f us  %-30s -> %-30s' % ((cur-t0)/1e3, g/1e3, short(seg[i-1][2]), short(k)))
    cur=max(cur,e)
print('idle (no kernel running) us', idle/1e3)
for (s,e,k,st) in seg: print('%8.1f %8.1f %s %s' % ((s-t0)/1e3, (e-s)/1e3, st, short(k)))
E
rm -f gpurun_out/$tag/bench_kernel_trace.csv
