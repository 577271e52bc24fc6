#!/bin/bash
# timeline of the last edit steps: busy time, idle gaps, gap histogram by preceding kernel
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=gaps; mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$tag -o bench -- python3 bench.py --task edit --steps 6 --warmup 3 --no-cpu-baseline --no-roofline "$@" > gpurun_out/$tag/bench.log 2>&1
grep -o '"ms_per_step": [0-9.]*' gpurun_out/$tag/bench.log
python3 - <<E
import csv, collections, re
rows=list(csv.DictReader(open('gpurun_out/$tag/bench_kernel_trace.csv')))
ev=sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
# last 40% of the trace = timed steps
t_end=ev[-1][1]; n=len(ev); ev=ev[int(n*0.6):]
span=ev[-1][1]-ev[0][0]
busy=0; cur_end=ev[0][0]; gaps=collections.defaultdict(lambda:[0,0.0]); big=[]
for i,(s,e,k) in enumerate(ev):
    if s>cur_end:
        g=s-cur_end
        pk=re.sub(r'\(.*','',ev[i-1][2]).replace('void ','')[:40]; nk=re.sub(r'\(.*','',k).replace('void ','')[:40]
        gaps[(pk,nk)][0]+=1; gaps[(pk,nk)][1]+=g
        if g>20000: big.append((g,pk,nk))
    busy+=max(0,e-max(s,cur_end)); cur_end=max(cur_end,e)
print('kernels',len(ev),'span ms',span/1e6,'busy ms',busy/1e6,'idle ms',(span-busy)/1e6)
tot=sum(v[1] for v in gaps.values()); cnt=sum(v[0] for v in gaps.values())
print('gaps',cnt,'mean us',tot/cnt/1e3)
for (pk,nk),v in sorted(gaps.items(), key=lambda kv:-kv[1][1])[:25]: print('%-40s -> %-40s n %5d tot %8.1f us mean %6.1f' % (pk,nk,v[0],v[1]/1e3,v[1]/v[0]/1e3))
print('big gaps', len(big), sum(b[0] for b in big)/1e6, 'ms')
for b in sorted(big, reverse=True)[:15]: print(b)
E
rm -f gpurun_out/$tag/bench_kernel_trace.csv
