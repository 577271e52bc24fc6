#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -o bench -- python3 bench.py --task edit --steps 10 --warmup 3 --no-cpu-baseline --no-roofline "$@" > gpurun_out/$tag/bench.log 2>&1
grep -o '"ms_per_step": [0-9.]*' gpurun_out/$tag/bench.log
python3 - <<E
import csv, collections, re
rows=list(csv.DictReader(open('gpurun_out/$tag/bench_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
grp=collections.defaultdict(float)
for r in rows:
    n=r['Name']
    k=re.sub(r'\(.*','',n).replace('void ','')
    k=re.sub(r'<.*','',k)
    if k.startswith('_Z'): k=re.sub(r'^_Z\d+','',k)[:16]
    grp[k]+=float(r['TotalDurationNs'])
print('total kernel ms', tot/1e6)
for k,v in sorted(grp.items(), key=lambda kv:-kv[1])[:24]: print('%-40s %8.2f ms %5.1f%%' % (k[:40], v/1e6, 100*v/tot))
E
rm -f gpurun_out/$tag/bench_kernel_trace.csv
