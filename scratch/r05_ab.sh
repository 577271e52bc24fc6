#!/bin/bash
# per-kernel means of the LAST 10 launches (the fitted state, not the fitting trajectory): random init vs --prefit 300
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r05ab
mkdir -p $out
prof() {
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $out/prof_$1 -o bench -- python3 bench.py --task recon --steps 10 --warmup 3 --no-cpu-baseline --no-variants --no-roofline $2 > $out/prof_$1.log 2>&1
  python3 - <<E >> $out/early_termination_kernels.txt
import csv, glob, collections
f = glob.glob('$out/prof_$1/**/bench_kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
rows = sorted(((sum(v[-10:]) / len(v[-10:]) / 1e3, k, len(v)) for k, v in d.items()), reverse=True)
print('--- bench.py --task recon --steps 10 --warmup 3 $2 : mean of the last 10 launches of each kernel, us (launches in the whole run)')
for t, k, n in rows[:16]: print(f'   {t:8.1f}  {n:5d}  {k[:90]}')
E
  rm -rf $out/prof_$1
}
rm -f $out/early_termination_kernels.txt
prof random_init ""
prof fitted "--prefit 300"
prof fitted_1000 "--prefit 1000"
cat $out/early_termination_kernels.txt
