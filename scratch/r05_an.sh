#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r05an
mkdir -p $out
prof() {
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/prof_$1 -o bench -- python3 bench.py --task edit --steps 10 --warmup 3 --no-cpu-baseline --no-roofline $2 > $out/prof_$1.log 2>&1
  python3 - <<E
import csv, glob, collections
f = glob.glob('$out/prof_$1/**/bench_kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
print('--- $1: ' + ', '.join(f"{k[:16]} {sum(v[-10:]) / len(v[-10:]) / 1e3:.0f}" for k, v in d.items() if 'k_bin3_accum' in k or 'k_bin3_emit' in k or 'k_field_bwd' in k or 'k_composite_run_bwd' in k))
E
  rm -rf $out/prof_$1
}
prof init ""
prof fitted "--edit-prefit 300"
