#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r05ac
mkdir -p $out
prof() {
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $out/prof_$1 -o bench -- python3 bench.py --task recon --steps 10 --warmup 3 --no-cpu-baseline --no-variants --no-roofline $2 > $out/prof_$1.log 2>&1
  python3 - <<E
import csv, glob, collections
f = glob.glob('$out/prof_$1/**/bench_kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
print('--- $1: ' + ', '.join(f"{k[:12]} {sum(v[-10:]) / len(v[-10:]) / 1e3:.0f}" for k, v in d.items() if 'k_bin3_accum' in k or 'k_bin3_emit' in k))
E
  rm -rf $out/prof_$1
}
make -s -C customnerf_amd/csrc -B -j48 TUNING=1 > $out/make_tuning.log 2>&1
prof all_fit "--prefit 300"
export CNERF_B3_ONLY=1; prof hashed_fit "--prefit 300"; prof hashed_init ""
export CNERF_B3_ONLY=2; prof dense_fit "--prefit 300"; prof dense_init ""
