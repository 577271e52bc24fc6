#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r05ae
mkdir -p $out
prof() {
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $out/prof_$1 -o bench -- python3 bench.py --task recon --steps 10 --warmup 3 --no-cpu-baseline --no-variants --no-roofline $2 > $out/prof_$1.log 2>&1
  python3 - <<E
import csv, glob, collections
f = glob.glob('$out/prof_$1/**/bench_kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
print('--- $1: ' + ', '.join(f"{k[:12]} {sum(v[-10:]) / len(v[-10:]) / 1e3:.0f}" for k, v in d.items() if 'k_bin3_accum' in k))
E
  rm -rf $out/prof_$1
}
timeout 600 python -m pytest tests/test_gpu_gridencoder.py tests/test_gpu_fullsize.py -q -x 2>&1 | grep -E "passed|failed"
prof c16_init ""; prof c16_fit "--prefit 300"
for cfg in "32 20000" "16 100000" "8 20000"; do
  set -- $cfg
  make -s -C customnerf_amd/csrc -B -j48 EXTRA="-DB3_WALK_COARSE=$1 -DB3_WALK_COARSE_ENTRIES=$2u" > $out/make.log 2>&1
  prof c$1_e$2_init ""; prof c$1_e$2_fit "--prefit 300"
done
