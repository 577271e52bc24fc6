#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r05v
mkdir -p $out
prof() {
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$1 -o bench -- python3 bench.py --task recon --steps 10 --warmup 3 --no-cpu-baseline --no-variants --no-roofline $2 > $out/prof_$1.log 2>&1
  python3 - <<E
import csv, glob
f = glob.glob('$out/prof_$1/**/bench_kernel_stats.csv', recursive=True)[0]
print('--- $1: ' + ', '.join(f"{r['Name'][:14]} {float(r['AverageNs'])/1e3:.0f}" for r in csv.DictReader(open(f)) if 'k_bin3_emit' in r['Name']))
E
  rm -rf $out/prof_$1
}
make -s -C customnerf_amd/csrc -B -j48 TUNING=1 > $out/make_tuning.log 2>&1
for a in 0 15 31 47 63 7 3; do export CNERF_B3_EMIT_ABL=$a; prof abl$a ""; done
