#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r05ah
mkdir -p $out
prof() {
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$1 -o bench -- python3 bench.py --task recon --steps 10 --warmup 3 --no-cpu-baseline --no-variants --no-roofline $2 > $out/prof_$1.log 2>&1
  python3 - <<E
import csv, glob
f = glob.glob('$out/prof_$1/**/bench_kernel_stats.csv', recursive=True)[0]
print('--- $1: ' + ', '.join(f"{r['Name'][:14]} {float(r['AverageNs'])/1e3:.0f}" for r in csv.DictReader(open(f)) if 'k_bin3_emit' in r['Name'] or 'k_bin3_accum' in r['Name']))
E
  rm -rf $out/prof_$1
}
make -s -C customnerf_amd/csrc -B -j48 TUNING=1 > $out/make_tuning.log 2>&1
for g in 0 512 1024 2048 4096; do export CNERF_B3_EMIT_GRID=$g; prof grid$g ""; done
export CNERF_B3_EMIT_GRID=512 CNERF_B3_EMIT_ABL=63; prof grid512_loads ""
