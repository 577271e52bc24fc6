#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r05h
mkdir -p $out
prof() {
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$1 -o bench -- python3 bench.py --task recon --steps 10 --warmup 3 --no-cpu-baseline --no-variants --no-roofline > $out/prof_$1.log 2>&1
  rm -f $out/prof_$1/*/bench_kernel_trace.csv $out/prof_$1/bench_kernel_trace.csv
  python3 - <<E
import csv, glob
f = glob.glob('$out/prof_$1/**/bench_kernel_stats.csv', recursive=True)[0]
print('--- $1: ' + ', '.join(f"{r['Name'][:14]} {float(r['AverageNs'])/1e3:.0f}" for r in csv.DictReader(open(f)) if 'k_bin3' in r['Name'] or 'field_bwd' in r['Name']))
E
}
make -s -C customnerf_amd/csrc -B -j48 TUNING=1 > $out/make_tuning.log 2>&1
for a in 0 15; do export CNERF_B3_EMIT_ABL=$a; prof abl$a; done
unset CNERF_B3_EMIT_ABL
export CNERF_B3_ONLY=1; prof hashed_only
export CNERF_B3_ONLY=2; prof dense_only
