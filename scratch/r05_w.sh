#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r05w
mkdir -p $out
prof() {
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$1 -o bench -- python3 bench.py --task recon --steps 10 --warmup 3 --no-cpu-baseline --no-variants --no-roofline $2 > $out/prof_$1.log 2>&1
  python3 - <<E
import csv, glob
f = glob.glob('$out/prof_$1/**/bench_kernel_stats.csv', recursive=True)[0]
print('--- $1: ' + ', '.join(f"{r['Name'][:14]} {float(r['AverageNs'])/1e3:.0f}" for r in csv.DictReader(open(f)) if 'k_bin3' in r['Name']))
E
  rm -rf $out/prof_$1
}
for cfg in "2048 1024" "1024 512" "1024 1024" "512 256" "2048 512"; do
  set -- $cfg
  make -s -C customnerf_amd/csrc -B -j48 EXTRA="-DB3_PTS=$1 -DB3_THREADS=$2" > $out/make_$1_$2.log 2>&1
  timeout 300 python -m pytest tests/test_gpu_gridencoder.py -q -x -k "scatter or binned or determin" 2>&1 | grep -E "passed|failed"
  prof pts$1_thr$2 ""
done
