#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r05ai
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
timeout 600 python -m pytest tests/test_gpu_gridencoder.py tests/test_gpu_fullsize.py -q -x 2>&1 | grep -E "passed|failed"
prof staged ""
make -s -C customnerf_amd/csrc -B -j48 EXTRA="-DB3_STAGE_POS=0" > $out/make.log 2>&1
prof direct ""
