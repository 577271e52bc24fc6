#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r05ag
mkdir -p $out
prof() {
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/prof_$1 -o bench -- python3 bench.py --task recon --grid bear --steps 10 --warmup 3 --no-cpu-baseline --no-variants --no-roofline $2 > $out/prof_$1.log 2>&1
  python3 - <<E
import csv, glob, collections
f = glob.glob('$out/prof_$1/**/bench_kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
print('--- $1: ' + ', '.join(f"{k[:12]} {sum(v[-10:]) / len(v[-10:]) / 1e3:.0f}" for k, v in d.items() if 'k_bin2_accum' in k or 'k_bin2_emit' in k))
E
  rm -rf $out/prof_$1
}
timeout 600 python -m pytest tests/test_gpu_gridencoder.py tests/test_gpu_fullsize.py -q -x 2>&1 | grep -E "passed|failed"
prof combine_init ""; prof combine_fit "--prefit 300"
make -s -C customnerf_amd/csrc -B -j48 EXTRA="-DB2_COMBINE=0" > $out/make.log 2>&1
prof plain_init ""; prof plain_fit "--prefit 300"
