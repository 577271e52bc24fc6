#!/bin/bash
# batch-size sweep of the recon step's kernels: scratch/batch_sweep.sh <tag> [extra bench args]
# one rocprofv3 kernel-stats run per ray count; scratch/batch_sweep_fit.py fits a + b N per kernel -> gpurun_out/<tag>/batch_sweep.json
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
for n in 2048 4096 8192 16384 32768; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/n$n -o bench -- python3 bench.py --task recon --rays $n --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-roofline "$@" > $out/n$n.log 2>&1
  rm -f $out/n$n/bench_kernel_trace.csv
  f=$(find $out/n$n -name bench_kernel_stats.csv | head -1)
  [ -n "$f" ] && cp $f $out/stats_$n.csv
  rm -rf $out/n$n
  grep -o '"ms_per_step": [0-9.]*' $out/n$n.log | head -1
done
python3 scratch/batch_sweep_fit.py $out > $out/batch_sweep.txt; cat $out/batch_sweep.txt
