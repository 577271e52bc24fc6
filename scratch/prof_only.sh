#!/bin/bash
# the two kernel-stats profiles of the end-of-round set only: scratch/prof_only.sh <tag>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/$1
mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_recon -o bench -- python3 bench.py --task recon --steps 20 --warmup 5 --no-cpu-baseline --no-variants > $out/prof_recon.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_edit -o bench -- python3 bench.py --task edit --steps 10 --warmup 3 --no-cpu-baseline --no-variants > $out/prof_edit.log 2>&1
rm -f $out/prof_recon/bench_kernel_trace.csv $out/prof_edit/bench_kernel_trace.csv
grep -o '"ms_per_step": [0-9.]*' $out/prof_recon.log | head -1; grep -o '"ms_per_step": [0-9.]*' $out/prof_edit.log | head -1
