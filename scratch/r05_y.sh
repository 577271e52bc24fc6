#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r05y
mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_edit -o bench -- python3 bench.py --task edit --steps 10 --warmup 3 --no-cpu-baseline --no-variants > $out/prof_edit.log 2>&1
rm -f $out/prof_edit/bench_kernel_trace.csv
timeout 300 python bench.py --task edit --steps 20 --warmup 5 --no-cpu-baseline --gemm-table $out/gemm_table.txt > $out/bench_edit.json 2> $out/bench_edit.err
ls $out/prof_edit
