#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for m in plain sharded; do
  sed "s/for mode in (\"plain\", \"sharded\"):/for mode in (\"$m\",):/" scratch/edit_dp_world1.py > scratch/_edw_$m.py
  mkdir -p gpurun_out/dpcmp_$m
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/dpcmp_$m -o e -- python3 scratch/_edw_$m.py > gpurun_out/dpcmp_$m/log 2>&1
  echo "== $m"; python3 scratch/edit_gaps.py gpurun_out/dpcmp_$m/e_kernel_trace.csv | head -14
  rm -f gpurun_out/dpcmp_$m/e_kernel_trace.csv scratch/_edw_$m.py
done
