#!/bin/bash
# A/B with the gather's live roofline figure: scratch/ab2.sh "VAR=val" ...
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = default ]; then timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > "gpurun_out/ab/b2_$v.log" 2>&1
  else timeout 200 env $v python bench.py --steps 30 --warmup 5 --no-cpu-baseline > "gpurun_out/ab/b2_$v.log" 2>&1; fi
  echo "$v: $(grep -o '"ms_per_step": [0-9.]*' "gpurun_out/ab/b2_$v.log") $(grep -o '"avg_launch_ms": [0-9.]*' "gpurun_out/ab/b2_$v.log") $(grep -o '"frac": [0-9.]*' "gpurun_out/ab/b2_$v.log")"
done
