#!/bin/bash
# A/B of env-selected variants: scratch/ab.sh "VAR=val" ... ("default" = no override); parity tests first
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/ab
python -m pytest tests/test_gpu_gridencoder.py -x -q -m gpu > gpurun_out/ab/pytest.log 2>&1; tail -3 gpurun_out/ab/pytest.log
for v in "$@"; do
  if [ "$v" = default ]; then python bench.py --task recon --no-variants --steps 30 --warmup 5 --no-cpu-baseline > "gpurun_out/ab/bench_$v.log" 2>&1
  else env $v python bench.py --task recon --no-variants --steps 30 --warmup 5 --no-cpu-baseline > "gpurun_out/ab/bench_$v.log" 2>&1; fi
  echo "$v: $(grep -o '"ms_per_step": [0-9.]*' "gpurun_out/ab/bench_$v.log" | head -1)"
done
