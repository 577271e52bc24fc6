#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r05s
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_sd_ops.py tests/test_gpu_sd_editing.py tests/test_gpu_sd_nets.py -q -x --timeout=400 > $out/pytest_sel.log 2>&1; tail -15 $out/pytest_sel.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl"
bash scratch/edit_step_kernels.sh r05s
timeout 300 python bench.py --task edit --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_edit.json 2> $out/bench_edit.err; python3 -c "
import json; d=json.load(open('$out/bench_edit.json')); print('edit', d['ms_per_step'], d['value'], d['roofline']['frac'], d['config'].get('steps_skipped_on_overflow'), d['config'].get('loss_scale'))"
