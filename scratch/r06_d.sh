#!/bin/bash
# round-6 fourth GPU call: tests after the tile-mapped statistics / swizzle choice / graphed DP step, full bench, edit kernel table
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r06d
mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_gridencoder.py tests/test_gpu_render.py tests/test_gpu_sd_ops.py tests/test_gpu_sd_nets.py tests/test_gpu_dp_two_ranks.py -q --timeout=900 > $out/pytest_sel.log 2>&1; tail -6 $out/pytest_sel.log
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; python3 -c "
import json; d=json.load(open('$out/bench.json')); print('recon', d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('fine_traversal')); print({k:(round(v.get('ms_per_step',0),3)) for k,v in d.get('variants',{}).items()}); print(d['variants'].get('small_batch')); s=d['secondary']; print('edit', s['ms_per_step'], s['roofline']['frac'], s.get('multi_view',{}).get('views_per_s'))"
bash scratch/edit_step_kernels.sh r06d > $out/edit_step_kernels.log 2>&1; head -3 $out/edit_step_kernels.log; head -8 $out/edit_step_kernels.txt; grep -n "epilogue\|concat\|layernorm" $out/edit_step_kernels.txt
