#!/bin/bash
# round-6 second GPU call: graph-cache / parity tests, the SD tests after the split-K statistics change, edit + recon bench, edit kernel table
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r06b
mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_render.py tests/test_gpu_field.py tests/test_gpu_sd_ops.py tests/test_gpu_sd_nets.py tests/test_gpu_sd_editing.py -q -s --timeout=600 > $out/pytest_sel.log 2>&1; tail -8 $out/pytest_sel.log; grep "headline parity" $out/pytest_sel.log
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; python3 -c "
import json; d=json.load(open('$out/bench.json')); print('recon', d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('fine_traversal')); print({k:(round(v.get('ms_per_step',0),3)) for k,v in d.get('variants',{}).items()}); s=d['secondary']; print('edit', s['ms_per_step'], s['roofline'])"
bash scratch/edit_step_kernels.sh r06b > $out/edit_step_kernels.log 2>&1; head -40 $out/edit_step_kernels.log
