#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r05af
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_gridencoder.py tests/test_gpu_fullsize.py tests/test_gpu_train.py tests/test_gpu_render.py -q -x --timeout=400 2>&1 | grep -E "passed|failed"
timeout 300 python bench.py --task recon --steps 20 --warmup 5 --no-cpu-baseline > $out/bench.json 2> $out/bench.err; python3 -c "
import json; d=json.load(open('$out/bench.json')); print(d['ms_per_step'], d['value']); print({k:(round(v.get('ms_per_step',0),3), round(v.get('value',0))) for k,v in d.get('variants',{}).items()})"
