#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r06q
mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_gridencoder.py tests/test_gpu_train.py tests/test_gpu_fullsize.py tests/test_gpu_render.py tests/test_gpu_uninitialised.py -q --timeout=600 > $out/pytest.log 2>&1; grep -E "passed|failed|error" $out/pytest.log | tail -3; grep -E "^FAILED|^ERROR" $out/pytest.log | head
for i in 1 2 3; do timeout 200 python3 bench.py --task recon --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-roofline 2>/dev/null | python3 -c "import json,sys; print('step %.4f ms' % json.load(sys.stdin)['ms_per_step'])"; done
