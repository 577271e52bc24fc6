#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r05t
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_sd_ops.py tests/test_gpu_sd_editing.py tests/test_gpu_sd_nets.py -q -x --timeout=400 > $out/pytest_sel.log 2>&1; grep -E "passed|failed" $out/pytest_sel.log
rm -rf /tmp/prev && mkdir -p /tmp/prev && tar -xf scratch/prev_tree.tar -C /tmp/prev && make -s -C /tmp/prev/customnerf_amd/csrc -j48 > $out/make_prev.log 2>&1
be() { (cd $1 && timeout 300 python bench.py --task edit --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$2', round(d['ms_per_step'],3), round(d['value'],2), d['config'].get('steps_skipped_on_overflow'))"); }
be /tmp/prev prev
be $GRAFT_REPO_ROOT new
be /tmp/prev prev
be $GRAFT_REPO_ROOT new
