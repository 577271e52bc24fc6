#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r06q
mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_render.py tests/test_gpu_train.py tests/test_gpu_sd_editing.py tests/test_gpu_raymarching.py -q -x --timeout=600 > $out/pytest.log 2>&1; grep -E "passed|failed|error" $out/pytest.log | tail -3
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python3 bench.py --task recon --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-roofline > $out/prof.log 2>&1
f=$(find $out/prof -name bench_kernel_stats.csv | head -1); python3 - <<P
import csv
for r in list(csv.DictReader(open('$f')))[:24]: print('%-60s %5s %9.1f' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
P
rm -rf $out/prof
for i in 1 2 3; do timeout 200 python3 bench.py --task recon --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-roofline 2>/dev/null | python3 -c "import json,sys; print('step %.4f ms' % json.load(sys.stdin)['ms_per_step'])"; done
