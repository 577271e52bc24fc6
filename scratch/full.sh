#!/bin/bash
# full GPU suite + smoke + driver-style bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/${1:-r05full}
mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu --timeout=400 > $out/pytest_gpu.log 2>&1; tail -6 $out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; python3 -c "
import json; d=json.load(open('$out/bench.json')); print('recon', d['ms_per_step'], d['value'], 'roofline', d['roofline']['frac']); print({k:(round(v.get('ms_per_step',0),3), round(v.get('value',0)), v.get('exchange_ms')) for k,v in d.get('variants',{}).items()}); s=d['secondary']; print('edit', s['ms_per_step'], s['value'], s['roofline']['frac'], s.get('multi_view',{}).get('views_per_s'))"
