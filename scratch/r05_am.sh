#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05am
t0=$(date +%s)
timeout 900 python bench.py > gpurun_out/r05am/bench.json 2> gpurun_out/r05am/bench.err
echo "rc $? wall $(( $(date +%s) - t0 )) s"
python3 -c "
import json; d=json.load(open('gpurun_out/r05am/bench.json')); s=d['secondary']
print('recon', round(d['ms_per_step'],3), round(d['value']), 'steps', d['steps'], d['warmup'])
print({k:(round(v.get('ms_per_step',0),3), round(v.get('value',0))) for k,v in d['variants'].items()})
print('edit', round(s['ms_per_step'],3), round(s['value'],2), 'fitted', s.get('fitted_field'), 'mv', s['multi_view']['views_per_s'])"
