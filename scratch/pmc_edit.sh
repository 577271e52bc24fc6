#!/bin/bash
# MfmaUtil / LdsUtil per kernel instantiation on the edit bench (separate --pmc passes) -> gpurun_out/<tag>/sd_mfma_pmc.json
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=$1
mkdir -p gpurun_out/$tag
for c in MfmaUtil LdsUtil; do
  timeout 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/$tag/$c -o b -- python3 bench.py --task edit --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/$c.log 2>&1
done
python3 - <<E
import csv, json, collections, re
out = {}
for c in ('MfmaUtil', 'LdsUtil'):
    try:
        rows = list(csv.DictReader(open('gpurun_out/$tag/%s/b_counter_collection.csv' % c)))
    except Exception as e:
        out[c] = {'error': repr(e)}; continue
    acc = collections.defaultdict(list)
    for r in rows:
        name = r['Kernel_Name']
        if not any(s in name for s in ('k_sd_', 'k_field', 'k_gn_', 'k_bin2', 'k_grid_fwd')): continue
        name = re.sub(r'\(.*', '', name).replace('void ', '')
        acc[name].append(float(r['Counter_Value']))
    out[c] = {k: {'launches': len(v), 'mean': round(sum(v) / len(v), 2), 'max': round(max(v), 2)} for k, v in sorted(acc.items())}
json.dump(out, open('gpurun_out/$tag/sd_mfma_pmc.json', 'w'), indent=1)
for k, v in out.get('MfmaUtil', {}).items():
    if isinstance(v, dict) and v.get('mean', 0) > 1: print(k[:60], v)
E
rm -f gpurun_out/$tag/*/b_kernel_trace.csv
