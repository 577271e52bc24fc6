"""scratch/final_r06.sh helper: <out>/pmc_<COUNTER>_gather_rows.csv (one rocprofv3 --pmc pass each) -> <out>/gather_pmc.json, the file bench.py's
`roofline.traffic` reads from profiles/.  Usage: python3 scratch/gather_pmc_summary.py <out-dir>"""
import collections
import csv
import json
import sys

out = sys.argv[1]
COUNTERS = {'FETCH_SIZE': 'fetch_size_kb', 'WRITE_SIZE': 'write_size_kb', 'TCP_TCC_READ_REQ_sum': 'tcp_tcc_read_req',
            'TCP_TOTAL_CACHE_ACCESSES_sum': 'tcp_total_cache_accesses', 'TA_BUSY_avr': 'ta_busy_avr_cycles'}
KERNELS = (('k_grid_fwd_fast_sm', 'k_grid_fwd_fast_sm<16> (sample-major)'), ('k_grid_fwd_fast(', 'k_grid_fwd_fast (level-major)'))
by = collections.defaultdict(dict)
for c, key in COUNTERS.items():
    try:
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f'{out}/pmc_{c}_gather_rows.csv')):
            for pat, name in KERNELS:
                if pat in r['Kernel_Name']:
                    acc[name].append(float(r['Counter_Value']))
                    break
        for name, v in acc.items():
            by[name][key] = sum(v) / len(v)
            by[name]['launches_in_pass'] = len(v)
    except Exception as e:                                   # a pass that produced nothing leaves its key null
        print(c, 'no rows', e)
# a timed step launches ONE gather per pass (coarse: level-major; importance: the TraversalTuner's choice): the headline mean is 1:1 over the
# kernels seen, not weighted by this short pass's launch counts (its untimed trial launches skew them)
f16 = {"points": 1048576}
for key in COUNTERS.values():
    vals = [d[key] for d in by.values() if key in d]
    f16[key] = sum(vals) / len(vals) if vals else None
note = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCP_TCC_READ_REQ_sum / TCP_TOTAL_CACHE_ACCESSES_sum / TA_BUSY_avr (separate passes, --kernel-trace only: "
        "scratch/final_r06.sh) on bench.py --task recon --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-variants, end of round 6, kernels "
        "k_grid_fwd_fast (coarse pass, level-major) and k_grid_fwd_fast_sm (importance pass, sample-major when the TraversalTuner picks it), 1,048,576 "
        "samples x 16 levels per launch; FETCH_SIZE / WRITE_SIZE in KB; FETCH_SIZE is half the real fetched bytes for streaming reads on gfx950 "
        "(MI355X_MICROARCH.md) and is doubled by bench.py.  Algorithmic bytes of one launch: 588 B x 1,048,576 = 0.617 GB.  Rows: "
        "profiles/r06_pmc_*_gather_rows.csv.  A timed step launches ONE gather per pass: `f16` is the 1:1 mean of the kernels' per-launch averages "
        "(`by_kernel`), not the launch-count-weighted mean of this short pass.")
json.dump({"_note": note, "f16": f16, "by_kernel": by}, open(f'{out}/gather_pmc.json', 'w'), indent=1)
print(f16)
