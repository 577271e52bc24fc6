#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r05o
mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_sd_nets.py -q -s -k "vae_encode or sds_train_step or clip_text_encoder" 2>&1 | grep -E "^\[|passed|failed" > $out/tolerances.log; cat $out/tolerances.log
timeout 300 python bench.py --task edit --steps 10 --warmup 3 --no-cpu-baseline --gemm-table $out/gemm_table.txt > $out/bench_edit.json 2> $out/bench_edit.err; head -30 $out/gemm_table.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_edit -o bench -- python3 bench.py --task edit --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $out/prof_edit.log 2>&1
rm -f $out/prof_edit/*/bench_kernel_trace.csv $out/prof_edit/bench_kernel_trace.csv
python3 - <<E
import csv, glob
f = glob.glob('$out/prof_edit/**/bench_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot / 1e6)
for r in rows[:45]: print(f"{int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.3f} ms {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:110]}")
E
