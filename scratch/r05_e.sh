#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r05e
mkdir -p $out
timeout 300 python scratch/scatter_debug.py > $out/scatter_debug.log 2>&1; cat $out/scatter_debug.log | tail -12
timeout 1500 python -m pytest tests/test_gpu_gridencoder.py tests/test_gpu_fullsize.py tests/test_gpu_train.py tests/test_gpu_render.py tests/test_gpu_sd_ops.py tests/test_gpu_field.py -q > $out/pytest_sel.log 2>&1; tail -8 $out/pytest_sel.log
timeout 600 python bench.py --task recon --steps 20 --warmup 5 --no-cpu-baseline > $out/bench.json 2> $out/bench.err; python3 -c "
import json; d=json.load(open('$out/bench.json')); print(d['ms_per_step'], d['value']); print({k:(v.get('ms_per_step'), v.get('value'), v.get('exchange_ms')) for k,v in d.get('variants',{}).items()})"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_recon -o bench -- python3 bench.py --task recon --steps 20 --warmup 5 --no-cpu-baseline --no-variants > $out/prof_recon.log 2>&1
rm -f $out/prof_recon/*/bench_kernel_trace.csv $out/prof_recon/bench_kernel_trace.csv
python3 - <<E
import csv, glob
f = glob.glob('$out/prof_recon/**/bench_kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print(f"{int(r['Calls']):5d} {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:90]}")
E
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_fit -o bench -- python3 bench.py --task recon --steps 20 --warmup 5 --no-cpu-baseline --no-variants --prefit 300 > $out/prof_fit.log 2>&1
rm -f $out/prof_fit/*/bench_kernel_trace.csv $out/prof_fit/bench_kernel_trace.csv
python3 - <<E
import csv, glob
f = glob.glob('$out/prof_fit/**/bench_kernel_stats.csv', recursive=True)[0]
print('--- prefit 300')
for r in list(csv.DictReader(open(f)))[:10]:
    print(f"{int(r['Calls']):5d} {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:90]}")
E
# tuning build for the GEMM schedule A/B
make -s -C customnerf_amd/csrc -B -j48 TUNING=1 > $out/make_tuning.log 2>&1; tail -2 $out/make_tuning.log
for v in "CNERF_SG_BIG=0" "CNERF_SGB_SCHED=0" "CNERF_SGB_SCHED=1" "CNERF_SGB_SCHED=2" "CNERF_SG_BIG=0" "CNERF_SGB_SCHED=1"; do
  echo "== $v"; env $v timeout 300 python scratch/gemm_graph_bench.py 2>/dev/null | grep -E "conv B1|M8192 N2560|M4096 N4096"
done > $out/gemm_ab.log 2>&1; cat $out/gemm_ab.log
