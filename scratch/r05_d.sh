#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r05d
mkdir -p $out
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
timeout 600 python scratch/gemm_graph_bench.py > $out/gemm_graph.log 2>&1; grep -E "conv|M8192|M4096" $out/gemm_graph.log
timeout 900 python scratch/edit_glue.py > $out/edit_glue.log 2>&1; head -60 $out/edit_glue.log
