#!/bin/bash
# round-6 third GPU call: SD tests after the LN fusion + flush fix, edit bench + kernel table, bear kernel table + gather A/B on the bear table
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r06c
mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_sd_ops.py tests/test_gpu_sd_nets.py tests/test_gpu_sd_editing.py tests/test_gpu_sd_clip.py -q -s --timeout=600 > $out/pytest_sel.log 2>&1; tail -6 $out/pytest_sel.log; grep "sds tiny\|unet\b" $out/pytest_sel.log | head
timeout 600 python bench.py --task edit --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_edit.json 2> $out/bench_edit.err; python3 -c "
import json; s=json.load(open('$out/bench_edit.json')); print('edit', s['ms_per_step'], s['roofline']['frac'], s.get('multi_view',{}).get('views_per_s'))"
bash scratch/edit_step_kernels.sh r06c > $out/edit_step_kernels.log 2>&1; head -3 $out/edit_step_kernels.log; head -12 $out/edit_step_kernels.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_bear -o bench -- python3 bench.py --task recon --grid bear --steps 20 --warmup 5 --no-cpu-baseline --no-variants > $out/prof_bear.log 2>&1
rm -f $out/prof_bear/bench_kernel_trace.csv; f=$(find $out/prof_bear -name bench_kernel_stats.csv | head -1); cp $f $out/bear_kernel_stats.csv; rm -rf $out/prof_bear
python3 - <<P
import csv
rows=list(csv.DictReader(open('$out/bear_kernel_stats.csv')))
for r in rows[:22]: print('%-60s %5s %9.1f' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
P
make -s -C customnerf_amd/csrc -B -j64 TUNING=1 > $out/make_tuning.log 2>&1; tail -2 $out/make_tuning.log
bash scratch/ab_recon.sh r06c/ab_bear "--no-tune-traversal --grid bear" "-" "CNERF_GRID_TRAV=1 CNERF_GRID_SPT=8" "CNERF_GRID_TRAV=1 CNERF_GRID_SPT=16" "CNERF_GRID_TRAV=1 CNERF_GRID_SPT=32" "CNERF_GRID_SWIZZLE=1" "CNERF_GRID_SWIZZLE=0" | tee $out/ab_bear.txt
bash scratch/ab_recon.sh r06c/ab_init "--no-tune-traversal" "-" "CNERF_GRID_TRAV=1 CNERF_GRID_SPT=16" "CNERF_GRID_TRAV=1 CNERF_GRID_SPT=32" | tee $out/ab_init.txt
