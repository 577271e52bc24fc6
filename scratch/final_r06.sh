#!/bin/bash
# end-of-round-6 measurement set: scratch/final_r06.sh <tag>  (every step under its own timeout); summaries are copied to profiles/ by hand
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=$1
out=gpurun_out/$tag
mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu --timeout=400 > $out/pytest_gpu.log 2>&1; tail -2 $out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; tail -c 300 $out/bench.json; echo
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_recon -o bench -- python3 bench.py --task recon --steps 20 --warmup 5 --no-cpu-baseline --no-variants > $out/prof_recon.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_edit -o bench -- python3 bench.py --task edit --steps 10 --warmup 3 --no-cpu-baseline --no-variants > $out/prof_edit.log 2>&1
rm -f $out/prof_recon/bench_kernel_trace.csv $out/prof_edit/bench_kernel_trace.csv
for c in FETCH_SIZE WRITE_SIZE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -o b -- python3 bench.py --task recon --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-variants > $out/pmc_$c.log 2>&1
  python3 - <<E
import csv, collections
try:
    rows = list(csv.DictReader(open('$out/pmc_$c/b_counter_collection.csv')))
    acc = collections.defaultdict(list)
    for r in rows: acc[r['Kernel_Name'][:34]].append(float(r['Counter_Value']))
    print('$c', {k: round(sum(v) / len(v), 1) for k, v in acc.items() if 'k_grid_fwd' in k or 'k_bin3' in k or 'k_field' in k})
    with open('$out/pmc_${c}_gather_rows.csv', 'w') as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys())); w.writeheader()
        for r in rows:
            if 'k_grid_fwd' in r['Kernel_Name']: w.writerow(r)
except Exception as e:
    print('$c', 'no data', e)
E
  rm -rf $out/pmc_$c
done
for c in MfmaUtil LdsUtil; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmce_$c -o b -- python3 bench.py --task edit --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-variants > $out/pmce_$c.log 2>&1
  python3 - <<E
import csv, collections, json
try:
    rows = list(csv.DictReader(open('$out/pmce_$c/b_counter_collection.csv')))
    acc = collections.defaultdict(list)
    for r in rows: acc[r['Kernel_Name'][:60]].append(float(r['Counter_Value']))
    d = {k: {"mean": round(sum(v) / len(v), 2), "max": round(max(v), 2), "launches": len(v)} for k, v in acc.items() if 'k_sd_' in k or 'k_field' in k}
    json.dump(d, open('$out/sd_pmc_$c.json', 'w'), indent=1)
    print('$c', {k[:40]: v["mean"] for k, v in list(d.items())[:8]})
except Exception as e:
    print('$c', 'no data', e)
E
  rm -rf $out/pmce_$c
done

python3 scratch/gather_pmc_summary.py $out
bash scratch/edit_step_kernels.sh $tag > $out/edit_step_kernels.log 2>&1; head -3 $out/edit_step_kernels.log
# bear table: kernel stats and the batch sweep (item 1a)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_bear -o bench -- python3 bench.py --task recon --grid bear --steps 20 --warmup 5 --no-cpu-baseline --no-variants > $out/prof_bear.log 2>&1
rm -f $out/prof_bear/bench_kernel_trace.csv
bash scratch/batch_sweep.sh $tag/sweep > $out/batch_sweep.log 2>&1; tail -30 $out/batch_sweep.log
