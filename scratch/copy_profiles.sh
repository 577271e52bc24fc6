#!/bin/bash
# scratch/copy_profiles.sh <tag>: the judged copies of a scratch/final_r06.sh run (gpurun_out/<tag>/ -> profiles/r06_*)
o=gpurun_out/$1
set -e
cp $o/bench.json profiles/r06_bench.json
cp $o/prof_recon/bench_kernel_stats.csv profiles/r06_bench_recon_kernel_stats.csv
cp $o/prof_edit/bench_kernel_stats.csv profiles/r06_bench_edit_kernel_stats.csv
cp $o/prof_bear/bench_kernel_stats.csv profiles/r06_bench_recon_bear_kernel_stats.csv
cp $o/gather_pmc.json profiles/r06_gather_pmc.json
for c in FETCH_SIZE WRITE_SIZE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr; do cp $o/pmc_${c}_gather_rows.csv profiles/r06_pmc_${c}_gather_rows.csv; done
cp $o/sd_pmc_MfmaUtil.json profiles/r06_sd_mfma_pmc.json
cp $o/sd_pmc_LdsUtil.json profiles/r06_sd_lds_pmc.json
cp $o/edit_step_kernels.txt profiles/r06_edit_step_kernels.txt
cp $o/sweep/batch_sweep.json profiles/r06_batch_sweep.json
cp $o/sweep/batch_sweep.txt profiles/r06_batch_sweep.txt
