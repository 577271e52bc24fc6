#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for w in 0.3 0.45 0.6 0.7; do
  mkdir -p gpurun_out/dw_$w
  CNERF_GRID_DENSE_W=$w rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dw_$w -o b -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
  python3 - <<E
import csv
rows=list(csv.DictReader(open('gpurun_out/dw_$w/b_kernel_stats.csv')))
for r in rows:
    if 'k_grid_fwd' in r['Name']: print('dense_w=$w', r['Name'][:30], float(r['AverageNs'])/1e3)
E
  rm -rf gpurun_out/dw_$w
done
scratch/pmc_multi.sh r2pmc3 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "TA_BUSY_avr TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" 2>&1 | grep "k_grid_fwd"
