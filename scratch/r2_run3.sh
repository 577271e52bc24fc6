#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
scratch/pmc_multi.sh r2pmc2 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES" 2>&1 | grep "k_field_fwd"
mkdir -p gpurun_out/r2trace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2trace -o t -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python3 - <<E
import csv
rows=list(csv.DictReader(open('gpurun_out/r2trace/t_kernel_trace.csv')))
rows=[r for r in rows if 'k_field_fwd' in r['Kernel_Name'] or 'k_grid_fwd' in r['Kernel_Name']]
for r in rows[-8:]:
    print(r['Kernel_Name'][:30], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, r.get('Grid_Size'), r.get('Workgroup_Size'), r.get('LDS_Block_Size'), r.get('VGPR_Count'), r.get('Accum_VGPR_Count'))
E
rm -f gpurun_out/r2trace/t_kernel_trace.csv
