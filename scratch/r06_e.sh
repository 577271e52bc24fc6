#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r06e
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_gridencoder.py -q -k "sample_major" --timeout=600 > $out/pytest_sm.log 2>&1; tail -4 $out/pytest_sm.log
timeout 600 python scratch/nan_hunt.py 3 2>&1 | tee $out/nan_release.txt
make -s -C customnerf_amd/csrc -B -j64 TUNING=1 > $out/make_tuning.log 2>&1; tail -2 $out/make_tuning.log
for spt in 8 4 32; do
  CNERF_GRID_SPT=$spt timeout 600 python -m pytest tests/test_gpu_gridencoder.py -q -k "sample_major" --timeout=600 2>&1 | tail -2
  CNERF_GRID_SPT=$spt timeout 600 python scratch/nan_hunt.py 3 2>&1 | tee -a $out/nan_spt.txt
done
CNERF_GRID_SPT=8 CNERF_GRID_TRAV=1 timeout 600 python scratch/nan_hunt.py 2 2>&1 | tee -a $out/nan_spt.txt
