#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r06o
mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_gridencoder.py tests/test_gpu_fullsize.py tests/test_gpu_uninitialised.py tests/test_gpu_train.py tests/test_gpu_render.py -q --timeout=900 -x 2>&1 | tail -4
make -s -C customnerf_amd/csrc -B -j64 TUNING=1 > $out/make_tuning.log 2>&1; tail -2 $out/make_tuning.log
bash scratch/ab_recon.sh r06o/ab "" "CNERF_B3_REGION_MIN_BINS=1000" "-" "CNERF_B3_REGION_MIN_BINS=16" "CNERF_B3_REGION_MIN_BINS=1000" "-" | tee $out/ab.txt
bash scratch/ab_recon.sh r06o/ab_bear "--grid bear" "CNERF_B3_REGION_MIN_BINS=1000" "-" "CNERF_B3_REGION_MIN_BINS=16" "CNERF_B3_REGION_MIN_BINS=1000" "-" | tee $out/ab_bear.txt
bash scratch/ab_recon.sh r06o/ab_fit "--prefit 300" "CNERF_B3_REGION_MIN_BINS=1000" "-" | tee $out/ab_fit.txt
bash scratch/ab_recon.sh r06o/ab_bear_fit "--grid bear --prefit 300" "CNERF_B3_REGION_MIN_BINS=1000" "-" | tee $out/ab_bear_fit.txt
