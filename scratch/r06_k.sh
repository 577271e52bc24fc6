#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r06k
mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_gridencoder.py tests/test_gpu_fullsize.py tests/test_gpu_uninitialised.py -q --timeout=900 -x > $out/pytest_sel.log 2>&1; tail -3 $out/pytest_sel.log
make -s -C customnerf_amd/csrc -B -j64 TUNING=1 > $out/make_tuning.log 2>&1; tail -2 $out/make_tuning.log
bash scratch/ab_recon.sh r06k/ab_bear "--grid bear" "CNERF_B3_WIDE=0" "-" "CNERF_B3_EMIT_ABL=1" "CNERF_B3_EMIT_ABL=2" "CNERF_B3_EMIT_ABL=8" "CNERF_B3_EMIT_ABL=16" | tee $out/ab_bear.txt
bash scratch/ab_recon.sh r06k/ab_bear_fit "--grid bear --prefit 300" "CNERF_B3_WIDE=0" "-" | tee $out/ab_bear_fit.txt
