#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r06l
mkdir -p $out
make -s -C customnerf_amd/csrc -B -j64 TUNING=1 > $out/make_tuning.log 2>&1; tail -2 $out/make_tuning.log
bash scratch/ab_recon.sh r06l/ab_bear "--grid bear" "-" "CNERF_B3_ONLY=1" "CNERF_B3_ONLY=2" | tee $out/ab_bear.txt
bash scratch/ab_recon.sh r06l/ab_t19 "" "-" "CNERF_B3_ONLY=1" "CNERF_B3_ONLY=2" | tee $out/ab_t19.txt
