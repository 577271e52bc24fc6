#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r06q
mkdir -p $out
make -s -C customnerf_amd/csrc -B -j64 TUNING=1 > $out/make_tuning.log 2>&1; tail -1 $out/make_tuning.log
for a in "" "--grid bear" "--prefit 300"; do
echo "=== $a"
bash scratch/ab_recon.sh r06q/ab "$a" "-" "CNERF_B3_ROUND=0" "CNERF_B3_RSLIST=0" "CNERF_B3_ROUND=0 CNERF_B3_RSLIST=0" "-" "CNERF_B3_ROUND=0 CNERF_B3_RSLIST=0"
done 2>&1 | tee $out/ab_rs.txt
