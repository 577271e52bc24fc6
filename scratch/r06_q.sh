#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r06q
mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_gridencoder.py tests/test_gpu_fullsize.py tests/test_gpu_train.py -q -x --timeout=600 > $out/pytest.log 2>&1; grep -E "passed|failed|error" $out/pytest.log | tail -3
make -s -C customnerf_amd/csrc -B -j64 TUNING=1 > $out/make_tuning.log 2>&1; tail -1 $out/make_tuning.log
for a in "--rays 16384" "--rays 32768" "--rays 65536" "--rays 8192"; do
echo "=== $a"
bash scratch/ab_recon.sh r06q/ab "$a" "-" "CNERF_B3_SEG_CAP=131072" "-" "CNERF_B3_SEG_CAP=131072"
done 2>&1 | tee $out/ab_rs3.txt
