cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for e in 0 1 2 3; do
export CNERF_B2_ACC_EXP=$e
mkdir -p gpurun_out/accexp$e
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/accexp$e -o b -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/accexp$e/log 2>&1
echo "EXP $e: $(grep k_bin2_accum gpurun_out/accexp$e/b_kernel_stats.csv | cut -d, -f1-4 | cut -c1-30,150-)"
rm -f gpurun_out/accexp$e/b_kernel_trace.csv
done
