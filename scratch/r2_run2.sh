#!/bin/bash
cd $GRAFT_REPO_ROOT
scratch/tr_read_test > gpurun_out/tr_read_test.log 2>&1
head -20 gpurun_out/tr_read_test.log
python3 bench.py --no-cpu-baseline > gpurun_out/r2_fwdfix_recon.json 2> gpurun_out/r2_fwdfix_recon.err
cat gpurun_out/r2_fwdfix_recon.json
timeout 900 python3 -m pytest tests/test_gpu_field.py tests/test_gpu_raymarching.py tests/test_gpu_render.py tests/test_gpu_train.py tests/test_gpu_sd_ops.py -x -q 2>&1 | tail -15
scratch/prof.sh r2prof_a | head -30
