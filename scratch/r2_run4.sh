#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_field.py tests/test_gpu_train.py tests/test_gpu_render.py -x -q 2>&1 | tail -15
python3 bench.py --no-cpu-baseline > gpurun_out/r2_x4_recon.json 2> gpurun_out/r2_x4_recon.err
cat gpurun_out/r2_x4_recon.json | cut -c1-400
scratch/prof.sh r2prof_b | head -12
