#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05q
timeout 400 python scratch/edit_ops.py gpurun_out/r05q/edit_ops.txt unet > gpurun_out/r05q/log.txt 2>&1; tail -3 gpurun_out/r05q/log.txt
