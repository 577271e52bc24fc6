#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r06q
mkdir -p $out
for a in "--grid bear" "--prefit 300" "--rays 2048" ""; do
for rep in 1 2; do
for v in "" "--no-fused-table-adam"; do
timeout 200 python3 bench.py --task recon --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-roofline $a $v 2>/dev/null | python3 -c "import json,sys; print('[$a $v] step %.4f ms' % json.load(sys.stdin)['ms_per_step'])"
done; done; done 2>&1 | tee $out/ab_fadam2.txt
