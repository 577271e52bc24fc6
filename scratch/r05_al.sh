#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05al
for p in 0 300 0 300; do
timeout 300 python bench.py --task edit --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --edit-prefit $p 2>gpurun_out/r05al/err_$p.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('prefit $p', round(d['ms_per_step'],3), round(d['value'],2), d['config'].get('steps_skipped_on_overflow'), d['config'].get('loss_scale'), d['config'].get('final_loss'))"
done
