"""Same-box A/B of the edit leg: round-5 norm launches (ops.TAIL_FUSION = False) against round 6's tail-kernel statistics / LayerNorm, alternating,
each measurement a fresh bench.run_edit (own trainer, own UNet graphs): scratch/edit_ab.py [pairs] [steps]"""
import argparse, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from customnerf_amd.sd import ops

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
args = argparse.Namespace(task="edit", gpus=1, steps=steps, warmup=5, path="run", dtype="f16", res=128, grid="synthetic", scaling="weak", dp="sharded", sds_views=1, prefit=0,
                          dp_selftest=False, graph=False, no_variants=True, no_cpu_baseline=True, no_roofline=True, gemm_table=None, rays=0, stage_events=False,
                          no_tune_traversal=False)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
res = {True: [], False: []}
for p in range(pairs):
    for fused in (False, True):
        ops.TAIL_FUSION = fused
        r = bench.run_edit(args, 1, 0, dev)
        res[fused].append(r["ms_per_step"])
        print(f"pair {p} tail_fusion={fused}: {r['ms_per_step']:.3f} ms/step, skipped {r['config']['steps_skipped_on_overflow']}, loss {r['config']['final_loss']:.4f}", flush=True)
for k in (False, True):
    v = sorted(res[k])
    print(f"tail_fusion={k}: median {v[len(v) // 2]:.3f} ms  min {v[0]:.3f}  max {v[-1]:.3f}")
