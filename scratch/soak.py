"""odd shapes through the trainers: finite losses / parameters, no exceptions"""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from customnerf_amd import scene as sc, tcnn
from customnerf_amd.nerf.network_grid import NeRFNetwork
from customnerf_amd.nerf.provider_utils import generate_rays
from customnerf_amd.trainer import ReconTrainer
dev = torch.device("cuda", 0)
def run(fp16, cuda_ray, N, num_steps, upsample_steps, L=16, steps=4, **okw):
    tcnn.set_default_dtype(torch.float16 if fp16 else torch.float32)
    torch.manual_seed(0)
    opt = sc.make_opt(cuda_ray=cuda_ray, fp16=fp16, **okw)
    opt.num_steps, opt.upsample_steps = num_steps, upsample_steps
    model = NeRFNetwork(opt).to(dev)
    H = W = 128
    c2w = torch.from_numpy(sc.poses(2)).to(dev)
    ro, rd = generate_rays(c2w, *sc.intrinsics(H, W), H, W, 1.0, 'nerfstudio')
    ro, rd = ro.view(2, 1, H * W, 3)[:, :, :N].contiguous(), rd.view(2, 1, H * W, 3)[:, :, :N].contiguous()
    rgb, mask = sc.targets(2, H, W); rgb, mask = rgb.to(dev)[:, :N].contiguous(), mask.to(dev)[:, :N].contiguous()
    if cuda_ray:
        import numpy as np
        from customnerf_amd import raymarching
        grid = torch.from_numpy(sc.sphere_density_grid(model.cascade, 128, opt.bound, 1.0, 100.0)).to(dev)
        model.density_grid.copy_(grid)
        model.density_bitfield = raymarching.packbits(model.density_grid, 10.0, model.density_bitfield)
    tr = ReconTrainer(model, opt, fp16=fp16)
    kw = dict(num_steps=num_steps, upsample_steps=upsample_steps, dt_gamma=0, max_steps=opt.max_steps)
    losses = []
    for i in range(steps):
        loss, _ = tr.train_step(ro[i % 2], rd[i % 2], rgb[i % 2], mask[i % 2], **kw)
        losses.append(float(loss))
    ok = all(l == l and abs(l) < 1e6 for l in losses) and all(bool(torch.isfinite(p).all()) for p in model.parameters())
    print(f"fp16={fp16} cuda_ray={cuda_ray} N={N} T={num_steps}+{upsample_steps} {okw}: losses {[round(l, 5) for l in losses]} {'OK' if ok else 'BAD'}", flush=True)
    return ok
good = True
for fp16 in (True, False):
    good &= run(fp16, False, 16384, 64, 64)
    good &= run(fp16, False, 1000, 64, 64)                 # ragged ray count
    good &= run(fp16, False, 4097, 48, 32)                 # non-split path (T != t)
    # (upsample_steps = 0 is not a case: the reference's run() raises there too under train_conf — `weights` is only bound inside `if upsample_steps > 0`,
    # renderer.py:333-384 — and this package's run() says so with a ValueError)
    good &= run(fp16, True, 5000, 64, 64)                  # occupancy-march path
    good &= run(fp16, False, 2048, 64, 64, soft_mask=True, train_conf=0.01)
print("ALL OK" if good else "FAILURES")
