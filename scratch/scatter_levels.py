"""Cost of the binned scatter by level range: GridEncoders with the benchmark's geometry but only levels 0..L-1, backward of 2.1 M ray-ordered samples
(one view's coarse + jittered fine-like samples).  Run under rocprofv3 --kernel-trace --stats; kernels are bracketed per L by a marker-size trick:
each L runs a different number of iterations so the per-L averages can be separated from the kernel trace by dispatch order."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from customnerf_amd import scene as sc, tcnn, raymarching
from customnerf_amd.gridencoder import GridEncoder
from customnerf_amd.nerf.provider_utils import generate_rays
dev = torch.device('cuda')
torch.manual_seed(0)
opt = sc.make_opt(fp16=True)
H = W = 128; S = 128
c2w = torch.from_numpy(sc.poses(8)).to(dev)[:1]
o, d = generate_rays(c2w, *sc.intrinsics(H, W), H, W, 1.0, 'nerfstudio')
o, d = o.view(-1, 3).contiguous(), d.view(-1, 3).contiguous()
aabb = torch.tensor([-opt.bound] * 3 + [opt.bound] * 3, device=dev, dtype=torch.float32)
nears, fars = raymarching.near_far_from_aabb(o, d, aabb, opt.min_near)
z = nears[:, None] + (fars - nears)[:, None] * ((torch.arange(S, device=dev)[None] + torch.rand(H * W, S, device=dev)) / S)
xyz = (o[:, None] + d[:, None] * z[..., None]).clamp(-opt.bound, opt.bound).reshape(-1, 3).contiguous()
scale = float(np.exp2(np.log2(2048 / 16) / 15))
for L in (1, 2, 3, 5, 8, 16):
    enc = GridEncoder(num_levels=L, level_dim=2, per_level_scale=scale, base_resolution=16, log2_hashmap_size=19).to(dev)
    enc.grad_in_place = False
    with torch.autocast('cuda', dtype=torch.float16):
        for it in range(4):
            out = enc(xyz, bound=opt.bound)
            g = torch.randn_like(out) * 0.01
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out.backward(g)
            e1.record(); torch.cuda.synchronize()
            enc.embeddings.grad = None
    print(f"levels 0..{L - 1}: backward {e0.elapsed_time(e1) * 1e3:8.1f} us", flush=True)
