"""Does the ORDER of the sample list matter to k_grid_fwd_fast?  Same 1.05 M coarse sample positions of one 128x128 view (bench geometry),
listed (A) ray-major [ray][s] with row-major pixels (what run() produces), (B) 8x8-pixel tiles, depth-major inside a tile [tile][s][ray in tile],
(C) 64-pixel row segments, depth-major, (D) 4x4 pixels x 4 depths per wave, (E) random permutation.  Timed with events around encode_into."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from customnerf_amd import scene as sc, tcnn, raymarching
from customnerf_amd.nerf.network_grid import NeRFNetwork
from customnerf_amd.nerf.provider_utils import generate_rays
dev = torch.device('cuda')
tcnn.set_default_dtype(torch.float16)
torch.manual_seed(0)
opt = sc.make_opt(fp16=True)
model = NeRFNetwork(opt).to(dev)
enc = model.pos_en
H = W = 128; S = 64
c2w = torch.from_numpy(sc.poses(8)).to(dev)[:1]
o, d = generate_rays(c2w, *sc.intrinsics(H, W), H, W, 1.0, 'nerfstudio')
o, d = o.view(-1, 3).contiguous(), d.view(-1, 3).contiguous()
aabb = torch.tensor([-opt.bound] * 3 + [opt.bound] * 3, device=dev, dtype=torch.float32)
nears, fars = raymarching.near_far_from_aabb(o, d, aabb, opt.min_near)
g = torch.Generator(device=dev); g.manual_seed(0)
for label, jitter in (("coarse (jittered uniform)", True), ("fine-like (clustered around a surface)", False)):
    if jitter:
        z = nears[:, None] + (fars - nears)[:, None] * ((torch.arange(S, device=dev)[None] + torch.rand(H * W, S, device=dev, generator=g)) / S)
    else:
        mid = 0.5 * (nears + fars) + 0.1 * torch.sin(torch.arange(H * W, device=dev) * 0.01)
        z = (mid[:, None] + 0.08 * torch.randn(H * W, S, device=dev, generator=g)).sort(dim=1).values
    xyz = o[:, None] + d[:, None] * z[..., None]                       # [rays, S, 3]
    xyz = ((xyz.clamp(-opt.bound, opt.bound) + opt.bound) / (2 * opt.bound)).view(H, W, S, 3)
    orders = {
        "A ray-major, row-major pixels": xyz.reshape(-1, 3),
        "B 8x8 tiles, depth-major in tile": xyz.view(H // 8, 8, W // 8, 8, S, 3).permute(0, 2, 4, 1, 3, 5).reshape(-1, 3),
        "B2 8x8 tiles, ray-major in tile": xyz.view(H // 8, 8, W // 8, 8, S, 3).permute(0, 2, 1, 3, 4, 5).reshape(-1, 3),
        "C 64-pixel row segments, depth-major": xyz.view(H, W // 64, 64, S, 3).permute(0, 1, 3, 2, 4).reshape(-1, 3),
        "D 4x4 pixels x 4 depths per wave": xyz.view(H // 4, 4, W // 4, 4, S // 4, 4, 3).permute(0, 2, 4, 1, 3, 5, 6).reshape(-1, 3),
        "D2 2x2 pixels x 16 depths per wave": xyz.view(H // 2, 2, W // 2, 2, S // 16, 16, 3).permute(0, 2, 4, 1, 3, 5, 6).reshape(-1, 3),
        "E random": xyz.reshape(-1, 3)[torch.randperm(H * W * S, device=dev, generator=g)],
    }
    print(label)
    for name, pts in orders.items():
        pts = pts.contiguous()
        out = torch.empty(enc.num_levels, pts.shape[0], enc.level_dim, dtype=torch.float16, device=dev)
        for _ in range(3):
            enc.encode_into(pts, out, 0, half=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            enc.encode_into(pts, out, 0, half=True)
        e1.record(); torch.cuda.synchronize()
        print(f"  {name:42s} {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us", flush=True)
