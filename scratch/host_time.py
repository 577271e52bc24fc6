"""host (enqueue) time per recon step: how far ahead of the GPU does Python run?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from customnerf_amd import scene as sc, tcnn
from customnerf_amd.nerf.network_grid import NeRFNetwork
from customnerf_amd.nerf.provider_utils import generate_rays
from customnerf_amd.trainer import ReconTrainer
dev = torch.device("cuda", 0)
tcnn.set_default_dtype(torch.float16)
torch.manual_seed(0)
opt = sc.make_opt(cuda_ray=False, fp16=True)
model = NeRFNetwork(opt).to(dev)
H = W = 128; V = 8
c2w = torch.from_numpy(sc.poses(V)).to(dev)
rays_o, rays_d = generate_rays(c2w, *sc.intrinsics(H, W), H, W, 1.0, 'nerfstudio')
rays_o, rays_d = rays_o.view(V, 1, H * W, 3), rays_d.view(V, 1, H * W, 3)
rgb, mask = sc.targets(V, H, W); rgb, mask = rgb.to(dev), mask.to(dev)
tr = ReconTrainer(model, opt, fp16=True)
kw = dict(num_steps=opt.num_steps, upsample_steps=opt.upsample_steps, dt_gamma=0, max_steps=opt.max_steps)
for i in range(5): tr.train_step(rays_o[i % V], rays_d[i % V], rgb[i % V], mask[i % V], **kw)
torch.cuda.synchronize()
ts = []
for i in range(6):
    t0 = time.perf_counter()
    tr.train_step(rays_o[i % V], rays_d[i % V], rgb[i % V], mask[i % V], **kw)
    ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()          # host never queues behind the GPU: pure enqueue cost
print("host enqueue ms per step:", [round(t * 1e3, 3) for t in ts])
