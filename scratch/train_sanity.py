"""300 recon steps on the synthetic scene: loss trend, loss-scale trajectory, skipped steps (overflow handling end to end)"""
import os, sys
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
if len(sys.argv) > 1:
    tr.scaler.state[0] = float(sys.argv[1])                  # start from a loss scale that overflows: the back-off path
kw = dict(num_steps=opt.num_steps, upsample_steps=opt.upsample_steps, dt_gamma=0, max_steps=opt.max_steps)
losses, scales = [], []
for i in range(300):
    loss, _ = tr.train_step(rays_o[i % V], rays_d[i % V], rgb[i % V], mask[i % V], **kw)
    if i % 25 == 0 or i == 299:
        losses.append(round(float(loss), 5)); scales.append(tr.scaler.get_scale())
print("loss every 25 steps:", losses)
print("loss scale:", scales, "good steps:", tr.scaler.good_steps())
finite = all(bool(torch.isfinite(p).all()) for p in model.parameters())
print("all parameters finite:", finite)
