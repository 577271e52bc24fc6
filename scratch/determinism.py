"""two identical 6-step trainings: are the parameters bit-identical?"""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from customnerf_amd import scene as sc, tcnn
from customnerf_amd.nerf.network_grid import NeRFNetwork
from customnerf_amd.nerf.provider_utils import generate_rays
from customnerf_amd.trainer import ReconTrainer
dev = torch.device("cuda", 0)
def train(cuda_ray):
    tcnn.set_default_dtype(torch.float16)
    torch.manual_seed(0)
    opt = sc.make_opt(cuda_ray=cuda_ray, fp16=True)
    model = NeRFNetwork(opt).to(dev)
    H = W = 128; V = 2
    c2w = torch.from_numpy(sc.poses(V)).to(dev)
    ro, rd = generate_rays(c2w, *sc.intrinsics(H, W), H, W, 1.0, 'nerfstudio')
    ro, rd = ro.view(V, 1, H * W, 3), rd.view(V, 1, H * W, 3)
    rgb, mask = sc.targets(V, H, W); rgb, mask = rgb.to(dev), mask.to(dev)
    if cuda_ray:
        from customnerf_amd import raymarching
        grid = torch.from_numpy(sc.sphere_density_grid(model.cascade, 128, opt.bound, 1.0, 100.0)).to(dev)
        model.density_grid.copy_(grid)
        model.density_bitfield = raymarching.packbits(model.density_grid, 10.0, model.density_bitfield)
    tr = ReconTrainer(model, opt, fp16=True)
    kw = dict(num_steps=opt.num_steps, upsample_steps=opt.upsample_steps, dt_gamma=0, max_steps=opt.max_steps)
    for i in range(6):
        tr.train_step(ro[i % V], rd[i % V], rgb[i % V], mask[i % V], **kw)
    torch.cuda.synchronize()
    return [hashlib.sha1(p.detach().cpu().numpy().tobytes()).hexdigest()[:12] for p in model.parameters()]
for cr in (False, True):
    a, b = train(cr), train(cr)
    print("cuda_ray", cr, "bit-identical:", a == b, a if a != b else "", b if a != b else "")
