"""How many scatter records w*g round to exactly zero in fp16 in the bench scene (they could be skipped exactly)."""
import sys, torch
sys.path.insert(0, '/root/repo')
from customnerf_amd import scene as sc, tcnn
from customnerf_amd.gridencoder import grid as ge
from customnerf_amd.nerf.network_grid import NeRFNetwork
from customnerf_amd.nerf.provider_utils import generate_rays
from customnerf_amd.trainer import ReconTrainer
tcnn.set_default_dtype(torch.float16)
torch.manual_seed(0)
opt = sc.make_opt(cuda_ray=False, fp16=True)
dev = torch.device('cuda')
model = NeRFNetwork(opt).to(dev)
H = W = 128; V = 8
c2w = torch.from_numpy(sc.poses(V)).to(dev)
rays_o, rays_d = generate_rays(c2w, *sc.intrinsics(H, W), H, W, 1.0, 'nerfstudio')
rays_o, rays_d = rays_o.view(V, 1, H * W, 3), rays_d.view(V, 1, H * W, 3)
rgb, mask = sc.targets(V, H, W); rgb, mask = rgb.to(dev), mask.to(dev)
tr = ReconTrainer(model, opt, fp16=True)
cap = {}
cls = ge._grid_encode
ob = cls.backward
def bw(ctx, grad):
    cap['g'] = grad.detach().clone()
    return ob(ctx, grad)
cls.backward = staticmethod(bw)
kw = dict(num_steps=opt.num_steps, upsample_steps=opt.upsample_steps, dt_gamma=0, max_steps=opt.max_steps)
for step in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    tr.train_step(rays_o[step % V], rays_d[step % V], rgb[step % V], mask[step % V], **kw)
    if step in (0, 5, 29, 99, 299):
        g = cap['g'].float()
        print('step', step, 'grad shape', tuple(cap['g'].shape), 'zero frac', float((g == 0).float().mean()), 'abs median', float(g.abs().median()), 'abs max', float(g.abs().max()),
              'rows all-zero', float((g.reshape(-1, g.shape[-1]) == 0).all(-1).float().mean()))
        # Monte-Carlo: w = product of three factors f or 1-f, f uniform; record = fp16(w*g)
        n = g.numel()
        idx = torch.randint(0, n, (4_000_000,), device=dev)
        gs = g.reshape(-1)[idx]
        f = torch.rand(4_000_000, 3, device=dev)
        z = 0.0
        for c in range(8):
            w = torch.ones_like(gs)
            for d in range(3):
                w = w * (f[:, d] if (c >> d) & 1 else 1 - f[:, d])
            z += float(((w * gs).half() == 0).float().mean()) / 8
        print('   est. zero-record fraction (per feature)', z)
