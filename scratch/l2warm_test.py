"""does a table-warming pass before the coarse gather make it as fast as the fine one? (hypothesis test, not a product path)"""
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
ro, rd = generate_rays(c2w, *sc.intrinsics(H, W), H, W, 1.0, 'nerfstudio')
ro, rd = ro.view(V, 1, H * W, 3), rd.view(V, 1, H * W, 3)
rgb, mask = sc.targets(V, H, W); rgb, mask = rgb.to(dev), mask.to(dev)
tr = ReconTrainer(model, opt, fp16=True)
kw = dict(num_steps=opt.num_steps, upsample_steps=opt.upsample_steps, dt_gamma=0, max_steps=opt.max_steps)
warm = os.environ.get("WARM", "0") == "1"
unit = torch.rand(1 << 20, 3, device=dev)
enc = torch.empty(16, 1 << 20, 2, dtype=torch.float16, device=dev)
for i in range(12):
    tr.train_step(ro[i % V], rd[i % V], rgb[i % V], mask[i % V], **kw)
    if warm:
        with torch.no_grad():
            model.pos_en.encode_into(unit, enc, 0, half=True)     # random points: touches the whole table once
torch.cuda.synchronize()
