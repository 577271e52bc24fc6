import sys, time; sys.path.insert(0,'.')
import torch
from customnerf_amd import scene as sc, tcnn
from customnerf_amd.nerf.network_grid import NeRFNetwork
from customnerf_amd.nerf.provider_utils import generate_rays
from customnerf_amd.trainer import ReconTrainer
tcnn.set_default_dtype(torch.float16)
opt = sc.make_opt(fp16=True)
model = NeRFNetwork(opt).cuda()
H=W=128; V=2
o,d = generate_rays(torch.from_numpy(sc.poses(V)).cuda(), *sc.intrinsics(H,W), H, W)
o=o.view(V,1,H*W,3); d=d.view(V,1,H*W,3)
rgb,mask = sc.targets(V,H,W); rgb=rgb.cuda(); mask=mask.cuda()
tr = ReconTrainer(model, opt, fp16=True)
kw=dict(num_steps=64, upsample_steps=64)
for i in range(2): tr.train_step(o[0], d[0], rgb[0], mask[0], **kw)
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("error")
import traceback
try:
    tr.train_step(o[0], d[0], rgb[0], mask[0], **kw)
    print("no sync found")
except Exception as e:
    traceback.print_exc()
