import sys, time; sys.path.insert(0,'.')
import torch
from customnerf_amd import scene as sc, tcnn
from customnerf_amd.nerf.network_grid import NeRFNetwork
from customnerf_amd.nerf.provider_utils import generate_rays
from customnerf_amd.trainer import ReconTrainer
tcnn.set_default_dtype(torch.float16)
torch.manual_seed(0)
opt = sc.make_opt(fp16=True)
model = NeRFNetwork(opt).cuda()
H=W=128; V=8
o,d = generate_rays(torch.from_numpy(sc.poses(V)).cuda(), *sc.intrinsics(H,W), H, W)
o=o.view(V,1,H*W,3); d=d.view(V,1,H*W,3)
rgb,mask = sc.targets(V,H,W); rgb=rgb.cuda(); mask=mask.cuda()
tr = ReconTrainer(model, opt, fp16=True)
kw=dict(num_steps=64, upsample_steps=64)
for i in range(5): tr.train_step(o[i%V], d[i%V], rgb[i%V], mask[i%V], **kw)
torch.cuda.synchronize()
n=20
t0=time.perf_counter()
for i in range(n): tr.train_step(o[i%V], d[i%V], rgb[i%V], mask[i%V], **kw)
t1=time.perf_counter()
torch.cuda.synchronize()
t2=time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/n:.2f} ms/step, total {1e3*(t2-t0)/n:.2f} ms/step")
# breakdown with profiler
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for i in range(5): tr.train_step(o[i%V], d[i%V], rgb[i%V], mask[i%V], **kw)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=25, max_name_column_width=50))
