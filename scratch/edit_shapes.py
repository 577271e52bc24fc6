"""Per-shape GEMM table of one eager edit step (event-bracketed launches): count, total / mean microseconds, TFLOP/s.
usage: python scratch/edit_shapes.py [sds_views]"""
import sys, os, collections, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from customnerf_amd import scene as sc, tcnn
from customnerf_amd.nerf.network_grid import NeRFNetwork
from customnerf_amd.nerf.provider_utils import generate_rays
from customnerf_amd.sd import StableDiffusion, ops as sdops
from customnerf_amd.sd.editing import EditTrainer
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device('cuda')
tcnn.set_default_dtype(torch.float16)
torch.manual_seed(0)
opt = sc.make_opt(cuda_ray=False, fp16=True, keep_bg=1000.0, lambda_sd=0.01, cfg=100.0, log_loss_item=False)
model = NeRFNetwork(opt).to(dev)
pre = copy.deepcopy(model).eval()
guidance = StableDiffusion(dev, '1.5', opt, seed=0, use_graph=False)
H = W = 128; V = 8
c2w = torch.from_numpy(sc.poses(V)).to(dev)
o, d = generate_rays(c2w, *sc.intrinsics(H, W), H, W, 1.0, 'nerfstudio')
o, d = o.view(V, 1, H * W, 3), d.view(V, 1, H * W, 3)
rgb, mask = sc.targets(V, H, W); rgb, mask = rgb.to(dev), mask.to(dev)
tr = EditTrainer(model, pre, guidance, opt, guidance.synthetic_text_embeds(0), guidance.synthetic_text_embeds(1), fp16=True)
view = lambda j: (rgb[j % V], mask[j % V], o[j % V], d[j % V], H, W, f"view{j % V}")
step = (lambda i: tr.train_step(view(i))) if nv == 1 else (lambda i: tr.train_step_multi([view(i * nv + k) for k in range(nv)]))
for i in range(3): step(i)
prof = []
sdops.set_profile(prof)
step(3)
sdops.set_profile(None)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for e0, e1, fl, shape in prof:
    a = agg.setdefault(shape, [0, 0.0, 0.0])
    a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e3; a[2] += fl
tot_us = sum(a[1] for a in agg.values()); tot_fl = sum(a[2] for a in agg.values())
print(f"views/step {nv}: {len(prof)} GEMM launches, {tot_us/1e3:.2f} ms, {tot_fl/1e12:.2f} TFLOP, {tot_fl/tot_us/1e6:.1f} TFLOP/s")
print(f"{'M':>7} {'N':>5} {'K':>6} mode batch  Hin  Cin ts up | count  total us  mean us  TFLOP/s  share")
for shape, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    M, N, K, mode, batch, Hin, Cin, ts, up = shape
    print(f"{M:7d} {N:5d} {K:6d} {mode:4d} {batch:5d} {Hin:4d} {Cin:4d} {ts:2d} {up:2d} | {a[0]:5d} {a[1]:9.1f} {a[1]/a[0]:8.1f} {a[2]/a[1]/1e6:8.1f} {100*a[1]/tot_us:6.1f}%")
