"""host (enqueue) time per edit step and of the UNet graph replay alone"""
import os, sys, time, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from customnerf_amd import scene as sc, tcnn
from customnerf_amd.nerf.network_grid import NeRFNetwork
from customnerf_amd.nerf.provider_utils import generate_rays
from customnerf_amd.sd import StableDiffusion
from customnerf_amd.sd.editing import EditTrainer
dev = torch.device("cuda", 0)
tcnn.set_default_dtype(torch.float16)
torch.manual_seed(0)
opt = sc.make_opt(cuda_ray=False, fp16=True, keep_bg=1000.0, lambda_sd=0.01, cfg=100.0, log_loss_item=False)
model = NeRFNetwork(opt).to(dev)
pretrained = copy.deepcopy(model).eval()
for p in pretrained.parameters(): p.requires_grad_(False)
guidance = StableDiffusion(dev, '1.5', opt, seed=0)
H = W = 128; V = 8
c2w = torch.from_numpy(sc.poses(V)).to(dev)
rays_o, rays_d = generate_rays(c2w, *sc.intrinsics(H, W), H, W, 1.0, 'nerfstudio')
rays_o, rays_d = rays_o.view(V, 1, H * W, 3), rays_d.view(V, 1, H * W, 3)
rgb, mask = sc.targets(V, H, W); rgb, mask = rgb.to(dev), mask.to(dev)
trainer = EditTrainer(model, pretrained, guidance, opt, guidance.synthetic_text_embeds(0), guidance.synthetic_text_embeds(1), fp16=True, world_size=1)
marks = {}
orig = guidance.unet.graphed
def timed(*a, **k):
    t0 = time.perf_counter(); r = orig(*a, **k); marks.setdefault('replay', []).append(time.perf_counter() - t0); return r
guidance.unet.graphed = timed
def step(i):
    v = i % V
    return trainer.train_step((rgb[v], mask[v], rays_o[v], rays_d[v], H, W, f"view{v}"))
for i in range(10): step(i)
torch.cuda.synchronize()
marks.clear(); ts = []
for i in range(26):
    t0 = time.perf_counter(); step(10 + i); ts.append(time.perf_counter() - t0); torch.cuda.synchronize()
print("host enqueue ms per edit step:", [round(t * 1e3, 2) for t in ts])
print("  of which UNet graph replay:", [round(t * 1e3, 2) for t in marks['replay']])

# free-running: how long does the host loop take when nothing syncs between steps?
torch.cuda.synchronize()
t0 = time.perf_counter(); hs = []
for i in range(12):
    step(40 + i); hs.append(time.perf_counter() - t0)
t1 = time.perf_counter() - t0
torch.cuda.synchronize()
t2 = time.perf_counter() - t0
print("free-running 12 steps: host loop done at %.1f ms, GPU done at %.1f ms" % (t1 * 1e3, t2 * 1e3))
print("host step completion times:", [round(h * 1e3, 1) for h in hs])
