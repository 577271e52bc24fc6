"""Host enqueue time vs wall time of the edit step.  usage: python scratch/edit_host.py  (run from the tree to measure)"""
import copy, os, sys, time, torch
sys.path.insert(0, os.getcwd())
from customnerf_amd import scene as sc, tcnn
from customnerf_amd.nerf.network_grid import NeRFNetwork
from customnerf_amd.nerf.provider_utils import generate_rays
from customnerf_amd.sd import StableDiffusion
from customnerf_amd.sd.editing import EditTrainer
dev = torch.device('cuda')
tcnn.set_default_dtype(torch.float16)
torch.manual_seed(0)
opt = sc.make_opt(cuda_ray=False, fp16=True, keep_bg=1000.0, lambda_sd=0.01, cfg=100.0, log_loss_item=False)
model = NeRFNetwork(opt).to(dev)
pre = copy.deepcopy(model).eval()
guidance = StableDiffusion(dev, '1.5', opt, seed=0)
H = W = 128; V = 8
c2w = torch.from_numpy(sc.poses(V)).to(dev)
o, d = generate_rays(c2w, *sc.intrinsics(H, W), H, W, 1.0, 'nerfstudio')
o, d = o.view(V, 1, H * W, 3), d.view(V, 1, H * W, 3)
rgb, mask = sc.targets(V, H, W); rgb, mask = rgb.to(dev), mask.to(dev)
tr = EditTrainer(model, pre, guidance, opt, guidance.synthetic_text_embeds(0), guidance.synthetic_text_embeds(1), fp16=True)
view = lambda j: (rgb[j % V], mask[j % V], o[j % V], d[j % V], H, W, f"view{j % V}")
for i in range(2 * V + 4):
    tr.train_step(view(i))
torch.cuda.synchronize()
for rep in range(3):
    N = 20
    host = []
    t0 = time.perf_counter()
    for i in range(N):
        h0 = time.perf_counter()
        tr.train_step(view(i))
        host.append(time.perf_counter() - h0)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / N
    host.sort()
    print(f"wall {wall * 1e3:.3f} ms/step; host enqueue median {host[N // 2] * 1e3:.3f} ms, min {host[0] * 1e3:.3f}, max {host[-1] * 1e3:.3f}", flush=True)
# one step in isolation: the host runs ahead of an idle GPU
torch.cuda.synchronize(); h0 = time.perf_counter(); tr.train_step(view(0)); h1 = time.perf_counter(); torch.cuda.synchronize(); h2 = time.perf_counter()
print(f"isolated step: host {1e3 * (h1 - h0):.3f} ms, until the GPU is done {1e3 * (h2 - h0):.3f} ms")
