"""Where does the torch glue of one edit step come from?  torch.profiler with stacks: every aten op that launches a GPU kernel (copy_, fill_, cat, mul, ...)
is attributed to the innermost customnerf_amd / bench source line that issued it.  usage: python scratch/edit_glue.py"""
import collections, copy, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from customnerf_amd import scene as sc, tcnn
from customnerf_amd.nerf.network_grid import NeRFNetwork
from customnerf_amd.nerf.provider_utils import generate_rays
from customnerf_amd.sd import StableDiffusion
from customnerf_amd.sd.editing import EditTrainer
from torch.profiler import profile, ProfilerActivity
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
for i in range(2 * V + 2):
    tr.train_step(view(i))
torch.cuda.synchronize()
N = 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for i in range(N):
        tr.train_step(view(i))
    torch.cuda.synchronize()
acc = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith('aten::'):
        continue
    dt = sum(k.duration for k in ev.kernels) if ev.kernels else 0.0
    if not ev.kernels:
        continue
    where = "?"
    for fr in (ev.stack or []):
        if 'customnerf_amd' in fr or 'bench.py' in fr or 'scratch' in fr:
            where = fr.split('/root/repo/')[-1] if '/root/repo/' in fr else fr[-90:]
            break
    k = (ev.name, where)
    acc[k][0] += 1
    acc[k][1] += dt
rows = sorted(acc.items(), key=lambda kv: -kv[1][1])
tot_n = sum(v[0] for v in acc.values()); tot_t = sum(v[1] for v in acc.values())
print(f"aten ops with GPU kernels: {tot_n / N:.1f} per step, {tot_t / N:.1f} us of kernel time per step")
for (name, where), (n, t) in rows[:70]:
    print(f"{n / N:6.1f}/step {t / N:8.1f} us  {name:28s} {where}")
