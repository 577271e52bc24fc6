"""Which torch (aten) ops does one edit step launch?  torch.profiler over 3 steps; prints ops by count with input shapes."""
import sys, os, copy, torch
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
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    for i in range(3):
        tr.train_step(view(i))
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="count", row_limit=70, max_name_column_width=40, max_shapes_column_width=70))
