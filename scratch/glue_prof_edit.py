"""Torch-op census of one edit step (which aten ops still launch kernels around the library calls)."""
import os, sys, copy, torch, collections
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
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
for p in pre.parameters(): p.requires_grad_(False)
g = StableDiffusion(dev, '1.5', opt, seed=0)
H = W = 128
c2w = torch.from_numpy(sc.poses(8)).to(dev)
o, d = generate_rays(c2w, *sc.intrinsics(H, W), H, W, 1.0, 'nerfstudio')
o, d = o.view(8, 1, H * W, 3), d.view(8, 1, H * W, 3)
rgb, mask = sc.targets(8, H, W); rgb, mask = rgb.to(dev), mask.to(dev)
tr = EditTrainer(model, pre, g, opt, g.synthetic_text_embeds(0), g.synthetic_text_embeds(1), fp16=True)
for i in range(10): tr.train_step((rgb[i % 8], mask[i % 8], o[i % 8], d[i % 8], H, W, f"view{i % 8}"))
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tr.train_step((rgb[2], mask[2], o[2], d[2], H, W, "view2"))
    torch.cuda.synchronize()
skip = {'aten::empty', 'aten::view', 'aten::as_strided', 'aten::empty_like', 'aten::reshape', 'aten::empty_strided', 'aten::select', 'aten::slice', 'aten::detach', 'aten::alias', 'aten::_unsafe_view', 'aten::contiguous', 'aten::to', 'aten::lift_fresh', 'aten::unsqueeze', 'aten::expand', 'aten::t', 'aten::transpose', 'aten::permute', 'aten::resize_', 'aten::result_type', 'aten::squeeze', 'aten::narrow', 'aten::view_as', 'aten::unbind', 'aten::split', 'aten::split_with_sizes', 'aten::_reshape_alias', 'aten::expand_as', 'aten::flatten', 'aten::item', 'aten::_local_scalar_dense', 'aten::is_nonzero'}
c = collections.Counter(ev.name for ev in prof.events() if ev.device_type.name == 'CPU' and ev.name.startswith('aten::') and ev.name not in skip)
print(c.most_common(30))
k = [ev for ev in prof.events() if ev.device_type.name == 'CUDA']
tot = sum(ev.device_time for ev in k)
at = sum(ev.device_time for ev in k if 'at::native' in ev.name or 'rocclr' in ev.name)
print('gpu kernels', len(k), 'total us', tot, 'torch-native us', at)
