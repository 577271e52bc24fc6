"""Where do the torch elementwise launches of one recon step come from?"""
import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from customnerf_amd import scene as sc, tcnn
from customnerf_amd.nerf.network_grid import NeRFNetwork
from customnerf_amd.nerf.provider_utils import generate_rays
from customnerf_amd.trainer import ReconTrainer
dev = torch.device('cuda')
tcnn.set_default_dtype(torch.float16)
torch.manual_seed(0)
opt = sc.make_opt(fp16=True)
model = NeRFNetwork(opt).to(dev)
H = W = 128
c2w = torch.from_numpy(sc.poses(8)).to(dev)
o, d = generate_rays(c2w, *sc.intrinsics(H, W), H, W, 1.0, 'nerfstudio')
o, d = o.view(8, 1, H * W, 3), d.view(8, 1, H * W, 3)
rgb, mask = sc.targets(8, H, W); rgb, mask = rgb.to(dev), mask.to(dev)
tr = ReconTrainer(model, opt, fp16=True)
kw = dict(num_steps=opt.num_steps, upsample_steps=opt.upsample_steps, dt_gamma=0, max_steps=opt.max_steps)
for i in range(3): tr.train_step(o[i], d[i], rgb[i], mask[i], **kw)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.train_step(o[3], d[3], rgb[3], mask[3], **kw)
    torch.cuda.synchronize()
for ev in prof.events():
    if ev.device_type.name == 'CPU' and ev.name.startswith('aten::') and ev.name not in ('aten::empty', 'aten::view', 'aten::as_strided', 'aten::empty_like', 'aten::reshape', 'aten::empty_strided', 'aten::select', 'aten::slice', 'aten::detach', 'aten::alias', 'aten::_unsafe_view', 'aten::contiguous', 'aten::to', 'aten::lift_fresh', 'aten::is_nonzero', 'aten::item', 'aten::_local_scalar_dense', 'aten::unsqueeze', 'aten::expand', 'aten::t', 'aten::transpose', 'aten::permute', 'aten::resize_', 'aten::result_type', 'aten::squeeze', 'aten::narrow', 'aten::view_as', 'aten::unbind', 'aten::split', 'aten::split_with_sizes'):
        st = [s for s in (ev.stack or []) if 'customnerf_amd' in s or 'glue_prof' in s]
        print(f"{ev.name:28s} shapes={[tuple(s) for s in (ev.input_shapes or [])][:3]} <- {st[:2]}")
