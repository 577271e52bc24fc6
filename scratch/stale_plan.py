"""VERDICT r4 item 5, 'control first': upper bound of what retiring the per-step exact histogram (k_bin2_hist x 2 + k_bin_scan_* on the side stream)
can buy — the recon step timed with a STALE scatter plan (the histogram / scan launches skipped, the emit re-using the previous step's counts)
against the normal step, alternating on one box.  The sample positions are frozen for the control (fixed jitter draws, learning rate 0: the field
and therefore the importance samples do not move), so the stale counts are the right counts — same main-stream work, correct sums, no histogram
(a plan that is really wrong overruns the bins' record ranges: memory fault).
usage: python scratch/stale_plan.py"""
import ctypes
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from customnerf_amd import scene as sc, tcnn                       # noqa: E402
from customnerf_amd.gridencoder import grid as ge                  # noqa: E402
from customnerf_amd.nerf.network_grid import NeRFNetwork           # noqa: E402
from customnerf_amd.nerf.provider_utils import generate_rays       # noqa: E402
from customnerf_amd.trainer import ReconTrainer                    # noqa: E402


class Proxy:
    def __init__(self, real):
        self._r, self.stale = real, False

    def __getattr__(self, n):
        f = getattr(self._r, n)
        if self.stale and n in ('cnerf_grid_encode_backward_prepare_rows', 'cnerf_grid_encode_backward_prepare_finish'):
            def fake(*a):
                ctypes.c_int.from_address(a[-2]).value = 1
                return 0
            return fake
        return f


ge.lib = Proxy(ge.lib)
dev = torch.device("cuda:0")
tcnn.set_default_dtype(torch.float16)
torch.manual_seed(0)
opt = sc.make_opt(fp16=True, lr=0.0)
model = NeRFNetwork(opt).to(dev)
H = W = 128
V = 8
c2w = torch.from_numpy(sc.poses(V)).to(dev)
ro, rd = generate_rays(c2w, *sc.intrinsics(H, W), H, W, 1.0, 'nerfstudio')
ro, rd = ro.view(V, 1, H * W, 3), rd.view(V, 1, H * W, 3)
rgb, mask = sc.targets(V, H, W)
rgb, mask = rgb.to(dev), mask.to(dev)
trainer = ReconTrainer(model, opt, fp16=True)
N = H * W
g = torch.Generator(device=dev).manual_seed(5)
draws = dict(z=torch.rand(N, opt.num_steps, device=dev, generator=g), u=torch.rand(N, opt.upsample_steps, device=dev, generator=g))
kw = dict(num_steps=opt.num_steps, upsample_steps=opt.upsample_steps, dt_gamma=0, max_steps=opt.max_steps, _draws=draws)
V = 1                                                    # one view: identical positions every step


def run(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        trainer.train_step(ro[i % V], rd[i % V], rgb[i % V], mask[i % V], **kw)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


run(10)
res = {"normal_ms": [], "stale_plan_ms": []}
for r in range(4):
    ge.lib.stale = False
    run(3)
    res["normal_ms"].append(run(40))
    ge.lib.stale = True
    run(3)
    res["stale_plan_ms"].append(run(40))
ge.lib.stale = False
res["gain_us_median"] = (sorted(res["normal_ms"])[1] + sorted(res["normal_ms"])[2] - sorted(res["stale_plan_ms"])[1] - sorted(res["stale_plan_ms"])[2]) * 500
print(json.dumps(res))
