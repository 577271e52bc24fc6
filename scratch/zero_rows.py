"""VERDICT r4 item 2, 'measure first': fraction of samples whose output gradients are exactly zero IN THE FORM k_field_bwd_x2 CONSUMES THEM
(half(g_sigma * exp') == 0 and half(g_rgbc * sigmoid') == 0 for all four channels) on the benchmark's recon step — at random initialisation
(the headline's state), after 300 and 1000 fitting steps on the analytic sphere scene (variants.trained_field's state).
Conservative test (what a compaction pass can evaluate without the MLP's raw outputs): |g_sigma| * sigma < 2^-26 and |g_c| * s_c (1 - s_c) < 2^-26
(half rounds |v| <= 2^-25 to zero; the factor two absorbs last-bit differences between forward value and backward recompute).
usage: python scratch/zero_rows.py > gpurun_out/zero_rows.json"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from customnerf_amd import scene as sc, tcnn, field as fld          # noqa: E402
from customnerf_amd.nerf.network_grid import NeRFNetwork           # noqa: E402
from customnerf_amd.nerf.provider_utils import generate_rays       # noqa: E402
from customnerf_amd.trainer import ReconTrainer                    # noqa: E402

dev = torch.device("cuda:0")
tcnn.set_default_dtype(torch.float16)
torch.manual_seed(0)
opt = sc.make_opt(fp16=True)
model = NeRFNetwork(opt).to(dev)
H = W = 128
V = 8
c2w = torch.from_numpy(sc.poses(V)).to(dev)
ro, rd = generate_rays(c2w, *sc.intrinsics(H, W), H, W, 1.0, 'nerfstudio')
ro, rd = ro.view(V, 1, H * W, 3), rd.view(V, 1, H * W, 3)
rgb, mask = sc.targets(V, H, W)
rgb, mask = rgb.to(dev), mask.to(dev)
trainer = ReconTrainer(model, opt, fp16=True)
kw = dict(num_steps=opt.num_steps, upsample_steps=opt.upsample_steps, dt_gamma=0, max_steps=opt.max_steps)

cap = {}
orig_fwd, orig_bwd = fld.FieldAttach.forward, fld.FieldFunction.backward


def fwd(ctx, enc, xyz, dirs, dir_group, enc_dim, ng, nr, p_net, p_den, p_rgb, gip, sigma, rgbc):
    cap['sigma'], cap['rgbc'] = sigma.detach(), rgbc.detach()
    return orig_fwd(ctx, enc, xyz, dirs, dir_group, enc_dim, ng, nr, p_net, p_den, p_rgb, gip, sigma, rgbc)


def bwd(ctx, g_sigma, g_rgbc):
    if cap.get('on'):
        cap['gs'], cap['gc'] = g_sigma.detach().float().clone(), g_rgbc.detach().float().clone()
    return orig_bwd(ctx, g_sigma, g_rgbc)


fld.FieldAttach.forward = staticmethod(fwd)
fld.FieldFunction.backward = staticmethod(bwd)


def measure(tag, n=4):
    out = []
    for v in range(n):
        cap['on'] = True
        trainer.train_step(ro[v], rd[v], rgb_t[v], mask_t[v], **kw)
        cap['on'] = False
        gs, gc, sig, c = cap['gs'], cap['gc'], cap['sigma'].float(), cap['rgbc'].float()
        thr = 2.0 ** -26
        sig_c = sig.clamp(3.0590232e-07, 3269017.4)               # exp(clamp(., -15, 15))
        z_sigma = gs.abs() * sig_c < thr
        z_col = ((gc.abs() * (c * (1 - c))) < thr).all(-1)
        z_plain = (gs == 0) & (gc == 0).all(-1)
        zero = z_sigma & z_col
        N = ro[v].shape[1]
        per_ray = zero.view(2, N, -1)                             # [coarse | fine] blocks
        out.append({"view": v, "zero_frac": float(zero.float().mean()), "zero_frac_exact_fp32_zero": float(z_plain.float().mean()),
                    "zero_sigma_only": float(z_sigma.float().mean()), "zero_colour_only": float(z_col.float().mean()),
                    "zero_frac_coarse": float(per_ray[0].float().mean()), "zero_frac_fine": float(per_ray[1].float().mean()),
                    "rays_all_zero": float(per_ray.all(0).all(-1).float().mean()),
                    "loss_scale": trainer.scaler.get_scale()})
    return {"state": tag, "views": out, "mean_zero_frac": sum(o["zero_frac"] for o in out) / len(out)}


res = []
rgb_t, mask_t = rgb, mask
res.append(measure("random init, benchmark targets (the headline's state)"))
rgb_t, mask_t = sc.sphere_targets(ro.reshape(V, -1, 3), rd.reshape(V, -1, 3))
done = 0
for target in (300, 1000, 3000):
    for i in range(done, target):
        trainer.train_step(ro[i % V], rd[i % V], rgb_t[i % V], mask_t[i % V], **kw)
    done = target
    res.append(measure(f"after {target} fitting steps on the analytic sphere scene"))
print(json.dumps(res, indent=1))
