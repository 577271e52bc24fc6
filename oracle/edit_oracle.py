"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product path).

The composed LGIE editing step of the reference's trainer, restated on the CPU from the other oracle pieces:
`Trainer_Nerf.train_step_editing` (nerf/utils_init_nerf.py:353-394) = render of the edited field (`torch_oracle.run`, renderer.py:278-405)
+ cached render of the frozen pretrained field (`get_pt`, utils_init_nerf.py:243-265) + the global / local SDS term (`train_step_sd`,
:282-308, on `sd_oracle.train_step_sd` = sd.py:97-155) + the background-preservation L1 (:388-391) with the `ori_bg` substitution (:378-380).

PARITY: the renderer half is pinned by the reference-produced golden vectors (tests/test_oracle_golden.py); the composition — which image and
prompt the SDS term takes, `local_t_ratio`, the `keep_bg` L1 — and the SDS arithmetic are pinned by tests/golden/editing.npz / sds.npz: the
reference's own train_step_editing / train_step run with a closed-form VAE and epsilon predictor (`encode_fn` / `eps_fn` here).  What stays
unpinned is the arithmetic INSIDE the UNet / VAE (third-party diffusers code, absent offline: DESIGN.md §2).  The `ori_bg` substitution follows
the evident intent of :378-380; as written the reference multiplies [B, 3, H, W] by [B, H, W, 1] and raises (recorded in editing.npz).
Every random draw of the step is an argument, so that the HIP path can be fed the same numbers.
"""
import torch
import torch.nn.functional as F

from . import sd_oracle as so
from . import torch_oracle as to


def train_step_editing(field, field_pretrained, rays_o, rays_d, rgbs, H, W, aabb, opt, vae_sd, vae_cfg, unet_sd, unet_cfg, text_z, text_z_fg,
                       alphas, draws, draws_pt, branch, t_draw, sample_noise, noise, size=(512, 512), encode_fn=None, eps_fn=None):
    """One editing step for one view (B = 1).  rays_o / rays_d [1, N, 3], rgbs [1, N, 3] (ground-truth colours, read by `ori_bg` only).
    draws / draws_pt: the run() draws of the edited / the pretrained render; branch: 'global' | 'local' (the outcome of the
    np.random.random() < global_ratio test of :296, or what g_only / l_only force); t_draw: the torch.randint timestep of sd.py:131 BEFORE
    the t_ratio scaling of :132; sample_noise / noise: the VAE posterior sample and the randn_like of sd.py:135.
    -> (loss, loss_dict, outputs) with autograd attached to `field`'s parameters."""
    B, N = rays_o.shape[:2]
    rkw = dict(num_steps=opt.num_steps, upsample_steps=opt.upsample_steps, perturb=True, training=True, train_conf=opt.train_conf,
               soft_mask=opt.soft_mask, conf_thr=opt.conf_thr, detach_bg=opt.detach_bg, detach_mask_from_field=opt.detach_mask_from_field)
    outputs = to.run(field, rays_o, rays_d, aabb, opt.min_near, draws=draws, **rkw)                              # :365
    img = lambda t: t.reshape(B, H, W, 3).permute(0, 3, 1, 2).contiguous()
    pred_rgb, pred_rgb_fg, pred_rgb_bg = img(outputs['image']), img(outputs['fg']['image']), img(outputs['bg']['image'])      # :367-374
    pred_mask = outputs['render_mask'].reshape(B, H, W, -1)
    with torch.no_grad():                                                                                           # get_pt, :243-251
        out_pt = to.run(field_pretrained, rays_o, rays_d, aabb, opt.min_near, draws=draws_pt, **rkw)
    pt_mask = out_pt['render_mask'].reshape(B, H, W, -1).detach()
    pt_rgb_bg = img(out_pt['bg']['image']).detach()
    if getattr(opt, 'ori_bg', False):                                                                               # :378-380
        non_edit = ((pt_mask + pred_mask) < 0.5).permute(0, 3, 1, 2)
        pt_rgb_bg = rgbs.reshape(B, H, W, 3).permute(0, 3, 1, 2) * non_edit + (~non_edit) * pt_rgb_bg
    loss, loss_dict = 0.0, {}
    if opt.lambda_sd:                                                                                               # :382-386 -> :282-308
        if branch == 'global':
            text_emb, img_rgb, t_ratio = text_z, pred_rgb, 1
        else:
            text_emb, img_rgb, t_ratio = text_z_fg, pred_rgb_fg, opt.local_t_ratio
        t = int(t_draw * t_ratio)                                                                                   # sd.py:132
        loss_sd, latents, grad = so.train_step_sd(vae_sd, vae_cfg, unet_sd, unet_cfg, img_rgb, text_emb, t, sample_noise, noise, alphas,
                                                  float(opt.cfg), float(opt.lambda_sd), size=size, encode_fn=encode_fn, eps_fn=eps_fn)
        loss = loss_sd
        loss_dict['loss_sds'] = loss_sd.detach()
        loss_dict['t'] = t
    if opt.keep_bg:                                                                                                 # :388-391
        loss_bg = opt.keep_bg * F.l1_loss(pt_rgb_bg, pred_rgb_bg)
        loss = loss + loss_bg
        loss_dict['loss_bg'] = loss_bg.detach()
    return loss, loss_dict, outputs
