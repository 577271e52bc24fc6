"""LGIE editing step: counterpart of Trainer_Nerf.train_step_editing / train_step_sd / get_pt
(nerf/utils_init_nerf.py:243-308, 353-394) on the MI355X renderer + SDS guidance.

One step = render the edited field (full view, fg/bg split) -> global or local SDS term (image or fg image, resized to 512,
VAE-encoded, UNet eps-prediction) -> background-preservation L1 against the cached render of the frozen pretrained field ->
backward through VAE and renderer -> fused Adam.  The per-view cache of the pretrained render stays on the GPU (the
reference parks it on the CPU and re-uploads it every step: utils_init_nerf.py:260-263)."""
import numpy as np
import torch
import torch.nn.functional as F

from ..optim import DynamicLossScaler, FusedAdam
from .edit_fn import RayImages, ScaledL1
from ..trainer import (CheckpointMixin, EvalMixin, allreduce_grads_flat, apply_optimizer_step, enable_grad_in_place, flat_grad_buffer, register_half_shadow,
                       setup_sharded_dp)


class EditTrainer(CheckpointMixin, EvalMixin):
    def __init__(self, model, model_pretrained, guidance, opt, text_z, text_z_fg, lr=None, fp16=True, world_size=1, loss_scale='dynamic', seed=0,
                 clip_guidance=None, clip_match_text=None, dp_mode='allreduce', init_scale=65536.0):
        """text_z / text_z_fg: the [2, 77, 768] (uncond, cond) embeddings of the global / local prompt; with `opt.clip_view` they are
        LISTS of three such tensors for the ", front view" / ", side view" / ", back view" prompts (prepare_text_embeddings,
        utils_init_nerf.py:318-336), `clip_guidance` is a customnerf_amd.sd.clip_view.CLIP and `clip_match_text` the token ids [3, 77] of
        the three view prompts (:344-345)."""
        self.model, self.model_pretrained, self.guidance, self.opt = model, model_pretrained, guidance, opt
        self.text_z, self.text_z_fg = text_z, text_z_fg
        self.clip_view = bool(getattr(opt, 'clip_view', False))
        self.clip_guidance, self.clip_match_text = clip_guidance, clip_match_text
        if self.clip_view:
            if clip_guidance is None or clip_match_text is None:
                raise ValueError("opt.clip_view needs clip_guidance and clip_match_text")
            if not (isinstance(text_z, (list, tuple)) and isinstance(text_z_fg, (list, tuple)) and len(text_z) == len(text_z_fg) == clip_match_text.shape[0]):
                raise ValueError("opt.clip_view needs one (text_z, text_z_fg) pair per view prompt")
        self.fp16, self.world_size = fp16, world_size
        # GradScaler policy, on device; init_scale = torch.cuda.amp.GradScaler's default unless the caller knows where the policy settles
        self.scaler = DynamicLossScaler(next(model.parameters()).device, init_scale=init_scale) if (fp16 and loss_scale == 'dynamic') else None
        self.loss_scale = 1.0 if (not fp16 or self.scaler is not None) else float(loss_scale)
        lr = opt.lr if lr is None else lr
        groups = model.get_params(lr)
        self.base_lrs = [g['lr'] for g in groups]
        self.optimizer = FusedAdam(groups, betas=(0.9, 0.99), eps=1e-15)
        self.optimizer.scaler = self.scaler
        register_half_shadow(self.optimizer, model, fp16)
        self.global_step = 0
        self.pt_dict = {}
        self._flat = flat_grad_buffer(self.model.parameters())          # .grad views of one flat buffer: the all-reduce runs in place
        enable_grad_in_place(model)
        self._dp = setup_sharded_dp(self, model, fp16) if (dp_mode == 'sharded' and world_size > 1) else None
        self._rng = np.random.RandomState(seed)
        self.replay = None            # tests: dict(branch='global'|'local', t=<timestep draw before t_ratio>, sample_noise=, noise=) replaces the step's SDS draws
        self.sds_resolution = int(getattr(opt, 'sds_resolution', 512))       # utils_init_nerf.py:303 resizes to 512 x 512
        guidance.set_system(self)
        self._render_kw = {k: v for k, v in vars(opt).items() if k != 'bg_color'}       # bg_color is passed explicitly (utils_init_nerf.py:365)

    def lr_factor(self):
        return 0.1 ** min(self.global_step / self.opt.iters, 1)

    def dp_describe(self):
        """one-line description of the gradient exchange (bench.py's config.parallelism)"""
        if self._dp is not None:
            return self._dp.describe()
        return "fp32 grad all-reduce, in place on the flat buffer"

    def _bg_color(self, rays_o, B, N):
        if getattr(self.opt, 'random_bg_c', False):
            return torch.rand((1, 3), device=rays_o.device).repeat(B * N, 1)
        if getattr(self.opt, 'black_bg_c', False):
            return torch.zeros((1, 3), device=rays_o.device).repeat(B * N, 1)
        if getattr(self.opt, 'white_bg_c', False):
            return torch.ones((1, 3), device=rays_o.device).repeat(B * N, 1)
        return None

    def get_pt(self, rays_o, rays_d, img_path, bg_color, B, H, W):
        """utils_init_nerf.py:243-267; with clip_view the pretrained render is matched against the view prompts once per view (:254-258)
        and the probabilities are cached with it (match_probs [B, 3], None otherwise)."""
        if img_path not in self.pt_dict:
            with torch.no_grad(), torch.autocast('cuda', dtype=torch.float16, enabled=self.fp16):
                out = self.model_pretrained.render(rays_o, rays_d, staged=False, perturb=True, bg_color=bg_color, force_all_rays=True, **self._render_kw)
            img = lambda t, c: t.reshape(B, H, W, c).permute(0, 3, 1, 2).contiguous().float().detach()
            match_probs = None
            if self.clip_view:
                match_probs = self.clip_guidance.match_view(img(out['image'], 3), self.clip_match_text)
            self.pt_dict[img_path] = (img(out['bg']['image'], 3), img(out['fg']['image'], 3), out['render_mask'].reshape(B, H, W, -1).float().detach(),
                                      img(out['fg']['depth'], 1), match_probs)
        pt_rgb_bg, pt_rgb_fg, pt_mask, pt_depth_fg, match_probs = self.pt_dict[img_path]
        return pt_rgb_fg, pt_rgb_bg, pt_mask, pt_depth_fg, match_probs

    def get_textz(self, B, match_probs=None):
        """utils_init_nerf.py:268-281: the direction-suffixed prompt whose view the CLIP match picked (one view per step: B == 1)"""
        if not self.clip_view:
            return self.text_z, self.text_z_fg
        if B != 1:
            raise ValueError("clip_view selects one prompt per step: B must be 1")
        if not hasattr(match_probs, '_cnerf_select'):
            match_probs._cnerf_select = int(match_probs.max(-1).indices[0])          # one host read per cached view, not per step
        t = match_probs._cnerf_select
        return self.text_z[t], self.text_z_fg[t]

    def train_step_sd(self, pred_rgb, pred_rgb_fg, match_probs=None):
        """utils_init_nerf.py:283-309: global (whole image, global prompt) or local (fg image, local prompt, scaled t) SDS term"""
        opt = self.opt
        text_z, text_z_fg = self.get_textz(pred_rgb.shape[0], match_probs)
        rp = self.replay or {}
        t_ratio = 1
        if getattr(opt, 'g_only', False):
            is_global = True
        elif getattr(opt, 'l_only', False):
            is_global = False
        elif 'branch' in rp:
            is_global = rp['branch'] == 'global'
        else:
            is_global = self._rng.random_sample() < opt.global_ratio
        if is_global:
            text_emb, img_rgb = text_z, pred_rgb
        else:
            text_emb, img_rgb, t_ratio = text_z_fg, pred_rgb_fg, opt.local_t_ratio
        r = self.sds_resolution
        sample_noise, noise = rp.get('sample_noise'), rp.get('noise')
        if sample_noise is None and noise is None:
            # the posterior-sample noise (sd.py:102) and the diffusion noise (sd.py:135) of the step: one generator launch for both
            lc = self.guidance.vae_cfg["latent_channels"]
            sample_noise, noise = torch.randn(2, img_rgb.shape[0], lc, r // 8, r // 8, device=img_rgb.device).unbind(0)
        latents = self.guidance.encode_imgs(img_rgb.float(), sample_noise=sample_noise, resize=(r, r))              # F.interpolate(..., (512, 512)) folded into the VAE front-end
        t_val = int(rp['t'] * t_ratio) if 't' in rp else None                                                     # sd.py:132
        return self.guidance.train_step(latents, text_emb, system=self, t_ratio=t_ratio, t_val=t_val, noise=noise)

    def train_step_editing(self, data):
        """utils_init_nerf.py:353-394.  data = (rgbs, mask, rays_o, rays_d, H, W, img_path)"""
        rgbs, mask, rays_o, rays_d, H, W, img_path = data
        B, N = rays_o.shape[:2]
        opt = self.opt
        bg_color = self._bg_color(rays_o, B, N)
        with torch.autocast('cuda', dtype=torch.float16, enabled=self.fp16):
            outputs = self.model.render(rays_o, rays_d, staged=False, perturb=True, force_all_rays=True, bg_color=bg_color, **self._render_kw)
        pred_rgb, pred_rgb_fg, pred_rgb_bg = self._pred_images(outputs, B, H, W)
        pred_ws = outputs['weights_sum'].reshape(B, H, W)
        pt_rgb_fg, pt_rgb_bg, pt_mask, pt_depth_fg, match_probs = self.get_pt(rays_o, rays_d, img_path, bg_color, B, H, W)
        if getattr(opt, 'ori_bg', False):
            non_edit = (pt_mask + outputs['render_mask'].reshape(B, H, W, -1)) < 0.5
            non_edit = non_edit.permute(0, 3, 1, 2)
            pt_rgb_bg = rgbs.reshape(B, H, W, 3).permute(0, 3, 1, 2) * non_edit + (~non_edit) * pt_rgb_bg
        loss, loss_dict = 0.0, {}
        if opt.lambda_sd:
            loss, loss_dict = self.train_step_sd(pred_rgb, pred_rgb_fg, match_probs)
        if opt.keep_bg:
            loss_bg = ScaledL1.apply(pt_rgb_bg, pred_rgb_bg, float(opt.keep_bg))            # opt.keep_bg * F.l1_loss(pt_rgb_bg, pred_rgb_bg): :389-391
            loss = loss + loss_bg
            loss_dict['loss_bg'] = loss_bg.detach()
        return pred_rgb, pred_ws, loss, loss_dict

    @staticmethod
    def _pred_images(outputs, B, H, W):
        """the whole / fg / bg renders as [B, 3, H, W] float32 images (utils_init_nerf.py:361-366).  The fused run() path hands over its raw
        composite buffer: one launch builds the three, one launch scatters their gradients back; other render paths go through torch views."""
        out_ray = outputs.get('_out_ray') if hasattr(outputs, 'get') else None
        if out_ray is not None and out_ray.is_cuda and out_ray.shape == (3, B * H * W, 6):
            return RayImages.apply(out_ray, B, H, W)
        img = lambda t: t.reshape(B, H, W, 3).permute(0, 3, 1, 2).contiguous().float()
        return img(outputs['image']), img(outputs['fg']['image']), img(outputs['bg']['image'])

    def train_step_editing_multi(self, views):
        """V camera views in one step (BASELINE.json north_star: the per-iteration batch shards "rays (and optionally SDS camera views)"): every
        view is rendered and gets its own background term exactly as in train_step_editing; the V images (or fg images — the global / local
        draw of utils_init_nerf.py:296 is made once per step) go through ONE VAE batch and ONE UNet batch of 2V, each with its own timestep
        and noise.  loss = mean over the views of the single-view loss."""
        opt = self.opt
        V = len(views)
        preds, preds_fg, losses_bg, match = [], [], [], None
        pred_rgb = pred_ws = None
        for data in views:
            rgbs, mask, rays_o, rays_d, H, W, img_path = data
            B, N = rays_o.shape[:2]
            bg_color = self._bg_color(rays_o, B, N)
            with torch.autocast('cuda', dtype=torch.float16, enabled=self.fp16):
                outputs = self.model.render(rays_o, rays_d, staged=False, perturb=True, force_all_rays=True, bg_color=bg_color, **self._render_kw)
            pred_rgb, pred_rgb_fg, pred_rgb_bg = self._pred_images(outputs, B, H, W)
            pred_ws = outputs['weights_sum'].reshape(B, H, W)
            pt_rgb_fg, pt_rgb_bg, pt_mask, pt_depth_fg, match = self.get_pt(rays_o, rays_d, img_path, bg_color, B, H, W)
            if getattr(opt, 'ori_bg', False):
                non_edit = ((pt_mask + outputs['render_mask'].reshape(B, H, W, -1)) < 0.5).permute(0, 3, 1, 2)
                pt_rgb_bg = rgbs.reshape(B, H, W, 3).permute(0, 3, 1, 2) * non_edit + (~non_edit) * pt_rgb_bg
            preds.append(pred_rgb)
            preds_fg.append(pred_rgb_fg)
            if opt.keep_bg:
                losses_bg.append(ScaledL1.apply(pt_rgb_bg, pred_rgb_bg, float(opt.keep_bg)))
        loss, loss_dict = 0.0, {}
        if opt.lambda_sd:
            loss, loss_dict = self.train_step_sd(torch.cat(preds, 0), torch.cat(preds_fg, 0), None if self.clip_view is False else match)
            loss = loss / V
        if opt.keep_bg:
            loss_bg = torch.stack(losses_bg).mean()
            loss = loss + loss_bg
            loss_dict['loss_bg'] = loss_bg.detach()
        return pred_rgb, pred_ws, loss, loss_dict

    def allreduce_grads(self):
        self._flat = allreduce_grads_flat(list(self.model.parameters()), self._flat, self.world_size)     # same path the gloo test exercises

    def train_step_multi(self, views):
        """one optimiser step on several camera views at once (train_step_editing_multi)"""
        return self.train_step(views, multi=True)

    def train_step(self, data, multi=False):
        """one optimiser step of the editing loop (train_one_epoch body, utils_init_nerf.py:599-629, with editing=True)"""
        from ..trainer import inf_check_is_folded, packed_weights_window
        self.model.train()
        self._inf_folded = inf_check_is_folded(self)                     # (round 6: the gradient producers raise found_inf themselves: trainer.inf_check_is_folded)
        if self.scaler is not None:
            self.scaler.watch(self._inf_folded)
        try:
            with packed_weights_window(self.model, self.opt):
                pred_rgb, pred_ws, loss, loss_dict = self.train_step_editing_multi(data) if multi else self.train_step_editing(data)
                if self.scaler is not None:
                    self.scaler.backward(loss)
                else:
                    (loss * self.loss_scale).backward()
            apply_optimizer_step(self)                                   # gradient exchange (world > 1) + scaler check + Adam + shadow hand-over
        finally:
            if self.scaler is not None:
                self.scaler.watch(False)
        self.global_step += 1
        return loss.detach(), loss_dict
