"""`StableDiffusion` guidance with the reference's surface (nerf/sd.py:35-155) on libcustomnerf_hip.so.

Same constructor arguments, `get_text_embeds`, `encode_imgs`, `train_step(latents, text_embeddings, ..., t_ratio)` and
`set_system`; the VAE encoder and the UNet are customnerf_amd.sd.vae.VAEEncoder / unet.UNet.  Weights: pass diffusers
state dicts (`unet_state`, `vae_state`, float16/float32 tensors keyed as in a `runwayml/stable-diffusion-v1-5` checkpoint);
without them the networks are seeded-random of the SD-1.5 shapes (synthetic mode: benchmarks and tests — there is no
network access to fetch the checkpoint), and `data: synthetic` is what bench.py reports.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import arch, ops
from .edit_fn import SDSLoss
from .unet import UNet
from .vae import VAEEncoder


class StableDiffusion(nn.Module):
    def __init__(self, device, sd_version='1.5', opt=None, unet_state=None, vae_state=None, unet_cfg=None, vae_cfg=None, seed=0,
                 text_encoder=None, tokenizer=None, use_graph=True):
        super().__init__()
        self.device = torch.device(device)
        self.sd_version = sd_version
        self.opt = opt
        if sd_version != '1.5' and (unet_state is None or unet_cfg is None):
            raise ValueError("only the SD-1.5 architecture is built in; other versions need unet_cfg / vae_cfg and their state dicts")
        self.unet_cfg = unet_cfg or arch.UNET_SD15
        self.vae_cfg = vae_cfg or arch.VAE_SD15
        self.synthetic = unet_state is None or vae_state is None
        if unet_state is None:
            unet_state = arch.random_state_dict(arch.unet_params(self.unet_cfg), seed + 1)
        if vae_state is None:
            vae_state = arch.random_state_dict(arch.vae_encoder_params(self.vae_cfg), seed + 2)
        self.unet = UNet(self.unet_cfg, unet_state, self.device)
        self.vae = VAEEncoder(self.vae_cfg, vae_state, self.device)
        self.text_encoder, self.tokenizer = text_encoder, tokenizer
        self.num_train_timesteps = 1000
        self.min_step = int(self.num_train_timesteps * 0.02)                                # sd.py:69
        self.max_step = int(self.num_train_timesteps * getattr(opt, 'max_ratio', 0.98))     # sd.py:70
        self.alphas_host = arch.alphas_cumprod(self.num_train_timesteps)                    # scheduler.alphas_cumprod (sd.py:71)
        self.alphas = self.alphas_host.to(self.device)
        self.system = None
        self.use_graph = use_graph
        self._ctx_cache = {}
        self._gen = torch.Generator().manual_seed(seed)                                     # host-side timestep draws: no device sync

    # ---- text (sd.py:77-94)
    def get_text_embeds(self, prompt, negative_prompt):
        """`text_encoder` is customnerf_amd.sd.text_encoder.CLIPTextEncoder (the SD-1.5 CLIP text tower on the HIP library) built from the
        checkpoint's `text_encoder` state dict; `tokenizer` is transformers' CLIPTokenizer (BPE vocabulary files: host-side plumbing)."""
        if self.text_encoder is None or self.tokenizer is None:
            raise NotImplementedError(
                "the CLIP tokenizer vocabulary / text-encoder weights (runwayml/stable-diffusion-v1-5) are not available offline; pass "
                "`text_encoder=CLIPTextEncoder(...)` and `tokenizer=` or use synthetic_text_embeds()")
        def enc(p):
            ids = self.tokenizer(p, padding='max_length', max_length=self.tokenizer.model_max_length, truncation=True, return_tensors='pt').input_ids
            with torch.no_grad():
                return self.text_encoder(ids.to(self.device))[0]
        return torch.cat([enc(negative_prompt), enc(prompt)])

    def synthetic_text_embeds(self, seed=0):
        """[2, 77, D] (uncond, text) stand-in with CLIP-like statistics (unit-variance LayerNorm output)."""
        g = torch.Generator().manual_seed(1000 + seed)
        return torch.randn(2, 77, self.unet_cfg["cross_attention_dim"], generator=g).to(self.device)

    # ---- image -> latents (sd.py:97-105); differentiable w.r.t. imgs
    def encode_imgs(self, imgs, sample_noise=None, resize=None):
        B, _, H, W = imgs.shape
        resize = resize or (H, W)
        if sample_noise is None:
            sample_noise = torch.randn(B, self.vae_cfg["latent_channels"], resize[0] // 8, resize[1] // 8, device=imgs.device)
        return self.vae.encode_imgs(imgs, sample_noise, resize)

    def load_custom_diffusion(self, attn_procs_state, new_token_embedding=None):
        """sd.py:56-59 (`opt.use_cd`): Custom-Diffusion cross-attention weights into the UNet, and the `<new1>` textual-inversion
        embedding into the text encoder.  Returns (#replaced attention layers, new token id or None)."""
        n = self.unet.load_attn_procs(attn_procs_state)
        tid = None
        if new_token_embedding is not None and self.text_encoder is not None and hasattr(self.text_encoder, "add_token_embedding"):
            tid = self.text_encoder.add_token_embedding(new_token_embedding)
        return n, tid

    def get_params(self, lr):
        return []

    def set_system(self, system):
        self.system = system

    def draw_timestep(self, system=None, t_ratio=1):
        """sd.py:120-131"""
        min_step, max_step = self.min_step, self.max_step
        if getattr(self.opt, 'stage_time', False) and system is not None and system.global_step > self.opt.iters / 2:
            max_step = int(max_step * 0.5)
        t = int(torch.randint(min_step, max_step + 1, [1], generator=self._gen))
        return int(t * t_ratio)

    def _ctx_half(self, text_embeddings, pairs=1):
        """float16 copy of a text embedding (repeated `pairs` times for a multi-view batch), cached per source tensor (storage + version):
        the UNet keeps one captured graph and one set of cross-attention K / V^T per context tensor, so the same prompt must arrive as the
        same tensor every step."""
        key = (text_embeddings.data_ptr(), text_embeddings._version, tuple(text_embeddings.shape), text_embeddings.dtype, pairs)
        ent = self._ctx_cache.get(key)
        if ent is None:
            if len(self._ctx_cache) >= 8:
                self._ctx_cache.pop(next(iter(self._ctx_cache)))
            ctx = text_embeddings.to(self.device, torch.float16)
            if pairs > 1:
                ctx = ctx.repeat(pairs, 1, 1)                                                          # (uncond, text) x pairs
            ent = (text_embeddings, ctx.contiguous())                                                  # the source reference pins the key
            self._ctx_cache[key] = ent
        return ent[1]

    def eps_pred(self, unet_in, t, text_embeddings):
        """unet_in [2V, h, w, 8]; t: one timestep, or V of them (one per (uncond, text) pair); text_embeddings [2, 77, D] is shared by the pairs.
        With the graph, the timesteps reach its static input by kernel argument (cnerf_set_floats) and `unet_in` is copied only when it is not
        that graph's own input buffer (unet_input_buffer): the single-view SDS step replays with no staging tensor and no copy."""
        B = unet_in.shape[0]
        ts = [float(x) for x in t for _ in (0, 1)] if isinstance(t, (list, tuple)) else [float(t)] * B
        ctx = self._ctx_half(text_embeddings, B // 2)
        if self.use_graph and B <= 16:
            shape = tuple(unet_in.shape)
            sx, st = self.unet.graph_inputs(shape, ctx)
            if sx.data_ptr() != unet_in.data_ptr():
                sx.copy_(unet_in)
            ops.set_floats(st, ts)
            return self.unet.graphed(None, None, ctx, shape=shape)
        tt = torch.tensor(ts, dtype=torch.float32).to(self.device, non_blocking=True)
        return self.unet.graphed(unet_in, tt, ctx) if self.use_graph else self.unet(unet_in, tt, ctx)

    def unet_input_buffer(self, text_embeddings, shape):
        """the captured graph's own input tensor for this prompt and shape, once it exists (else None): cnerf_sd_add_noise writes into it directly"""
        if not self.use_graph or self.unet is None:
            return None
        ent = self.unet.graph_inputs(tuple(shape), self._ctx_half(text_embeddings, shape[0] // 2), create=False)
        return None if ent is None else ent[0]

    def sds_grad(self, latents, text_embeddings, t, noise):
        """sd.py:133-148 on device: add_noise, UNet on the CFG pair, `text + g (text - uncond)`, (1 - abar_t) weighting, nan_to_num.
        latents / noise [V, 4, h, w]: V views go through ONE UNet call of batch 2V (pairs laid out (uncond, text) per view); `t` is then an
        int (shared) or a list of V timesteps (BASELINE.json north_star: "optionally SDS camera views")."""
        V = latents.shape[0]
        if V == 1 and not isinstance(t, (list, tuple)):
            ab = float(self.alphas_host[t])
            _, _, h, w = latents.shape
            unet_in = ops.add_noise(latents.contiguous(), noise, ab, out=self.unet_input_buffer(text_embeddings, (2, h, w, 8)))
            eps = self.eps_pred(unet_in, t, text_embeddings)
            return ops.sds_grad(eps, noise, ab, float(self.opt.cfg), float(self.opt.lambda_sd))
        ts = list(t) if isinstance(t, (list, tuple)) else [int(t)] * V
        latents, noise = latents.contiguous(), noise.contiguous()
        _, _, h, w = latents.shape
        unet_in = torch.empty(2 * V, h, w, 8, dtype=torch.float16, device=latents.device)
        for v in range(V):
            ops.add_noise(latents[v:v + 1], noise[v:v + 1], float(self.alphas_host[ts[v]]), out=unet_in[2 * v:2 * v + 2])
        eps = self.eps_pred(unet_in, ts, text_embeddings)
        grad = torch.empty_like(noise)
        for v in range(V):
            ops.sds_grad(eps[2 * v:2 * v + 2], noise[v:v + 1], float(self.alphas_host[ts[v]]), float(self.opt.cfg), float(self.opt.lambda_sd), out=grad[v:v + 1])
        return grad

    def train_step(self, latents, text_embeddings, mask=None, img_path=None, tuning=False, gt_rgb=None, t_val=None, system=None, is_all=False,
                   camera=None, tuning_cls=False, t_ratio=1, noise=None):
        V = latents.shape[0]
        if t_val is not None:
            t = t_val if isinstance(t_val, (list, tuple)) else int(t_val)
        elif V == 1:
            t = self.draw_timestep(system, t_ratio)
        else:
            t = [self.draw_timestep(system, t_ratio) for _ in range(V)]              # every view of a multi-view step gets its own timestep
        with torch.no_grad():
            lat = latents.detach().float().contiguous()                          # NCHW, dense: the kernels below index it as raw memory
            if noise is None:
                noise = torch.randn(lat.shape, device=lat.device, dtype=torch.float32)
            noise = noise.contiguous()
            grad = self.sds_grad(lat, text_embeddings, t, noise)
        loss = SDSLoss.apply(latents if latents.dtype == torch.float32 else latents.float(), grad)                                          # 0.5 * mse(latents, (latents - grad).detach(), 'sum'): sd.py:150-152
        if getattr(self.opt, 'log_loss_item', True):
            return loss, dict(loss_sds=loss.item())
        return loss, dict(loss_sds=loss.detach())
