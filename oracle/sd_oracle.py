"""TEST INFRASTRUCTURE ONLY — CPU restatement (plain PyTorch, float32, NCHW) of the score-distillation half of the path.
Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only; never by the product.

PARITY UNPINNED: the arithmetic of these networks lives in `diffusers` (third-party, un-vendored, unpinned:
requirements.txt:24, call sites nerf/sd.py:54-65, 102, 136-140); neither the library nor the SD-1.5 weights exist offline and
the reference holds no test or golden vector for them.  This file restates the PUBLIC architecture of
UNet2DConditionModel / AutoencoderKL(encoder) for the SD-1.5 configuration module by module, in diffusers' state-dict
vocabulary, and the reference's own glue around it:
    encode_imgs   nerf/sd.py:97-105        (2x-1, VAE encode, posterior sample, x0.18215)
    train_step    nerf/sd.py:115-155       (t draw, add_noise, CFG with the reference's `text + g (text - uncond)`, SDS gradient)
    train_step_sd nerf/utils_init_nerf.py:286-309 (bilinear 512 resize in front of encode_imgs)
That glue IS pinned: tests/golden/sds.npz / editing.npz hold the outputs of the reference's own train_step / train_step_editing run with a
closed-form epsilon predictor and VAE (tests/golden/make_golden.py), and tests/test_oracle_golden.py replays them through `sds_grad` /
`train_step_sd` here (their `eps_fn` / `encode_fn` arguments).  Unpinned remains what happens INSIDE the UNet / VAE.
"""
import math

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------ building blocks
def _gn(x, sd, p, groups, eps):
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], eps)


def _conv(x, sd, p, stride=1, padding=1):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


def _lin(x, sd, p):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def resnet(x, temb, sd, p, groups, eps):
    """diffusers ResnetBlock2D: norm1 -> silu -> conv1 -> (+ time_emb_proj(silu(temb))) -> norm2 -> silu -> conv2 -> + shortcut(x)"""
    h = _conv(F.silu(_gn(x, sd, p + "norm1", groups, eps)), sd, p + "conv1")
    if temb is not None:
        h = h + _lin(F.silu(temb), sd, p + "time_emb_proj")[:, :, None, None]
    h = _conv(F.silu(_gn(h, sd, p + "norm2", groups, eps)), sd, p + "conv2")
    if (p + "conv_shortcut.weight") in sd:
        x = _conv(x, sd, p + "conv_shortcut", padding=0)
    return x + h


def _mha(q, k, v, heads):
    B, Tq, C = q.shape
    d = C // heads
    qh, kh, vh = (t.view(B, -1, heads, d).transpose(1, 2) for t in (q, k, v))
    w = torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(d), dim=-1)
    return (w @ vh).transpose(1, 2).reshape(B, Tq, C)


def transformer(x, ctx, sd, p, heads, groups):
    """diffusers Transformer2DModel (conv projections) with one BasicTransformerBlock (self-attn, cross-attn, GEGLU feed-forward)"""
    B, C, H, W = x.shape
    res = x
    h = _conv(_gn(x, sd, p + "norm", groups, 1e-6), sd, p + "proj_in", padding=0)
    h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    t = p + "transformer_blocks.0."
    n = F.layer_norm(h, (C,), sd[t + "norm1.weight"], sd[t + "norm1.bias"], 1e-5)
    h = h + _lin(_mha(_lin(n, sd, t + "attn1.to_q"), _lin(n, sd, t + "attn1.to_k"), _lin(n, sd, t + "attn1.to_v"), heads), sd, t + "attn1.to_out.0")
    n = F.layer_norm(h, (C,), sd[t + "norm2.weight"], sd[t + "norm2.bias"], 1e-5)
    h = h + _lin(_mha(_lin(n, sd, t + "attn2.to_q"), _lin(ctx, sd, t + "attn2.to_k"), _lin(ctx, sd, t + "attn2.to_v"), heads), sd, t + "attn2.to_out.0")
    n = F.layer_norm(h, (C,), sd[t + "norm3.weight"], sd[t + "norm3.bias"], 1e-5)
    a, gate = _lin(n, sd, t + "ff.net.0.proj").chunk(2, dim=-1)
    h = h + _lin(a * F.gelu(gate), sd, t + "ff.net.2")
    h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
    return _conv(h, sd, p + "proj_out", padding=0) + res


def timestep_embedding(t, dim):
    """diffusers get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0)"""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    arg = t.float()[:, None] * freqs[None]
    return torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1)


# ------------------------------------------------------------------------------------------------ UNet2DConditionModel forward
def unet_forward(sd, cfg, x, t, ctx):
    """x [B,4,h,w], t [B] (timesteps), ctx [B,77,cross_attention_dim] -> eps [B,4,h,w]"""
    boc, G, eps, heads = cfg["block_out_channels"], cfg["groups"], cfg["eps"], cfg["heads"]
    temb = timestep_embedding(t, boc[0])
    temb = _lin(F.silu(_lin(temb, sd, "time_embedding.linear_1")), sd, "time_embedding.linear_2")
    h = _conv(x, sd, "conv_in")
    skips = [h]
    for i in range(len(boc)):
        for j in range(cfg["layers_per_block"]):
            h = resnet(h, temb, sd, f"down_blocks.{i}.resnets.{j}.", G, eps)
            if cfg["attn_blocks"][i]:
                h = transformer(h, ctx, sd, f"down_blocks.{i}.attentions.{j}.", heads, G)
            skips.append(h)
        if i < len(boc) - 1:
            h = _conv(h, sd, f"down_blocks.{i}.downsamplers.0.conv", stride=2)
            skips.append(h)
    h = resnet(h, temb, sd, "mid_block.resnets.0.", G, eps)
    h = transformer(h, ctx, sd, "mid_block.attentions.0.", heads, G)
    h = resnet(h, temb, sd, "mid_block.resnets.1.", G, eps)
    attn_up = tuple(reversed(cfg["attn_blocks"]))
    for i in range(len(boc)):
        for j in range(cfg["layers_per_block"] + 1):
            h = torch.cat([h, skips.pop()], dim=1)
            h = resnet(h, temb, sd, f"up_blocks.{i}.resnets.{j}.", G, eps)
            if attn_up[i]:
                h = transformer(h, ctx, sd, f"up_blocks.{i}.attentions.{j}.", heads, G)
        if i < len(boc) - 1:
            h = _conv(F.interpolate(h, scale_factor=2.0, mode="nearest"), sd, f"up_blocks.{i}.upsamplers.0.conv")
    h = F.silu(_gn(h, sd, "conv_norm_out", G, eps))
    return _conv(h, sd, "conv_out")


# ------------------------------------------------------------------------------------------------ AutoencoderKL.encode
def vae_moments(sd, cfg, x):
    """x [B,3,H,W] in [-1,1] -> moments [B, 2*latent, H/8, W/8] (mean | logvar): diffusers Encoder + quant_conv"""
    boc, G, eps = cfg["block_out_channels"], cfg["groups"], cfg["eps"]
    h = _conv(x, sd, "encoder.conv_in")
    for i in range(len(boc)):
        for j in range(cfg["layers_per_block"]):
            h = resnet(h, None, sd, f"encoder.down_blocks.{i}.resnets.{j}.", G, eps)
        if i < len(boc) - 1:
            h = _conv(F.pad(h, (0, 1, 0, 1)), sd, f"encoder.down_blocks.{i}.downsamplers.0.conv", stride=2, padding=0)
    h = resnet(h, None, sd, "encoder.mid_block.resnets.0.", G, eps)
    a = "encoder.mid_block.attentions.0."
    B, C, H, W = h.shape
    n = _gn(h, sd, a + "group_norm", G, eps).permute(0, 2, 3, 1).reshape(B, H * W, C)
    o = _lin(_mha(_lin(n, sd, a + "to_q"), _lin(n, sd, a + "to_k"), _lin(n, sd, a + "to_v"), 1), sd, a + "to_out.0")
    h = h + o.reshape(B, H, W, C).permute(0, 3, 1, 2)
    h = resnet(h, None, sd, "encoder.mid_block.resnets.1.", G, eps)
    h = _conv(F.silu(_gn(h, sd, "encoder.conv_norm_out", G, eps)), sd, "encoder.conv_out")
    return _conv(h, sd, "quant_conv", padding=0)


def encode_imgs(sd, cfg, imgs, sample_noise):
    """nerf/sd.py:97-105 — imgs [B,3,H,W] in [0,1]; DiagonalGaussianDistribution.sample() with the given normal draw"""
    mean, logvar = vae_moments(sd, cfg, 2 * imgs - 1).chunk(2, dim=1)
    std = torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0))
    return (mean + std * sample_noise) * cfg["scaling_factor"]


# ------------------------------------------------------------------------------------------------ train_step (sd.py:115-155)
def sds_grad(unet_sd, unet_cfg, latents, text_embeddings, t, noise, alphas, guidance_scale, lambda_sd, eps_fn=None):
    """latents [1,4,h,w], text_embeddings [2,77,D] (uncond, text), integer t, noise ~ N(0,1) like latents.
    eps_fn(x [2,4,h,w], t [2], ctx) replaces the UNet (tests/golden/sds.npz: the reference's own train_step was run with a closed-form one)."""
    ab = alphas[t]
    noisy = ab.sqrt() * latents + (1 - ab).sqrt() * noise                                  # scheduler.add_noise
    x2, t2 = torch.cat([noisy] * 2), torch.full((2,), float(t))
    eps = eps_fn(x2, t2, text_embeddings) if eps_fn is not None else unet_forward(unet_sd, unet_cfg, x2, t2, text_embeddings)
    e_uncond, e_text = eps.chunk(2)
    e = e_text + guidance_scale * (e_text - e_uncond)                                      # sd.py:141 (the reference's form)
    return torch.nan_to_num((1 - ab) * (e - noise) * lambda_sd)


def train_step_sd(vae_sd, vae_cfg, unet_sd, unet_cfg, img_rgb, text_embeddings, t, sample_noise, noise, alphas, guidance_scale, lambda_sd, size=(512, 512),
                  encode_fn=None, eps_fn=None):
    """utils_init_nerf.py:303-308 + sd.py:97-155 for one view: img_rgb [1,3,H,W] in [0,1] (requires_grad for the image gradient).
    Returns (loss, latents, grad): loss = 0.5 * sum (latents - (latents - grad).detach())^2, so d loss / d latents = grad.
    encode_fn(imgs in [0,1]) / eps_fn replace the VAE / the UNet (tests/golden/editing.npz: the reference's own step was run with closed-form ones)."""
    pred_512 = F.interpolate(img_rgb, size, mode="bilinear", align_corners=False)
    latents = encode_fn(pred_512) if encode_fn is not None else encode_imgs(vae_sd, vae_cfg, pred_512, sample_noise)
    with torch.no_grad():
        grad = sds_grad(unet_sd, unet_cfg, latents.detach(), text_embeddings, t, noise, alphas, guidance_scale, lambda_sd, eps_fn=eps_fn)
    target = (latents - grad).detach()
    loss = 0.5 * F.mse_loss(latents, target, reduction="sum")
    return loss, latents, grad


# ------------------------------------------------------------------------------------------------ CLIP text encoder (get_text_embeds, sd.py:77-94)
def clip_text_forward(sd, cfg, input_ids):
    """transformers CLIPTextModel (ViT-L/14 text tower of SD-1.5): pre-LN layers, causal attention, quick-GELU; returns last_hidden_state."""
    B, T = input_ids.shape
    W, Hh = cfg["width"], cfg["heads"]
    h = sd["text_model.embeddings.token_embedding.weight"][input_ids] + sd["text_model.embeddings.position_embedding.weight"][:T][None]
    mask = torch.full((T, T), float("-inf")).triu(1)
    for i in range(cfg["layers"]):
        p = f"text_model.encoder.layers.{i}."
        n = F.layer_norm(h, (W,), sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"], cfg["eps"])
        q, k, v = (_lin(n, sd, p + f"self_attn.{x}_proj").view(B, T, Hh, W // Hh).transpose(1, 2) for x in ("q", "k", "v"))
        w = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(W // Hh) + mask, dim=-1)
        h = h + _lin((w @ v).transpose(1, 2).reshape(B, T, W), sd, p + "self_attn.out_proj")
        n = F.layer_norm(h, (W,), sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"], cfg["eps"])
        f = _lin(n, sd, p + "mlp.fc1")
        h = h + _lin(f * torch.sigmoid(1.702 * f), sd, p + "mlp.fc2")
    return F.layer_norm(h, (W,), sd["text_model.final_layer_norm.weight"], sd["text_model.final_layer_norm.bias"], cfg["eps"])


# ---- CLIP view classifier (nerf/clip.py, utils_init_nerf.py:254-258) ------------------------------------------------------------------------
# OpenAI `clip` is third-party and absent offline (PARITY UNPINNED like the rest of this file): restated from the published
# clip/model.py — VisionTransformer (conv1 patch embedding, class token, ln_pre, ResidualAttentionBlocks, ln_post, proj), the text
# transformer (causal mask, ln_final, EOT pooling, text_projection) and CLIP.forward's normalised, logit_scale.exp()-scaled logits.
def clip_preprocess(img, size=224, mean=(0.48145466, 0.4578275, 0.40821073), std=(0.26862954, 0.26130258, 0.27577711)):
    """torchvision T.Resize(size, BICUBIC, antialias=None) on a tensor (smaller edge -> size, F.interpolate bicubic without antialias),
    T.CenterCrop(size), T.Normalize — nerf/clip.py:13-17, written with torch ops only (torchvision is not needed)."""
    B, C, H, W = img.shape
    if H <= W:
        nh, nw = size, int(size * W / H)
    else:
        nh, nw = int(size * H / W), size
    x = F.interpolate(img.float(), size=(nh, nw), mode="bicubic", align_corners=False)
    top, left = int(round((nh - size) / 2.0)), int(round((nw - size) / 2.0))
    x = x[:, :, top:top + size, left:left + size]
    return (x - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)


def _clip_blocks(h, sd, prefix, layers, heads, eps, mask):
    n_, T, W = h.shape
    for i in range(layers):
        p = f"{prefix}resblocks.{i}."
        x = F.layer_norm(h, (W,), sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], eps)
        qkv = x @ sd[p + "attn.in_proj_weight"].t() + sd[p + "attn.in_proj_bias"]
        q, k, v = (t.view(n_, T, heads, W // heads).transpose(1, 2) for t in qkv.split(W, dim=-1))
        s = q @ k.transpose(-1, -2) / math.sqrt(W // heads)
        if mask is not None:
            s = s + mask
        a = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(n_, T, W)
        h = h + a @ sd[p + "attn.out_proj.weight"].t() + sd[p + "attn.out_proj.bias"]
        x = F.layer_norm(h, (W,), sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], eps)
        f = x @ sd[p + "mlp.c_fc.weight"].t() + sd[p + "mlp.c_fc.bias"]
        h = h + (f * torch.sigmoid(1.702 * f)) @ sd[p + "mlp.c_proj.weight"].t() + sd[p + "mlp.c_proj.bias"]
    return h


def clip_encode_image(sd, cfg, image):
    x = F.conv2d(image, sd["visual.conv1.weight"], stride=cfg["vision_patch_size"])
    B, W = x.shape[0], x.shape[1]
    x = x.reshape(B, W, -1).permute(0, 2, 1)
    x = torch.cat([sd["visual.class_embedding"].expand(B, 1, W), x], 1) + sd["visual.positional_embedding"]
    x = F.layer_norm(x, (W,), sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"], cfg["eps"])
    x = _clip_blocks(x, sd, "visual.transformer.", cfg["vision_layers"], cfg["vision_heads"], cfg["eps"], None)
    x = F.layer_norm(x[:, 0, :], (W,), sd["visual.ln_post.weight"], sd["visual.ln_post.bias"], cfg["eps"])
    return x @ sd["visual.proj"]


def clip_encode_text(sd, cfg, text):
    n_, T = text.shape
    W = cfg["transformer_width"]
    x = sd["token_embedding.weight"][text] + sd["positional_embedding"][:T]
    mask = torch.full((T, T), float("-inf")).triu(1)
    x = _clip_blocks(x, sd, "transformer.", cfg["transformer_layers"], cfg["transformer_heads"], cfg["eps"], mask)
    x = F.layer_norm(x, (W,), sd["ln_final.weight"], sd["ln_final.bias"], cfg["eps"])
    return x[torch.arange(n_), text.argmax(dim=-1)] @ sd["text_projection"]


def clip_forward(sd, cfg, image, text):
    """CLIP.forward: (logits_per_image [B, n], logits_per_text [n, B])"""
    i, t = clip_encode_image(sd, cfg, image), clip_encode_text(sd, cfg, text)
    i, t = i / i.norm(dim=1, keepdim=True), t / t.norm(dim=1, keepdim=True)
    li = sd["logit_scale"].exp() * i @ t.t()
    return li, li.t()
