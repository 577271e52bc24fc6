"""Architecture tables of the two networks on the SDS path, in diffusers' state-dict vocabulary.

The reference loads `runwayml/stable-diffusion-v1-5` through diffusers (nerf/sd.py:48-54): UNet2DConditionModel and
AutoencoderKL.  Neither the library nor the weights are available offline, so this module states the public SD-1.5
architecture as a list of (parameter name, shape) in diffusers' naming — a real checkpoint's `unet` / `vae` state dict
loads by key — and builds seeded random state dicts of exactly those shapes for tests and benchmarks.
"""
import math

import torch

UNET_SD15 = dict(in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, heads=8,
                 cross_attention_dim=768, attn_blocks=(True, True, True, False), groups=32, eps=1e-5)
VAE_SD15 = dict(in_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2, groups=32, eps=1e-6,
                scaling_factor=0.18215)


# structurally identical small networks for CPU-oracle-sized tests (GroupNorm(32) needs >= 4 channels per group)
UNET_TINY = dict(UNET_SD15, block_out_channels=(128, 256, 384, 384), cross_attention_dim=96)
VAE_TINY = dict(VAE_SD15, block_out_channels=(128, 128, 256, 256))


def _resnet(p, cin, cout, temb):
    out = [(p + "norm1.weight", (cin,)), (p + "norm1.bias", (cin,)), (p + "conv1.weight", (cout, cin, 3, 3)), (p + "conv1.bias", (cout,))]
    if temb:
        out += [(p + "time_emb_proj.weight", (cout, temb)), (p + "time_emb_proj.bias", (cout,))]
    out += [(p + "norm2.weight", (cout,)), (p + "norm2.bias", (cout,)), (p + "conv2.weight", (cout, cout, 3, 3)), (p + "conv2.bias", (cout,))]
    if cin != cout:
        out += [(p + "conv_shortcut.weight", (cout, cin, 1, 1)), (p + "conv_shortcut.bias", (cout,))]
    return out


def _transformer(p, c, ctx):
    t = p + "transformer_blocks.0."
    out = [(p + "norm.weight", (c,)), (p + "norm.bias", (c,)), (p + "proj_in.weight", (c, c, 1, 1)), (p + "proj_in.bias", (c,))]
    for n in ("norm1", "norm2", "norm3"):
        out += [(t + n + ".weight", (c,)), (t + n + ".bias", (c,))]
    out += [(t + "attn1.to_q.weight", (c, c)), (t + "attn1.to_k.weight", (c, c)), (t + "attn1.to_v.weight", (c, c)),
            (t + "attn1.to_out.0.weight", (c, c)), (t + "attn1.to_out.0.bias", (c,)),
            (t + "attn2.to_q.weight", (c, c)), (t + "attn2.to_k.weight", (c, ctx)), (t + "attn2.to_v.weight", (c, ctx)),
            (t + "attn2.to_out.0.weight", (c, c)), (t + "attn2.to_out.0.bias", (c,)),
            (t + "ff.net.0.proj.weight", (8 * c, c)), (t + "ff.net.0.proj.bias", (8 * c,)), (t + "ff.net.2.weight", (c, 4 * c)), (t + "ff.net.2.bias", (c,))]
    out += [(p + "proj_out.weight", (c, c, 1, 1)), (p + "proj_out.bias", (c,))]
    return out


def unet_up_channels(cfg):
    """(resnet_in, skip, out) channel triples of the up path, per block (diffusers get_up_block wiring)."""
    boc = cfg["block_out_channels"]
    rev = tuple(reversed(boc))
    n = len(boc)
    L = cfg["layers_per_block"] + 1
    blocks = []
    prev = rev[0]
    for i in range(n):
        out_c = rev[i]
        in_c = rev[min(i + 1, n - 1)]
        res = []
        for j in range(L):
            skip = in_c if j == L - 1 else out_c
            rin = prev if j == 0 else out_c
            res.append((rin, skip, out_c))
        blocks.append(res)
        prev = out_c
    return blocks


def unet_params(cfg):
    boc = cfg["block_out_channels"]
    temb = boc[0] * 4
    ctx = cfg["cross_attention_dim"]
    out = [("conv_in.weight", (boc[0], cfg["in_channels"], 3, 3)), ("conv_in.bias", (boc[0],)),
           ("time_embedding.linear_1.weight", (temb, boc[0])), ("time_embedding.linear_1.bias", (temb,)),
           ("time_embedding.linear_2.weight", (temb, temb)), ("time_embedding.linear_2.bias", (temb,))]
    cin = boc[0]
    for i, c in enumerate(boc):
        for j in range(cfg["layers_per_block"]):
            out += _resnet(f"down_blocks.{i}.resnets.{j}.", cin, c, temb)
            if cfg["attn_blocks"][i]:
                out += _transformer(f"down_blocks.{i}.attentions.{j}.", c, ctx)
            cin = c
        if i < len(boc) - 1:
            out += [(f"down_blocks.{i}.downsamplers.0.conv.weight", (c, c, 3, 3)), (f"down_blocks.{i}.downsamplers.0.conv.bias", (c,))]
    c = boc[-1]
    out += _resnet("mid_block.resnets.0.", c, c, temb) + _transformer("mid_block.attentions.0.", c, ctx) + _resnet("mid_block.resnets.1.", c, c, temb)
    attn_up = tuple(reversed(cfg["attn_blocks"]))
    for i, res in enumerate(unet_up_channels(cfg)):
        for j, (rin, skip, oc) in enumerate(res):
            out += _resnet(f"up_blocks.{i}.resnets.{j}.", rin + skip, oc, temb)
            if attn_up[i]:
                out += _transformer(f"up_blocks.{i}.attentions.{j}.", oc, ctx)
        if i < len(boc) - 1:
            oc = res[0][2]
            out += [(f"up_blocks.{i}.upsamplers.0.conv.weight", (oc, oc, 3, 3)), (f"up_blocks.{i}.upsamplers.0.conv.bias", (oc,))]
    out += [("conv_norm_out.weight", (boc[0],)), ("conv_norm_out.bias", (boc[0],)), ("conv_out.weight", (cfg["out_channels"], boc[0], 3, 3)),
            ("conv_out.bias", (cfg["out_channels"],))]
    return out


def vae_encoder_params(cfg):
    boc = cfg["block_out_channels"]
    out = [("encoder.conv_in.weight", (boc[0], cfg["in_channels"], 3, 3)), ("encoder.conv_in.bias", (boc[0],))]
    cin = boc[0]
    for i, c in enumerate(boc):
        for j in range(cfg["layers_per_block"]):
            out += _resnet(f"encoder.down_blocks.{i}.resnets.{j}.", cin, c, 0)
            cin = c
        if i < len(boc) - 1:
            out += [(f"encoder.down_blocks.{i}.downsamplers.0.conv.weight", (c, c, 3, 3)), (f"encoder.down_blocks.{i}.downsamplers.0.conv.bias", (c,))]
    c = boc[-1]
    a = "encoder.mid_block.attentions.0."
    out += _resnet("encoder.mid_block.resnets.0.", c, c, 0)
    out += [(a + "group_norm.weight", (c,)), (a + "group_norm.bias", (c,))]
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        out += [(a + n + ".weight", (c, c)), (a + n + ".bias", (c,))]
    out += _resnet("encoder.mid_block.resnets.1.", c, c, 0)
    lc = 2 * cfg["latent_channels"]
    out += [("encoder.conv_norm_out.weight", (c,)), ("encoder.conv_norm_out.bias", (c,)), ("encoder.conv_out.weight", (lc, c, 3, 3)), ("encoder.conv_out.bias", (lc,)),
            ("quant_conv.weight", (lc, lc, 1, 1)), ("quant_conv.bias", (lc,))]
    return out


# older diffusers checkpoints name the VAE attention projections differently
VAE_ATTN_ALIASES = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


def random_state_dict(params, seed, device="cpu", dtype=torch.float32):
    """Seeded random weights of the given (name, shape) table: weights N(0, 1/fan_in) (activations stay O(1) through the
    depth, so float16 evaluation is meaningful), norm scales ~ 1, biases small."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd = {}
    for name, shape in params:
        if name.endswith(".bias"):
            t = torch.randn(shape, generator=g) * 0.02
        elif len(shape) == 1:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = torch.randn(shape, generator=g) / math.sqrt(fan_in)
        sd[name] = t.to(device=device, dtype=dtype)
    return sd


def count(params):
    n = 0
    for _, shape in params:
        k = 1
        for s in shape:
            k *= s
        n += k
    return n


def alphas_cumprod(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
    """SD-1.5 scheduler config (scheduler_config.json: scaled_linear betas); the reference reads `scheduler.alphas_cumprod`
    (sd.py:71)."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)
