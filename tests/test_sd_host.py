"""CPU: host-side logic of the SDS half — architecture tables (diffusers vocabulary, known SD-1.5 parameter counts), the GEMM
descriptor's C layout, weight packing identities, scheduler constants, and the CPU oracle's own sanity.  No GPU compute."""
import ctypes
import os
import subprocess
import tempfile

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sd15_architecture_tables():
    from customnerf_amd.sd import arch
    up = arch.unet_params(arch.UNET_SD15)
    assert arch.count(up) == 859_520_964                       # the published size of the SD-1.5 UNet
    assert arch.count(arch.vae_encoder_params(arch.VAE_SD15)) == 34_163_592 + 72   # AutoencoderKL encoder + quant_conv
    names = dict(up)
    assert len(names) == len(up) == 686                        # no duplicate keys; diffusers' key count for this model
    assert names["conv_in.weight"] == (320, 4, 3, 3) and names["time_embedding.linear_1.weight"] == (1280, 320)
    assert names["down_blocks.2.attentions.1.transformer_blocks.0.attn2.to_k.weight"] == (1280, 768)
    assert names["down_blocks.1.resnets.0.conv_shortcut.weight"] == (640, 320, 1, 1)
    assert names["up_blocks.1.resnets.2.conv1.weight"] == (1280, 1920, 3, 3)       # 1280 + 640 skip
    assert names["up_blocks.3.resnets.0.conv1.weight"] == (320, 960, 3, 3)
    assert names["up_blocks.2.upsamplers.0.conv.weight"] == (640, 640, 3, 3)
    assert names["mid_block.attentions.0.transformer_blocks.0.ff.net.0.proj.weight"] == (10240, 1280)
    assert "down_blocks.3.attentions.0.norm.weight" not in names and "up_blocks.0.attentions.0.norm.weight" not in names
    vn = dict(arch.vae_encoder_params(arch.VAE_SD15))
    assert vn["encoder.conv_out.weight"] == (8, 512, 3, 3) and vn["quant_conv.weight"] == (8, 8, 1, 1)
    assert vn["encoder.mid_block.attentions.0.to_q.weight"] == (512, 512)
    assert "encoder.down_blocks.3.downsamplers.0.conv.weight" not in vn
    a = arch.alphas_cumprod()
    assert a.shape == (1000,) and abs(float(a[0]) - 0.99915) < 1e-5 and abs(float(a[999]) - 0.0046604) < 1e-5   # scaled_linear 0.00085..0.012


def test_gemm_descriptor_layout_matches_the_header():
    from customnerf_amd._lib import SdGemmDesc
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "customnerf_sd.h"
int main(void) {
    printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(CnerfSdGemm), offsetof(CnerfSdGemm, M), offsetof(CnerfSdGemm, alpha), offsetof(CnerfSdGemm, sa_o),
           offsetof(CnerfSdGemm, mode), offsetof(CnerfSdGemm, Cin), offsetof(CnerfSdGemm, tstride));
    return 0;
}'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(d, "t"), os.path.join(d, "t.c")])
        out = subprocess.check_output([os.path.join(d, "t")], text=True).split()
    got = [ctypes.sizeof(SdGemmDesc)] + [getattr(SdGemmDesc, f).offset for f in ("M", "alpha", "sa_o", "mode", "Cin", "tstride")]
    assert [int(x) for x in out] == got


def test_weight_packing_identities():
    """pack_conv: NHWC im2col GEMM == conv2d; pack_conv_dgrad: conv of dY with the packed weights == autograd's input gradient."""
    from customnerf_amd.sd import pack
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 16, 6, 5, generator=g, requires_grad=True)
    w = torch.randn(24, 16, 3, 3, generator=g)
    y = F.conv2d(x, w, padding=1)
    cols = F.unfold(x.detach(), 3, padding=1).view(1, 16, 9, -1).permute(0, 3, 2, 1).reshape(30, 9 * 16)    # (kh, kw, ci) K order
    np.testing.assert_allclose((cols @ pack.pack_conv(w).float().t()).numpy(), y.detach().permute(0, 2, 3, 1).reshape(30, 24).numpy(), rtol=2e-2, atol=2e-2)
    dy = torch.randn_like(y)
    y.backward(dy)
    wd = pack.pack_conv_dgrad(w).float()                                                                     # [Cin, 9 * pad8(Cout)]
    dcols = F.unfold(dy, 3, padding=1).view(1, 24, 9, -1).permute(0, 3, 2, 1).reshape(30, 9 * 24)
    np.testing.assert_allclose((dcols @ wd.t()).numpy(), x.grad.permute(0, 2, 3, 1).reshape(30, 16).numpy(), rtol=3e-2, atol=5e-2)
    assert pack.pack_conv(torch.randn(8, 3, 3, 3)).shape == (8, 72)                                          # Cin padded to 8
    assert pack.pack_linear_T(torch.randn(5, 7)).shape == (7, 5)


def test_sd_oracle_sanity():
    from customnerf_amd.sd import arch
    from oracle import sd_oracle as so
    cfg = arch.UNET_TINY
    sd = arch.random_state_dict(arch.unet_params(cfg), 0)
    g = torch.Generator().manual_seed(0)
    x, ctx = torch.randn(2, 4, 8, 8, generator=g), torch.randn(2, 77, cfg["cross_attention_dim"], generator=g)
    with torch.no_grad():
        e = so.unet_forward(sd, cfg, x, torch.tensor([10.0, 10.0]), ctx)
        e2 = so.unet_forward(sd, cfg, x, torch.tensor([900.0, 900.0]), ctx)
    assert e.shape == x.shape and torch.isfinite(e).all() and float(e.std()) > 0.05 and float((e - e2).abs().max()) > 1e-3
    lat = torch.randn(1, 4, 8, 8, generator=g)
    noise = torch.randn(1, 4, 8, 8, generator=g)
    grad = so.sds_grad(sd, cfg, lat, ctx, 500, noise, arch.alphas_cumprod(), 7.5, 0.01)
    assert grad.shape == lat.shape and torch.isfinite(grad).all()
    vcfg = arch.VAE_TINY
    vsd = arch.random_state_dict(arch.vae_encoder_params(vcfg), 1)
    img = torch.rand(1, 3, 32, 32, generator=g, requires_grad=True)
    z = so.encode_imgs(vsd, vcfg, img, torch.zeros(1, 4, 4, 4))
    z.sum().backward()
    assert z.shape == (1, 4, 4, 4) and torch.isfinite(img.grad).all() and float(img.grad.abs().max()) > 0
    emb = so.timestep_embedding(torch.tensor([0.0, 1.0]), 8)
    np.testing.assert_allclose(emb[0].numpy(), [1, 1, 1, 1, 0, 0, 0, 0], atol=1e-7)                           # cos | sin, flip_sin_to_cos
    np.testing.assert_allclose(emb[1, 4].item(), np.sin(1.0), atol=1e-6)


def test_clip_vitb32_parameter_table_and_oracle_preprocess():
    """The OpenAI CLIP ViT-B/32 state dict (what nerf/clip.py loads) has 151,277,313 parameters; the oracle front-end is
    torchvision's Resize(224, bicubic) + CenterCrop + Normalize written with torch ops."""
    import torch
    from customnerf_amd.sd import clip_view as cv
    from oracle import sd_oracle as so
    n = sum(int(torch.Size(s).numel()) for _, s in cv.clip_params(cv.CLIP_VITB32))
    assert n == 151_277_313
    names = [k for k, _ in cv.clip_params(cv.CLIP_VITB32)]
    assert len(names) == len(set(names)) and "visual.transformer.resblocks.11.attn.in_proj_weight" in names and "transformer.resblocks.11.mlp.c_proj.bias" in names
    img = torch.full((1, 3, 40, 60), 0.5)
    out = so.clip_preprocess(img, 32)
    assert out.shape == (1, 3, 32, 32)
    exp = torch.tensor([(0.5 - m) / s for m, s in zip(cv.CLIP_MEAN, cv.CLIP_STD)]).view(1, 3, 1, 1).expand_as(out)
    assert torch.allclose(out, exp, atol=1e-5)                         # bicubic weights sum to one: a constant image stays constant
    sd = cv.random_clip_state_dict(cv.CLIP_TINY, 0)
    li, lt = so.clip_forward(sd, cv.CLIP_TINY, torch.randn(2, 3, 64, 64), torch.randint(1, 999, (3, 77)))
    assert li.shape == (2, 3) and torch.equal(li.t(), lt)
