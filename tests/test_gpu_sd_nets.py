"""GPU parity of the SDS networks (customnerf_amd.sd: UNet eps-prediction, VAE encoder forward + input gradient, the SDS
train_step) against the float32 CPU restatement in oracle/sd_oracle.py, on seeded random weights of the SD-1.5 shapes.
PARITY UNPINNED w.r.t. diffusers: both sides restate the public architecture; what is pinned here is that the
HIP graph computes the same function as the plain-PyTorch statement, layer for layer, in float16 vs float32.
Tolerances: float16 activations through ~60 layers -> errors are reported relative to the output's own scale."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import sd_oracle as so      # noqa: E402


def rel_err(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12)), float((a - b).norm() / (b.norm() + 1e-12))


def half_sd(sd):
    """the product stores weights as float16: give the oracle the same (rounded) values"""
    return {k: (v.half().float() if v.dim() > 1 else v) for k, v in sd.items()}


def to_nhwc8(x):
    B, C, H, W = x.shape
    out = torch.zeros(B, H, W, 8, dtype=torch.float16)
    out[..., :C] = x.permute(0, 2, 3, 1).half()
    return out.cuda()


# ("sd15", 64) is the size the reference runs the UNet at: latents [2, 4, 64, 64] (nerf/sd.py:140)
@pytest.mark.parametrize("cfg_name,hw", [("tiny", 32), ("tiny", 24), ("sd15", 16), ("sd15", 64)])
def test_unet_forward(cfg_name, hw):
    from customnerf_amd.sd import arch
    from customnerf_amd.sd.unet import UNet
    cfg = arch.UNET_TINY if cfg_name == "tiny" else arch.UNET_SD15
    sd = half_sd(arch.random_state_dict(arch.unet_params(cfg), seed=3))
    g = torch.Generator().manual_seed(hw)
    x = torch.randn(2, 4, hw, hw, generator=g).half().float()
    ctx = torch.randn(2, 77, cfg["cross_attention_dim"], generator=g).half().float()
    t = torch.tensor([481.0, 481.0])
    with torch.no_grad():
        ref = so.unet_forward(sd, cfg, x, t, ctx)
        net = UNet(cfg, sd, "cuda")
        out = net(to_nhwc8(x), t.cuda(), ctx.half().cuda())
        out_g = net.graphed(to_nhwc8(x), t.cuda(), ctx.half().cuda()).clone()               # the graph's static output buffer: copy
        out_g2 = net.graphed(to_nhwc8(x * 0.5), t.cuda(), ctx.half().cuda()).clone()      # replay with new inputs
        ref2 = so.unet_forward(sd, cfg, x * 0.5, t, ctx) if cfg_name == "tiny" else None
    assert out.shape == (2, hw, hw, 4)
    emax, el2 = rel_err(out.permute(0, 3, 1, 2), ref)
    print(f"[unet {cfg_name}-{hw}] max rel {emax:.3e}, L2 rel {el2:.3e}")
    assert emax < 5e-3 and el2 < 5e-3, (emax, el2)      # measured on MI355X (round 4): 1.4e-3 .. 1.7e-3 at every size (float16 activations vs the float32 oracle)
    # the HIP-graph replay is the same computation, and since round 4 (fixed-point GroupNorm statistics) the same bits
    assert torch.equal(out_g, out)
    if ref2 is not None:
        assert rel_err(out_g2.permute(0, 3, 1, 2), ref2)[1] < 1e-2


# ("sd15", 512) is the size the reference encodes at (utils_init_nerf.py:303: F.interpolate(..., (512, 512)))
@pytest.mark.parametrize("cfg_name,size", [("tiny", 64), ("sd15", 128), ("sd15", 512)])
def test_vae_encode_forward_backward(cfg_name, size):
    from customnerf_amd.sd import arch
    from customnerf_amd.sd.vae import VAEEncoder
    cfg = arch.VAE_TINY if cfg_name == "tiny" else arch.VAE_SD15
    sd = half_sd(arch.random_state_dict(arch.vae_encoder_params(cfg), seed=5))
    g = torch.Generator().manual_seed(size)
    img = torch.rand(1, 3, 48, 40, generator=g)
    noise = torch.randn(1, 4, size // 8, size // 8, generator=g)
    dlat = torch.randn(1, 4, size // 8, size // 8, generator=g)
    img_ref = img.clone().requires_grad_(True)
    lat_ref = so.encode_imgs(sd, cfg, torch.nn.functional.interpolate(img_ref, (size, size), mode="bilinear", align_corners=False), noise)
    lat_ref.backward(dlat)
    vae = VAEEncoder(cfg, sd, "cuda")
    img_g = img.cuda().requires_grad_(True)
    lat = vae.encode_imgs(img_g, noise.cuda(), resize=(size, size))
    lat.backward(dlat.cuda(), retain_graph=(cfg_name == "tiny"))
    if cfg_name == "tiny":
        # a second backward through the same graph: the residual blocks' pooled scratch statistics were consumed by the first pass and must
        # not be reused (sd/vae.py::_ResnetFn) — the accumulated gradient is exactly twice the first one
        g1 = img_g.grad.clone()
        lat.backward(dlat.cuda())
        assert torch.equal(img_g.grad, 2 * g1)
        img_g.grad = g1
    assert lat.shape == lat_ref.shape and lat.dtype == torch.float32
    emax, el2 = rel_err(lat, lat_ref)
    print(f"[vae {cfg_name}-{size}] latents max rel {emax:.3e}, L2 rel {el2:.3e}")
    assert emax < 5e-3 and el2 < 3e-3, ("latents", emax, el2)      # measured: 0.7e-3 .. 1.3e-3 max, 0.6e-3 .. 0.7e-3 L2
    gmax, gl2 = rel_err(img_g.grad, img_ref.grad)
    print(f"[vae {cfg_name}-{size}] d latents / d image max rel {gmax:.3e}, L2 rel {gl2:.3e}")
    # measured (round 5, MI355X): max 1.9e-3 / 2.1e-3 / 2.4e-3, L2 2.0e-3 / 1.9e-3 / 1.9e-3 for the three cases; bound = 3 x the largest
    assert gmax < 7.5e-3 and gl2 < 6e-3, ("d latents / d image", gmax, gl2)


def test_sds_train_step_matches_oracle():
    """StableDiffusion.train_step + encode_imgs against oracle.train_step_sd: same t, same noise draws -> same SDS gradient on the
    latents and the same gradient on the rendered image."""
    from customnerf_amd.sd import arch
    from customnerf_amd.sd.guidance import StableDiffusion
    ucfg, vcfg = arch.UNET_TINY, arch.VAE_TINY
    usd = half_sd(arch.random_state_dict(arch.unet_params(ucfg), seed=11))
    vsd = half_sd(arch.random_state_dict(arch.vae_encoder_params(vcfg), seed=12))
    opt = types.SimpleNamespace(cfg=7.5, lambda_sd=0.01, max_ratio=0.98, stage_time=False, iters=1000, log_loss_item=True)
    guide = StableDiffusion("cuda", "1.5", opt, unet_state=usd, vae_state=vsd, unet_cfg=ucfg, vae_cfg=vcfg)
    assert not guide.synthetic and guide.min_step == 20 and guide.max_step == 980
    g = torch.Generator().manual_seed(2)
    img = torch.rand(1, 3, 32, 32, generator=g)
    text = torch.randn(2, 77, ucfg["cross_attention_dim"], generator=g).half().float()
    sample_noise = torch.randn(1, 4, 16, 16, generator=g)
    noise = torch.randn(1, 4, 16, 16, generator=g)
    t = 437
    img_ref = img.clone().requires_grad_(True)
    loss_ref, lat_ref, grad_ref = so.train_step_sd(vsd, vcfg, usd, ucfg, img_ref, text, t, sample_noise, noise, guide.alphas_host, opt.cfg, opt.lambda_sd,
                                                   size=(128, 128))
    loss_ref.backward()
    img_g = img.cuda().requires_grad_(True)
    lat = guide.encode_imgs(img_g, sample_noise.cuda(), resize=(128, 128))
    loss, ld = guide.train_step(lat, text.cuda(), t_val=t, noise=noise.cuda())
    loss.backward()
    assert isinstance(ld["loss_sds"], float)
    with torch.no_grad():
        grad = lat.detach() - (lat.detach() - guide.sds_grad(lat.detach(), text.cuda(), t, noise.cuda()))
    e_lat, e_grad, e_img = rel_err(lat, lat_ref)[1], rel_err(grad, grad_ref)[1], rel_err(img_g.grad, img_ref.grad)[1]
    e_loss = abs(float(loss) - float(loss_ref)) / float(loss_ref)
    print(f"[sds tiny] L2 rel: latents {e_lat:.3e}, SDS gradient {e_grad:.3e}, d loss / d image {e_img:.3e}; loss rel {e_loss:.3e}")
    # measured (round 5, MI355X): latents 5.7e-4, SDS gradient 4.2e-3, d loss / d image 4.3e-3, loss 6.7e-5; bounds = 3 x measured.
    # Round 6 (GroupNorm statistics from the split-K tail: the fixed-point partials are cut differently, every downstream rounding moves): loss 3.2e-4
    assert e_lat < 2e-3
    assert e_grad < 1.3e-2
    assert e_loss < 1e-3
    assert e_img < 1.3e-2
    # timestep draws follow sd.py:120-131
    ts = [guide.draw_timestep(None, 1) for _ in range(200)]
    assert min(ts) >= 20 and max(ts) <= 980
    assert all(t_ <= 980 * 0.5 + 1 for t_ in [guide.draw_timestep(None, 0.5) for _ in range(50)])
    with pytest.raises(NotImplementedError):
        guide.get_text_embeds(["a"], [""])


@pytest.mark.parametrize("cfg_name", ["tiny", "sd15"])
def test_clip_text_encoder(cfg_name):
    """get_text_embeds' text tower (causal attention, quick-GELU) against the oracle restatement of transformers' CLIPTextModel"""
    from customnerf_amd.sd import arch
    from customnerf_amd.sd import text_encoder as te
    cfg = te.CLIP_TEXT_TINY if cfg_name == "tiny" else te.CLIP_TEXT_SD15
    sd = half_sd(arch.random_state_dict(te.clip_text_params(cfg), seed=21))
    if cfg_name == "sd15":
        assert arch.count(te.clip_text_params(cfg)) == 123_060_480          # CLIP ViT-L/14 text model
    g = torch.Generator().manual_seed(4)
    ids = torch.randint(0, cfg["vocab_size"], (2, 77), generator=g)
    with torch.no_grad():
        ref = so.clip_text_forward(sd, cfg, ids)
        out = te.CLIPTextEncoder(cfg, sd, "cuda")(ids.cuda())[0]
    assert out.shape == (2, 77, cfg["width"])
    emax, el2 = rel_err(out, ref)
    print(f"[clip text {cfg_name}] max rel {emax:.3e}, L2 rel {el2:.3e}")
    # measured (round 5, MI355X): tiny 9.7e-4 max / 7.6e-4 L2, SD-1.5 shapes 1.9e-3 / 1.0e-3; bounds = 3 x the larger
    assert emax < 6e-3 and el2 < 3.5e-3, (emax, el2)
    # causality: changing a late token must not change earlier positions
    ids2 = ids.clone()
    ids2[:, 50] = (ids2[:, 50] + 1) % cfg["vocab_size"]
    with torch.no_grad():
        out2 = te.CLIPTextEncoder(cfg, sd, "cuda")(ids2.cuda())[0]
    assert torch.equal(out2[:, :50], out[:, :50]) and not torch.equal(out2[:, 50:], out[:, 50:])


def test_get_text_embeds_with_injected_tokenizer():
    """sd.py:77-94: [uncond ; text] order, max_length padding — with a stand-in tokenizer (the BPE vocabulary is not available offline)"""
    import types
    from customnerf_amd.sd import arch
    from customnerf_amd.sd import text_encoder as te
    from customnerf_amd.sd.guidance import StableDiffusion
    cfg = te.CLIP_TEXT_TINY
    sd = arch.random_state_dict(te.clip_text_params(cfg), seed=5)

    class Tok:
        model_max_length = 77

        def __call__(self, text, padding=None, max_length=None, truncation=None, return_tensors=None):
            ids = torch.zeros(len(text), max_length, dtype=torch.long)
            for i, t in enumerate(text):
                codes = [ord(c) % cfg["vocab_size"] for c in t][:max_length]
                ids[i, :len(codes)] = torch.tensor(codes, dtype=torch.long)
            return types.SimpleNamespace(input_ids=ids)

    opt = types.SimpleNamespace(cfg=7.5, lambda_sd=0.01, max_ratio=0.98, stage_time=False, iters=10, log_loss_item=False)
    ucfg, vcfg = arch.UNET_TINY, arch.VAE_TINY
    guide = StableDiffusion("cuda", "1.5", opt, unet_state=arch.random_state_dict(arch.unet_params(ucfg), 1), vae_state=arch.random_state_dict(arch.vae_encoder_params(vcfg), 2),
                            unet_cfg=ucfg, vae_cfg=vcfg, text_encoder=te.CLIPTextEncoder(cfg, sd, "cuda"), tokenizer=Tok())
    z = guide.get_text_embeds(["a corgi in a forest"], [""])
    assert z.shape == (2, 77, cfg["width"])
    with torch.no_grad():
        ref_text = so.clip_text_forward(half_sd(sd), cfg, Tok()(["a corgi in a forest"], max_length=77).input_ids)
    assert rel_err(z[1:], ref_text)[1] < 1e-2                       # row 1 = the prompt, row 0 = the negative prompt
    assert not torch.equal(z[0], z[1])


def test_custom_diffusion_attn_procs_and_textual_inversion():
    """sd.py:56-59: load_attn_procs replaces the cross-attention K/V projections (checked against the oracle with the same weights swapped
    in), load_textual_inversion appends a token embedding."""
    from customnerf_amd.sd import arch
    from customnerf_amd.sd import text_encoder as te
    from customnerf_amd.sd.unet import UNet
    cfg = arch.UNET_TINY
    sd = half_sd(arch.random_state_dict(arch.unet_params(cfg), seed=31))
    g = torch.Generator().manual_seed(9)
    procs, sd_cd = {}, dict(sd)
    for k in [k for k in sd if k.endswith("attn2.to_k.weight") or k.endswith("attn2.to_v.weight")]:
        w = (torch.randn(sd[k].shape, generator=g) / sd[k].shape[1] ** 0.5).half().float()
        which = "to_k" if k.endswith("to_k.weight") else "to_v"
        procs[k.replace(f"attn2.{which}.weight", f"attn2.processor.{which}_custom_diffusion.weight")] = w
        sd_cd[k] = w
    x = torch.randn(2, 4, 16, 16, generator=g).half().float()
    ctx = torch.randn(2, 77, cfg["cross_attention_dim"], generator=g).half().float()
    t = torch.tensor([300.0, 300.0])
    net = UNet(cfg, sd, "cuda")
    with torch.no_grad():
        base = net(to_nhwc8(x), t.cuda(), ctx.half().cuda()).clone()
        assert net.load_attn_procs(procs) == 16
        out = net(to_nhwc8(x), t.cuda(), ctx.half().cuda())
        ref = so.unet_forward(sd_cd, cfg, x, t, ctx)
    assert rel_err(out.permute(0, 3, 1, 2), ref)[1] < 1e-2
    assert rel_err(out, base)[1] > 1e-2                                     # the processors do change the prediction
    enc = te.CLIPTextEncoder(te.CLIP_TEXT_TINY, arch.random_state_dict(te.clip_text_params(te.CLIP_TEXT_TINY), 1), "cuda")
    tid = enc.add_token_embedding(torch.ones(te.CLIP_TEXT_TINY["width"]))
    assert tid == te.CLIP_TEXT_TINY["vocab_size"]
    ids = torch.zeros(1, 77, dtype=torch.long)
    ids[0, 3] = tid
    assert torch.isfinite(enc(ids.cuda())[0]).all()
