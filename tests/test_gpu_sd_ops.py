"""GPU parity of the SDS primitives (include/customnerf_sd.h) against plain PyTorch float32 on the CPU.
diffusers' arithmetic is third-party and unpinned: these checks pin each primitive to the torch op the
public implementation calls (conv2d / linear / group_norm / layer_norm / softmax / gelu / interpolate)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def h(t):
    """round to half precision, keep float32 (what the kernels see)"""
    return t.half().float()


def close(a, b, rtol, atol, msg=""):
    np.testing.assert_allclose(a.detach().float().cpu().numpy(), b.detach().float().cpu().numpy(), rtol=rtol, atol=atol, err_msg=msg)


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("M,N,K", [(200, 320, 768), (154, 1280, 768), (8192, 320, 320), (128, 1280, 11520), (2, 1280, 320), (4096, 77, 40), (300, 8, 512)])
def test_linear(M, N, K):
    from customnerf_amd.sd import ops
    g = torch.Generator().manual_seed(M + N + K)
    x = h(torch.randn(M, K, generator=g))
    w = h(torch.randn(N, K, generator=g) / math.sqrt(K))
    b = torch.randn(N, generator=g)
    r = h(torch.randn(M, N, generator=g))
    y = ops.linear(x.half().cuda(), w.half().cuda(), bias=b.cuda(), residual=r.half().cuda())
    close(y, x @ w.t() + b + r, 2e-3, 4e-3)
    y2 = ops.linear(x.half().cuda(), w.half().cuda(), act=ops.ACT_SILU, alpha=0.5)
    close(y2, F.silu(0.5 * (x @ w.t())), 2e-3, 3e-3)
    y3 = ops.linear(x.half().cuda(), w.half().cuda(), act=ops.ACT_GELU, out32=True)
    assert y3.dtype == torch.float32
    close(y3, F.gelu(x @ w.t()), 1e-3, 1e-3)


CONVS = [  # B, Cin, H, W, Cout, k, stride, pad, ups
    (2, 320, 16, 16, 320, 3, 1, 1, 1),
    (2, 8, 16, 16, 320, 3, 1, 1, 1),          # conv_in (4 latent channels padded to 8)
    (1, 128, 40, 24, 128, 3, 1, 1, 1),
    (2, 640, 8, 8, 1280, 3, 1, 1, 1),         # split-K
    (2, 320, 16, 16, 320, 3, 2, 1, 1),        # UNet downsample
    (2, 1280, 4, 4, 1280, 3, 1, 1, 2),        # UNet upsample: nearest 2x fused into the conv
    (2, 320, 16, 16, 640, 1, 1, 0, 1),        # 1x1 shortcut
    (1, 64, 9, 7, 72, 3, 1, 1, 1),            # ragged everything
]


@pytest.mark.parametrize("B,Cin,H,W,Cout,k,stride,pad,ups", CONVS)
def test_conv2d_forward(B, Cin, H, W, Cout, k, stride, pad, ups):
    from customnerf_amd.sd import ops, pack
    g = torch.Generator().manual_seed(Cin + Cout + H)
    x = h(torch.randn(B, Cin, H, W, generator=g))
    w = h(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k))
    b = torch.randn(Cout, generator=g)
    temb = torch.randn(B, Cout, generator=g)
    xr = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups == 2 else x
    ref = F.conv2d(xr, w, b, stride=stride, padding=pad) + temb[:, :, None, None]
    res = h(torch.randn_like(ref))
    y = ops.conv2d(nhwc(x).half().cuda(), pack.pack_conv(w).cuda(), b.cuda(), k, stride=stride, pad=pad, ups=ups, bias_rows=temb.cuda(),
                   residual=nhwc(res).half().cuda())
    close(nchw(y), ref + res, 2e-3, 5e-3)


def test_conv2d_vae_downsample_and_input_gradients():
    """VAE downsample = F.pad(0,1,0,1) + conv(stride 2, pad 0); input-gradients of stride-1 and stride-2 convolutions through the
    transposed loader (tstride) against autograd."""
    from customnerf_amd.sd import ops, pack
    g = torch.Generator().manual_seed(3)
    B, C, H, W, Co = 1, 128, 24, 16, 136
    x = h(torch.randn(B, C, H, W, generator=g)).requires_grad_(True)
    w = h(torch.randn(Co, C, 3, 3, generator=g) / math.sqrt(9 * C))
    ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, None, stride=2)
    y = ops.conv2d(nhwc(x.detach()).half().cuda(), pack.pack_conv(w).cuda(), None, 3, stride=2, pad=0, out_hw=(H // 2, W // 2))
    close(nchw(y), ref, 2e-3, 4e-3)
    dy = h(torch.randn_like(ref))
    ref.backward(dy)
    dx = ops.conv2d(nhwc(dy).half().cuda(), pack.pack_conv_dgrad(w).cuda(), None, 3, stride=1, pad=2, tstride=2, out_hw=(H, W))
    close(nchw(dx), x.grad, 3e-3, 5e-3, "stride-2 dgrad")
    # stride 1, pad 1
    x2 = h(torch.randn(2, 64, 12, 20, generator=g)).requires_grad_(True)
    w2 = h(torch.randn(40, 64, 3, 3, generator=g) / 24.0)
    r2 = F.conv2d(x2, w2, None, padding=1)
    dy2 = h(torch.randn_like(r2))
    r2.backward(dy2)
    dx2 = ops.conv2d(nhwc(dy2).half().cuda(), pack.pack_conv_dgrad(w2).cuda(), None, 3, stride=1, pad=1)
    close(nchw(dx2), x2.grad, 3e-3, 5e-3, "stride-1 dgrad")
    # UNet-style stride 2 pad 1 dgrad and a 1x1
    x3 = h(torch.randn(1, 32, 10, 10, generator=g)).requires_grad_(True)
    w3 = h(torch.randn(48, 32, 3, 3, generator=g) / 17.0)
    r3 = F.conv2d(x3, w3, None, stride=2, padding=1)
    dy3 = h(torch.randn_like(r3))
    r3.backward(dy3)
    dx3 = ops.conv2d(nhwc(dy3).half().cuda(), pack.pack_conv_dgrad(w3).cuda(), None, 3, stride=1, pad=1, tstride=2, out_hw=(10, 10))
    close(nchw(dx3), x3.grad, 3e-3, 5e-3, "stride-2 pad-1 dgrad")


@pytest.mark.parametrize("B,HW,C,G,eps,silu", [(2, 256, 320, 32, 1e-5, True), (2, 64, 2560, 32, 1e-5, True), (1, 4096, 128, 32, 1e-6, True), (2, 100, 960, 32, 1e-6, False),
                                               (1, 777, 512, 32, 1e-6, False),
                                               # the UNet's shapes at 64 x 64 latents (and a ragged row count)
                                               (2, 4096, 320, 32, 1e-5, True), (2, 1024, 1920, 32, 1e-5, True), (1, 256, 2560, 32, 1e-5, True),
                                               (2, 4096, 960, 32, 1e-5, True), (1, 4096, 512, 32, 1e-6, False), (2, 4097, 640, 32, 1e-5, True)])
def test_groupnorm_forward_backward(B, HW, C, G, eps, silu):
    from customnerf_amd.sd import ops
    g = torch.Generator().manual_seed(C + HW)
    x = h(torch.randn(B, C, HW, generator=g) * 2 + 0.5).requires_grad_(True)
    gamma = (torch.rand(C, generator=g) + 0.5)
    beta = torch.randn(C, generator=g) * 0.3
    ref = F.group_norm(x, G, gamma, beta, eps)
    if silu:
        ref = F.silu(ref)
    xg = x.detach().permute(0, 2, 1).contiguous().half().cuda()
    y, sums = ops.groupnorm(xg, gamma.cuda(), beta.cuda(), G, eps, silu)
    close(y.permute(0, 2, 1), ref, 2e-3, 3e-3)
    dy = h(torch.randn_like(ref))
    ref.backward(dy)
    dyg = dy.permute(0, 2, 1).contiguous().half().cuda()
    dx = ops.groupnorm_backward(xg, dyg, gamma.cuda(), beta.cuda(), G, eps, silu, sums)
    close(dx.permute(0, 2, 1), x.grad, 5e-3, 4e-3)
    # statistics are exact integer (fixed-point) sums: the same bits on every run, whatever order the workgroups' atomics arrive in
    assert sums.dtype == torch.int64
    for _ in range(3):
        y2, sums2 = ops.groupnorm(xg, gamma.cuda(), beta.cuda(), G, eps, silu)
        assert torch.equal(sums2, sums) and torch.equal(y2, y)
        assert torch.equal(ops.groupnorm_backward(xg, dyg, gamma.cuda(), beta.cuda(), G, eps, silu, sums), dx)
    n = HW * (C // G)
    mean_ref = x.detach().half().float().view(B, G, -1).mean(-1)
    assert float((ops.gn_sums_to_float(sums)[..., 0].cpu() / n - mean_ref.double()).abs().max()) < 1e-4
    # (round 4) residual blocks: a second gradient arriving at x is added inside the apply kernel (one rounding instead of two), with the
    # scratch statistics handed over pre-zeroed — the same dx + r up to that rounding, and the same bits whether the scratch is passed or not
    r = h(torch.randn(B, HW, C, generator=g)).half().cuda()
    scratch = torch.zeros(B, G, 2, dtype=torch.int64, device="cuda")
    dxr = ops.groupnorm_backward(xg, dyg, gamma.cuda(), beta.cuda(), G, eps, silu, sums, residual=r, scratch=scratch)
    two_roundings = (dx.float() + r.float())
    assert float((dxr.float() - two_roundings).abs().max()) <= float(two_roundings.abs().max()) * 2 ** -10
    assert torch.equal(dxr, ops.groupnorm_backward(xg, dyg, gamma.cuda(), beta.cuda(), G, eps, silu, sums, residual=r))
    assert torch.equal(ops.groupnorm_backward(xg, dyg, gamma.cuda(), beta.cuda(), G, eps, silu, sums, scratch=torch.zeros_like(scratch)), dx)



def test_groupnorm_statistics_surface_overflow_and_non_finite_inputs():
    """integer sums cannot carry Inf / NaN and wrap silently — a non-finite element, a thread partial >= 2^30 or a total >= 2^40 must
    poison the (image, group) statistics (sd_gn_fix.h) so that the normalised group is NaN, as the float sums it replaced would have been; the
    other groups are untouched."""
    from customnerf_amd.sd import ops
    B, HW, C, G = 2, 4096, 128, 32
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, HW, C, generator=g).half().cuda()
    gamma, beta = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    y0, s0 = ops.groupnorm(x, gamma, beta, G, 1e-5, False)
    assert bool(torch.isfinite(y0).all()) and bool(torch.isfinite(ops.gn_sums_to_float(s0)).all())
    cg = C // G
    for bad, what in ((float("inf"), "inf element"), (float("nan"), "nan element")):
        xb = x.clone()
        xb[1, 77, 5 * cg + 1] = bad
        y, s = ops.groupnorm(xb, gamma, beta, G, 1e-5, False)
        f = ops.gn_sums_to_float(s)
        assert bool(torch.isnan(f[1, 5]).any()) and bool(torch.isfinite(f[0]).all()) and bool(torch.isfinite(f[1, :5]).all()) and bool(torch.isfinite(f[1, 6:]).all()), what
        yg = y.view(B, HW, G, cg)
        assert bool(torch.isnan(yg[1, :, 5]).all()) and bool(torch.isfinite(yg[0]).all()) and bool(torch.isfinite(yg[1, :, :5]).all()), what
    # a finite tensor whose sum of squares leaves the representable range (16384 elements of 60000^2 = 5.9e13 > 2^40): NaN, not a wrapped finite value
    xb = x.clone()
    xb[0, :, 2 * cg:3 * cg] = 60000.0
    y, s = ops.groupnorm(xb, gamma, beta, G, 1e-5, False)
    f = ops.gn_sums_to_float(s)
    assert bool(torch.isnan(f[0, 2, 1])) and bool(torch.isnan(y.view(B, HW, G, cg)[0, :, 2]).all()) and bool(torch.isfinite(y.view(B, HW, G, cg)[0, :, 3:]).all())
    # the backward statistics: a non-finite dy poisons the group's dx
    dy = torch.randn(B, HW, C, generator=g).half().cuda()
    dy[1, 9, 7 * cg] = float("inf")
    dx = ops.groupnorm_backward(x, dy, gamma, beta, G, 1e-5, False, s0)
    dxg = dx.view(B, HW, G, cg)
    assert bool(torch.isnan(dxg[1, :, 7]).all()) and bool(torch.isfinite(dxg[0]).all()) and bool(torch.isfinite(dxg[1, :, :7]).all())

@pytest.mark.parametrize("B,T,C,heads", [(2, 77, 768, 12), (1, 200, 80, 2), (1, 130, 160, 2), (2, 65, 320, 2), (1, 100, 96, 2)])
def test_attention_causal(B, T, C, heads):
    """causal = 1 (the CLIP text tower's mask): query i attends to keys <= i, in both kernel forms (LDS-DMA at head dims 64 / 40 / 80 / 160,
    register-staged at 48) and through the pre-transposed-V entry"""
    from customnerf_amd.sd import ops
    g = torch.Generator().manual_seed(T + C)
    q, k, v = (h(torch.randn(B, T, C, generator=g)) for _ in range(3))
    o = ops.attention(q.half().cuda(), k.half().cuda(), v.half().cuda(), heads, causal=True)
    qh, kh, vh = (t.view(B, T, heads, C // heads).transpose(1, 2) for t in (q, k, v))
    ref = F.scaled_dot_product_attention(qh, kh, vh, is_causal=True).transpose(1, 2).reshape(B, T, C)
    close(o, ref, 5e-3, 5e-3)
    o_vt = ops.attention_vt(q.half().cuda(), k.half().cuda(), ops.transpose_v(v.half().cuda()), heads, causal=True)
    assert torch.equal(o, o_vt)


@pytest.mark.parametrize("rows,C", [(100, 320), (77, 640), (513, 1280), (3, 768)])
def test_layernorm(rows, C):
    from customnerf_amd.sd import ops
    g = torch.Generator().manual_seed(C)
    x = h(torch.randn(rows, C, generator=g) * 3 + 1)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    y = ops.layernorm(x.half().cuda(), gamma.cuda(), beta.cuda(), 1e-5)
    close(y, F.layer_norm(x, (C,), gamma, beta, 1e-5), 2e-3, 4e-3)


def test_geglu_add_silu_concat_transpose():
    from customnerf_amd.sd import ops
    g = torch.Generator().manual_seed(0)
    x = h(torch.randn(37, 2 * 1280, generator=g) * 2)
    a, gate = x.chunk(2, dim=-1)
    close(ops.geglu(x.half().cuda()), a * F.gelu(gate), 2e-3, 2e-3)
    p, q = h(torch.randn(1001, generator=g)), h(torch.randn(1001, generator=g))
    close(ops.add(p.half().cuda(), q.half().cuda()), p + q, 1e-3, 1e-3)
    close(ops.silu(p.half().cuda()), F.silu(p), 1e-3, 1e-3)
    u, v = h(torch.randn(5, 7, 16, generator=g)), h(torch.randn(5, 7, 24, generator=g))
    assert torch.equal(ops.concat_channels(u.half().cuda(), v.half().cuda()).cpu().float(), torch.cat([u, v], -1))
    m = h(torch.randn(3, 77, 320, generator=g))
    dst = torch.full((3, 320, 80), 7.0, dtype=torch.float16, device="cuda")
    ops.transpose_batched(m.half().cuda(), 77, 320, 320, 80, 3, 77 * 320, 320 * 80, dst)
    assert torch.equal(dst[:, :, :77].cpu().float(), m.transpose(1, 2)) and torch.all(dst[:, :, 77:] == 0)


# the last two rows are the shapes the SD-1.5 UNet runs at 64x64 latents (nerf/sd.py:140): 4096-token self-attention (the most expensive
# instantiation, k_sd_attention<3,2>: head dim 40) and its cross-attention against the 77 text tokens
@pytest.mark.parametrize("B,Tq,Tk,C,heads", [(2, 256, 256, 320, 8), (2, 64, 77, 1280, 8), (1, 300, 300, 512, 1), (2, 1024, 77, 640, 8), (1, 100, 1100, 80, 2),
                                             (2, 4096, 4096, 320, 8), (2, 4096, 77, 320, 8), (3, 200, 333, 512, 8), (1, 130, 70, 96, 2),
                                             # edges of the LDS-DMA kernel's tiling (64- / 32-key iterations, ring of three buffers, partial query blocks)
                                             (1, 1, 1, 40, 1), (1, 33, 65, 80, 1), (1, 31, 129, 160, 1), (2, 257, 63, 128, 2), (1, 64, 192, 40, 1), (1, 5, 64, 64, 1)])
def test_attention(B, Tq, Tk, C, heads):
    from customnerf_amd.sd import ops
    g = torch.Generator().manual_seed(Tq + Tk)
    q, k, v = (h(torch.randn(B, T, C, generator=g)) for T in (Tq, Tk, Tk))
    if Tk >= 1024:
        q = h(q * 3.0)                                   # sharper rows: with thousands of unit-variance keys the softmax would average everything away
    o = ops.attention(q.half().cuda(), k.half().cuda(), v.half().cuda(), heads)
    d = C // heads
    qh, kh, vh = (t.view(B, -1, heads, d).transpose(1, 2) for t in (q, k, v))
    ref = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(B, Tq, C)
    close(o, ref, 5e-3, 5e-3)
    # (round 4) ops.attention hands V over as it is — the kernel transposes its key tiles in LDS; the transposed-operand entry (the
    # cross-attention path: V^T cached per prompt) runs the same MFMA sequence on the same values: identical bits.  Also as strided views of a
    # fused qkv projection, which is how the UNet calls it.
    if C // heads <= 160:
        o_vt = ops.attention_vt(q.half().cuda(), k.half().cuda(), ops.transpose_v(v.half().cuda()), heads)
        assert torch.equal(o, o_vt)
    if Tq == Tk:
        qkv = torch.cat([q, k, v], -1).half().cuda()
        o_view = ops.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads)
        assert torch.equal(o_view, o)


def test_softmax_backward():
    from customnerf_amd.sd import ops
    g = torch.Generator().manual_seed(5)
    s = h(torch.randn(50, 300, generator=g) * 2).requires_grad_(True)
    p = torch.softmax(s, -1)
    dp = h(torch.randn(50, 300, generator=g))
    p.backward(dp)
    P = torch.zeros(50, 304, dtype=torch.float16, device="cuda")
    P[:, :300] = p.detach().half().cuda()
    dP = torch.zeros(50, 304, dtype=torch.float16, device="cuda")
    dP[:, :300] = dp.half().cuda()
    ops.softmax_backward_(P, dP, 50, 300, 304)
    close(dP[:, :300], s.grad, 1e-2, 2e-4)
    assert torch.all(dP[:, 300:] == 0)


def test_image_front_end_and_sds_tail():
    from customnerf_amd.sd import ops
    g = torch.Generator().manual_seed(1)
    img = torch.rand(1, 3, 32, 32, generator=g).requires_grad_(True)
    ref = 2 * F.interpolate(img, (128, 128), mode="bilinear", align_corners=False) - 1
    out = ops.image_to_vae_input(img.detach().cuda(), 128, 128)
    close(out[..., :3].permute(0, 3, 1, 2), ref, 1e-3, 1e-3)
    assert torch.all(out[..., 3:] == 0)
    d_out = h(torch.randn(1, 128, 128, 8, generator=g))
    ref.backward(d_out[..., :3].permute(0, 3, 1, 2))
    d_img = ops.image_to_vae_input_backward(d_out.half().cuda(), 1, 32, 32)
    close(d_img, img.grad, 1e-4, 1e-4)
    # non-integer scale (utils_init_nerf.py:303 is called with whatever H, W the view has)
    img2 = torch.rand(2, 3, 24, 40, generator=g)
    close(ops.image_to_vae_input(img2.cuda(), 64, 64)[..., :3].permute(0, 3, 1, 2), 2 * F.interpolate(img2, (64, 64), mode="bilinear", align_corners=False) - 1, 1e-3, 1e-3)
    # ... and its input gradient (a gather over the bilinear footprint: fixed summation order, so repeatable bit for bit), up- and down-scaling
    for (Hi, Wi, Ho, Wo) in ((24, 40, 64, 64), (128, 128, 512, 512), (50, 30, 20, 45), (7, 5, 7, 5), (1, 3, 9, 2)):
        im = torch.rand(2, 3, Hi, Wi, generator=g).requires_grad_(True)
        rf = 2 * F.interpolate(im, (Ho, Wo), mode="bilinear", align_corners=False) - 1
        dy = h(torch.randn(2, Ho, Wo, 8, generator=g))
        rf.backward(dy[..., :3].permute(0, 3, 1, 2))
        got = ops.image_to_vae_input_backward(dy.half().cuda(), 2, Hi, Wi)
        close(got, im.grad, 1e-4, 2e-4)
        assert torch.equal(got, ops.image_to_vae_input_backward(dy.half().cuda(), 2, Hi, Wi))
    # timestep embedding (diffusers get_timestep_embedding, flip_sin_to_cos=True, freq shift 0)
    t = torch.tensor([981.0, 20.0])
    half = 160
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    arg = t[:, None] * freqs[None]
    close(ops.timestep_embedding(t.cuda(), 320), torch.cat([torch.cos(arg), torch.sin(arg)], -1), 0, 2e-3)
    # add_noise + CFG / SDS gradient (sd.py:133-148)
    lat, noise = torch.randn(1, 4, 8, 8, generator=g), torch.randn(1, 4, 8, 8, generator=g)
    ab = 0.37
    xin = ops.add_noise(lat.cuda(), noise.cuda(), ab)
    want = (math.sqrt(ab) * lat + math.sqrt(1 - ab) * noise).permute(0, 2, 3, 1)
    close(xin[0, ..., :4], want[0], 1e-3, 1e-3)
    assert torch.equal(xin[0], xin[1]) and torch.all(xin[..., 4:] == 0)
    eps = h(torch.randn(2, 8, 8, 8, generator=g))
    eps[1, 0, 0, 0] = float("nan")
    gr = ops.sds_grad(eps.half().cuda(), noise.cuda(), ab, 7.5, 0.01)
    eu, et = eps[0, ..., :4].permute(2, 0, 1)[None], eps[1, ..., :4].permute(2, 0, 1)[None]
    ref_g = torch.nan_to_num((1 - ab) * ((et + 7.5 * (et - eu)) - noise) * 0.01)
    close(gr, ref_g, 1e-5, 1e-6)


@pytest.mark.parametrize("B,C,H,Co", [(2, 320, 64, 320), (1, 128, 96, 128), (2, 64, 8, 128), (2, 320, 16, 640),
                                      (1, 128, 256, 128), (2, 256, 128, 256),                        # the VAE's large-M shapes
                                      (2, 640, 8, 1280), (2, 1280, 16, 1280), (2, 640, 32, 640), (8, 1280, 8, 1280), (2, 2560, 8, 1280)])   # split-K schedules (UNet 8^2 / 16^2 / 32^2)
def test_gemm_fused_groupnorm_statistics(B, C, H, Co):
    """the producing conv accumulates the next GroupNorm's statistics in its epilogue (gn=): same normalised output as the two-pass norm.
    Round 6: also when the library runs the problem split-K (the tail kernel accumulates them) — the request is always served."""
    from customnerf_amd.sd import ops, pack
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(B, H, H, C, generator=g).half().cuda()
    w = pack.pack_conv(torch.randn(Co, C, 3, 3, generator=g) / math.sqrt(9 * C)).cuda()
    b = torch.randn(Co, generator=g).cuda()
    gamma, beta = (torch.rand(Co, generator=g) + 0.5).cuda(), torch.randn(Co, generator=g).cuda()
    sums = torch.zeros(B, 32, 2, dtype=torch.int64, device="cuda")        # 64-bit fixed-point statistics (order-independent sums)
    y, ok = ops.conv2d(x, w, b, 3, gn=(sums, 32, H * H))
    y_ref = ops.conv2d(x, w, b, 3)
    assert torch.equal(y, y_ref)
    n_ref, s_ref = ops.groupnorm(y_ref, gamma, beta, 32, 1e-5, True)
    assert ok
    close(ops.gn_sums_to_float(sums).float(), ops.gn_sums_to_float(s_ref).float(), 2e-4, 1e-2)
    n_fused, _ = ops.groupnorm(y, gamma, beta, 32, 1e-5, True, sums=sums)
    close(n_fused, n_ref, 2e-3, 2e-3)
    # exact integer sums: a second run of either producer gives the same bits whatever the arrival order of its workgroups
    sums2 = torch.zeros_like(sums)
    ops.conv2d(x, w, b, 3, gn=(sums2, 32, H * H))
    _, s_ref2 = ops.groupnorm(y_ref, gamma, beta, 32, 1e-5, True)
    assert torch.equal(sums, sums2) and torch.equal(s_ref, s_ref2)


def test_large_m_gemms_match_the_reference_and_their_row_blocks():
    """The VAE's shapes (M = 64 K ... 262 K rows): the same problem cut into row blocks must give the same BITS (a tile's K loop does not depend on
    where the tile sits; both sizes run without split-K) — dense with bias / activation / residual; ragged M and N, a 3 x 3 convolution and the
    transposed-stride input gradient of the downsampling convolutions against the CPU reference."""
    from customnerf_amd.sd import ops, pack
    g = torch.Generator().manual_seed(77)
    M, N, K = 65536, 128, 1152
    x = torch.randn(M, K, generator=g).half().cuda()
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).half().cuda()
    b = torch.randn(N, generator=g).cuda()
    r = torch.randn(M, N, generator=g).half().cuda()
    for kw in (dict(bias=b, residual=r), dict(act=ops.ACT_SILU, alpha=0.5), dict(bias=b, act=ops.ACT_GELU)):
        big = ops.linear(x, w, **kw)
        parts = []
        for m0 in range(0, M, 32768):
            kws = dict(kw)
            if 'residual' in kws:
                kws['residual'] = r[m0:m0 + 32768]
            parts.append(ops.linear(x[m0:m0 + 32768], w, **kws))
        assert torch.equal(big, torch.cat(parts, 0)), kw.keys()
    Mr, Nr = 65536 + 77, 328
    xr = torch.randn(Mr, K, generator=g).half().cuda()
    wr = (torch.randn(Nr, K, generator=g) / math.sqrt(K)).half().cuda()
    br = torch.randn(Nr, generator=g).cuda()
    yr = ops.linear(xr, wr, bias=br)
    rows = torch.cat([torch.arange(0, 300), torch.arange(Mr - 300, Mr)])
    close(yr[rows.cuda()].float().cpu(), xr[rows.cuda()].float().cpu() @ wr.float().cpu().t() + br.cpu(), 2e-3, 4e-3)
    B, C, H, Co = 1, 128, 256, 128
    xc = h(torch.randn(B, C, H, H, generator=g))
    wc = h(torch.randn(Co, C, 3, 3, generator=g) / math.sqrt(9 * C))
    bc = torch.randn(Co, generator=g)
    yc = ops.conv2d(nhwc(xc).half().cuda(), pack.pack_conv(wc).cuda(), bc.cuda(), 3)
    close(nchw(yc), F.conv2d(xc, wc, bc, padding=1), 2e-3, 5e-3)
    x2 = h(torch.randn(1, 128, 256, 256, generator=g)).requires_grad_(True)
    w2 = h(torch.randn(128, 128, 3, 3, generator=g) / math.sqrt(9 * 128))
    ref = F.conv2d(F.pad(x2, (0, 1, 0, 1)), w2, None, stride=2)
    dy = h(torch.randn_like(ref))
    ref.backward(dy)
    dx = ops.conv2d(nhwc(dy).half().cuda(), pack.pack_conv_dgrad(w2).cuda(), None, 3, stride=1, pad=2, tstride=2, out_hw=(256, 256))
    close(nchw(dx), x2.grad, 3e-3, 5e-3, "stride-2 dgrad at 256 x 256")


@pytest.mark.parametrize("M,C", [(8192, 320), (128, 1280), (77, 64)])
def test_linear_fused_geglu(M, C):
    """GEGLU folded into the projection's epilogue (interleaved value/gate rows) == proj followed by the GEGLU kernel"""
    from customnerf_amd.sd import ops, pack
    g = torch.Generator().manual_seed(M)
    x = h(torch.randn(M, C, generator=g))
    w = h(torch.randn(8 * C, C, generator=g) / math.sqrt(C))
    b = torch.randn(8 * C, generator=g)
    a, gate = (x @ w.t() + b).chunk(2, dim=-1)
    wi, bi = pack.interleave_geglu(w, b)
    y = ops.linear(x.half().cuda(), wi.half().cuda(), bias=bi.cuda(), act=ops.ACT_GEGLU)
    assert y.shape == (M, 4 * C)
    close(y, a * F.gelu(gate), 3e-3, 3e-3)


def test_splitk_workspace_outgrown_after_graph_capture():
    """A captured graph keeps the split-K workspace pointer it was captured with (UNet.graphed).  When a later, larger GEMM outgrows
    the shared workspace, the old buffer must stay allocated: otherwise the caching allocator hands its block to another tensor and
    the replay's split-K partial sums land in that tensor (and eps itself is summed from whatever the other owner wrote)."""
    from customnerf_amd.sd import ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(11)
    M, N, K = 128, 1280, 11520                              # a split-K shape (test_linear covers its numerics)
    x = h(torch.randn(M, K, generator=g)).half().cuda()
    w = h(torch.randn(N, K, generator=g) / math.sqrt(K)).half().cuda()
    saved = ops._WS.pop(dev, None)
    try:
        ops._WS[dev] = torch.empty(32 << 20, dtype=torch.uint8, device=dev)       # small enough to be outgrown below
        out = torch.empty(M, N, dtype=torch.float16, device=dev)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            ops.linear(x, w, out=out)
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            ops.linear(x, w, out=out)
        graph.replay()
        torch.cuda.synchronize()
        first = out.clone()
        old = ops._WS[dev]
        old_ptr, old_bytes = old.data_ptr(), old.numel()
        del old
        grown = ops._workspace(old_bytes + (16 << 20), dev)                       # what a larger post-capture GEMM does
        assert grown.data_ptr() != old_ptr
        assert any(b.data_ptr() == old_ptr for b in ops._WS_RETIRED), "the outgrown workspace must be kept alive"
        # anything allocated now must not alias the retired block; poison fresh allocations of the same size class and replay
        poison = [torch.full((old_bytes,), 0x7f, dtype=torch.uint8, device=dev) for _ in range(2)]
        assert all(p.data_ptr() != old_ptr for p in poison)
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, first)
        assert all(bool((p == 0x7f).all()) for p in poison), "a graph replay wrote into memory it no longer owns"
    finally:
        if saved is not None:
            ops._WS[dev] = saved


def test_edit_step_glue_functions_match_the_torch_expressions():
    """csrc/edit_ops.hip through sd/edit_fn.py, each against the chain of torch ops it replaced in the editing step (round 5): the ray buffer ->
    NCHW images and back (exact: pure data movement), keep_bg * l1_loss and its gradient, the VAE posterior sample (clamp / exp / sample / scale)
    and its gradient into the half moments, the SDS loss (value and gradient), and the host-float writer."""
    import torch.nn.functional as F
    from customnerf_amd.sd import ops
    from customnerf_amd.sd.edit_fn import RayImages, ScaledL1, SDSLoss, SampleLatents
    g = torch.Generator(device='cuda').manual_seed(3)
    B, H, W = 2, 12, 20
    # ---- ray buffer <-> images
    out_ray = torch.randn(3, B * H * W, 6, device='cuda', generator=g).requires_grad_(True)
    ref_ray = out_ray.detach().clone().requires_grad_(True)
    imgs = RayImages.apply(out_ray, B, H, W)
    refs = [ref_ray[v][:, 0:3].reshape(B, H, W, 3).permute(0, 3, 1, 2).contiguous() for v in range(3)]
    for a, b in zip(imgs, refs):
        assert torch.equal(a, b)
    d_all, d_bg = torch.randn(B, 3, H, W, device='cuda', generator=g), torch.randn(B, 3, H, W, device='cuda', generator=g)
    torch.autograd.backward([imgs[0], imgs[2]], [d_all, d_bg])                       # the fg image is unused: its gradient arrives as None
    torch.autograd.backward([refs[0], refs[2]], [d_all, d_bg])
    assert torch.equal(out_ray.grad, ref_ray.grad)
    # ---- scaled L1 (an exact tie a == b has gradient 0, like torch's sgn)
    a = torch.rand(B, 3, H, W, device='cuda', generator=g)
    b = torch.rand(B, 3, H, W, device='cuda', generator=g)
    b[0, 0, 0, :5] = a[0, 0, 0, :5]
    b1, b2 = b.clone().requires_grad_(True), b.clone().requires_grad_(True)
    l1 = ScaledL1.apply(a, b1, 1000.0)
    l2 = 1000.0 * F.l1_loss(a, b2)
    up = torch.tensor(32768.0, device='cuda')
    l1.backward(up); l2.backward(up)
    assert l1.shape == () and abs(float(l1) - float(l2)) < 2e-6 * abs(float(l2))
    assert torch.allclose(b1.grad, b2.grad, rtol=1e-6, atol=0) and float(b1.grad[0, 0, 0, 0]) == 0.0
    # ---- posterior sample
    h = w = 16
    mom = (torch.randn(B, h, w, 8, device='cuda', generator=g) * 3).half()
    mom[0, 0, 0, 4] = 25.0; mom[0, 0, 1, 5] = -31.0                                  # outside the clamp: no gradient into logvar there
    noise = torch.randn(B, 4, h, w, device='cuda', generator=g)
    m1, m2 = mom.clone().requires_grad_(True), mom.clone().requires_grad_(True)
    lat = SampleLatents.apply(m1, noise, 0.18215)
    mm = m2.float().permute(0, 3, 1, 2)
    mean, logvar = mm.chunk(2, dim=1)
    lat_ref = (mean + torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0)) * noise) * 0.18215
    assert lat.shape == (B, 4, h, w) and lat.is_contiguous()
    assert torch.allclose(lat, lat_ref, rtol=2e-6, atol=1e-7)
    d_lat = torch.randn(B, 4, h, w, device='cuda', generator=g)
    lat.backward(d_lat); lat_ref.backward(d_lat)
    assert m1.grad.dtype == torch.float16 and float(m1.grad[0, 0, 0, 4]) == 0.0 and float(m1.grad[0, 0, 1, 5]) == 0.0
    fin = torch.isfinite(m2.grad.float())
    assert torch.allclose(m1.grad.float()[fin], m2.grad.float()[fin], rtol=2e-3, atol=1e-6)       # one half rounding each, of float32 values 1 ulp apart
    # ---- SDS loss
    x = torch.randn(1, 4, 64, 64, device='cuda', generator=g)
    gr = torch.randn(1, 4, 64, 64, device='cuda', generator=g) * 30
    x1, x2 = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    s1 = SDSLoss.apply(x1, gr)
    s2 = 0.5 * F.mse_loss(x2, (x2 - gr).detach(), reduction='sum')
    s1.backward(up); s2.backward(up)
    assert abs(float(s1) - float(s2)) < 1e-5 * float(s2)
    assert torch.equal(x1.grad, x2.grad)
    # ---- host floats
    dst = torch.zeros(2, device='cuda')
    ops.set_floats(dst, (437.0, 437.0))
    assert dst.tolist() == [437.0, 437.0]


def test_split_k_gemms_are_exact_under_repetition():
    """Shapes the cost model splits along K (small M, long K: the 8 x 8 / 16 x 16 UNet levels; fp32 partial tiles in the shared workspace, summed in
    split order by the tail pass) — dense with bias + residual, GEGLU pairs, fp32 output, a 3 x 3 convolution, ragged M / N — each against the CPU
    reference, and 60 back-to-back launches on rotating inputs bit-equal to a second pass over the same inputs (the workspace is reused by every
    launch: a tail that read a partial tile of the previous launch shows up as a mismatch)."""
    from customnerf_amd.sd import ops, pack
    g = torch.Generator().manual_seed(91)
    M, N, K = 128, 1280, 11520
    w = h(torch.randn(N, K, generator=g) / math.sqrt(K))
    b = torch.randn(N, generator=g)
    wc, bc = w.half().cuda(), b.cuda()
    xs = [h(torch.randn(M, K, generator=g)) for _ in range(6)]
    r = h(torch.randn(M, N, generator=g))
    y = ops.linear(xs[0].half().cuda(), wc, bias=bc, residual=r.half().cuda())
    close(y, xs[0] @ w.t() + b + r, 2e-3, 4e-3)
    y32 = ops.linear(xs[1].half().cuda(), wc, act=ops.ACT_GELU, out32=True)
    close(y32, F.gelu(xs[1] @ w.t()), 1e-3, 1e-3)
    yg = ops.linear(xs[2].half().cuda(), wc, bias=bc, act=ops.ACT_GEGLU)
    full = xs[2] @ w.t() + b
    close(yg, full[:, 0::2] * F.gelu(full[:, 1::2]), 2e-3, 4e-3)
    Mr, Nr = 100, 1284                                                             # ragged: the last tiles lie partly outside
    wr = h(torch.randn(Nr, K, generator=g) / math.sqrt(K))
    xr = h(torch.randn(Mr, K, generator=g))
    close(ops.linear(xr.half().cuda(), wr.half().cuda()), xr @ wr.t(), 2e-3, 4e-3)
    x4 = torch.randn(2, 8, 8, 640, generator=g).half()
    w4 = torch.randn(1280, 640, 3, 3, generator=g) / math.sqrt(9 * 640)
    y4 = ops.conv2d(x4.cuda(), pack.pack_conv(w4).cuda(), None, 3)
    close(y4.permute(0, 3, 1, 2), F.conv2d(x4.float().permute(0, 3, 1, 2), w4.half().float(), padding=1), 2e-3, 4e-3)
    xg = [x.half().cuda() for x in xs]
    first = [ops.linear(xg[i % 6], wc, bias=bc).clone() for i in range(60)]
    torch.cuda.synchronize()
    again = [ops.linear(xg[i % 6], wc, bias=bc).clone() for i in range(60)]
    for i in range(60):
        assert torch.equal(first[i], again[i]) and torch.equal(first[i], first[i % 6]), i


@pytest.mark.parametrize("B,H,C1,C2", [(2, 8, 1280, 1280), (2, 16, 1280, 640), (2, 32, 640, 320), (2, 64, 320, 320), (8, 8, 1280, 1280), (1, 8, 64, 64)])
def test_concat_accumulates_groupnorm_statistics(B, H, C1, C2):
    """round 6: the UNet's up blocks normalise a channel concat first thing — cnerf_sd_concat_gn accumulates that norm's statistics while it copies
    (a k_gn_stats launch per up block before): same copy, same fixed-point statistics as the stand-alone pass up to the rounding of the partial sums,
    bit-identical from run to run."""
    from customnerf_amd.sd import ops
    g = torch.Generator().manual_seed(C1 + H)
    a = (torch.randn(B, H, H, C1, generator=g) * 1.5 + 0.3).half().cuda()
    b = (torch.randn(B, H, H, C2, generator=g) * 0.7 - 0.2).half().cuda()
    y_ref = ops.concat_channels(a, b)
    assert torch.equal(y_ref, torch.cat([a, b], -1))
    sums = torch.zeros(B, 32, 2, dtype=torch.int64, device="cuda")
    y, ok = ops.concat_channels(a, b, gn=(sums, 32, H * H))
    assert torch.equal(y, y_ref)
    if (C1 + C2) // 32 < 8:                               # fewer than 8 channels per group: not admissible, nothing accumulated
        assert not ok and torch.all(sums == 0)
        return
    assert ok
    gamma, beta = (torch.rand(C1 + C2, generator=g) + 0.5).cuda(), torch.randn(C1 + C2, generator=g).cuda()
    n_ref, s_ref = ops.groupnorm(y_ref, gamma, beta, 32, 1e-5, True)
    close(ops.gn_sums_to_float(sums).float(), ops.gn_sums_to_float(s_ref).float(), 2e-4, 1e-2)
    n_fused, _ = ops.groupnorm(y, gamma, beta, 32, 1e-5, True, sums=sums)
    close(n_fused, n_ref, 2e-3, 2e-3)
    sums2 = torch.zeros_like(sums)
    ops.concat_channels(a, b, gn=(sums2, 32, H * H))
    assert torch.equal(sums, sums2)


@pytest.mark.parametrize("M,N,K,res", [(128, 1280, 1280, True), (512, 1280, 1280, True), (2048, 640, 640, True), (2048, 640, 640, False), (512, 1280, 5120, True),
                                       (8192, 320, 320, True), (77, 768, 768, False), (130, 320, 1280, True)])
def test_linear_fused_layernorm(M, N, K, res):
    """round 6: LayerNorm of a projection's output rows written by the split-K tail kernel (ops.linear(ln=)) — y and LayerNorm(y) must be
    bit-identical to the projection followed by the stand-alone layernorm launch, whichever schedule the library picks (the one-launch GEMM
    falls back to that launch inside ops.linear)."""
    from customnerf_amd.sd import ops, pack
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(1, M, K, generator=g).half().cuda()
    w = pack.pack_linear(torch.randn(N, K, generator=g) / math.sqrt(K)).cuda()
    b = torch.randn(N, generator=g).cuda()
    r = torch.randn(1, M, N, generator=g).half().cuda() if res else None
    gamma, beta = (torch.rand(N, generator=g) + 0.5).cuda(), torch.randn(N, generator=g).cuda()
    y_ref = ops.linear(x, w, bias=b, residual=r)
    n_ref = ops.layernorm(y_ref, gamma, beta)
    y, n = ops.linear(x, w, bias=b, residual=r, ln=(gamma, beta))
    assert torch.equal(y, y_ref) and torch.equal(n, n_ref)
    want = torch.nn.functional.layer_norm(y_ref.float().cpu(), (N,), gamma.cpu(), beta.cpu(), 1e-5)
    close(n, want, 2e-3, 2e-3)
