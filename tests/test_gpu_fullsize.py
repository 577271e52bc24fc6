"""GPU: size-independent properties at BASELINE.json's FULL sizes (cfg2: 16384 rays x 128 samples = 2,097,152 points, hash grid
L16 T2^19; cfg3: 512x512 VAE input, 64x64 latents) — where the CPU oracle would take minutes, the domain's own invariants are
checked instead: linearity of the gather in the table, adjointness of scatter and gather (<enc(T), G> = <T, scatter(G)>),
bit-determinism of the exact fixed-point scatter, sortedness / normalisation of the merged samples and composites, adjointness of
the implicit-GEMM convolution and its input-gradient loader at 512x512, linearity of the UNet-sized GEMMs."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N_RAYS, T = 16384, 128


def _ray_points(seed=0):
    """ray-structured points like the benchmark's full pass: 128 sorted samples along 16384 rays through the [-2,2]^3 box"""
    g = torch.Generator(device="cuda").manual_seed(seed)
    o = torch.randn(N_RAYS, 1, 3, device="cuda", generator=g) * 0.2 + torch.tensor([0.0, 0.0, 3.5], device="cuda")
    d = torch.nn.functional.normalize(torch.randn(N_RAYS, 1, 3, device="cuda", generator=g) * 0.25 + torch.tensor([0.0, 0.0, -1.0], device="cuda"), dim=-1)
    z = torch.sort(torch.rand(N_RAYS, T, 1, device="cuda", generator=g) * 4.0 + 1.5, dim=1).values
    return (o + d * z).clamp(-2, 2).reshape(-1, 3)


def _encoder():
    from customnerf_amd.gridencoder import GridEncoder
    return GridEncoder(num_levels=16, log2_hashmap_size=19, desired_resolution=2048, gridtype='hash').cuda()


def test_gather_linear_in_table_and_adjoint_to_scatter_full_size():
    enc = _encoder()
    x = _ray_points()
    g = torch.Generator(device="cuda").manual_seed(1)
    t1 = torch.rand(enc.embeddings.shape, device="cuda", generator=g) * 2 - 1
    t2 = torch.rand(enc.embeddings.shape, device="cuda", generator=g) * 2 - 1
    with torch.no_grad():
        enc.embeddings.copy_(t1)
        e1 = enc.encode(x, bound=2.0, half=False).clone()
        enc.embeddings.copy_(t2)
        e2 = enc.encode(x, bound=2.0, half=False).clone()
        enc.embeddings.copy_(t1 + 2.0 * t2)
        e12 = enc.encode(x, bound=2.0, half=False)
    assert e1.shape == (16, N_RAYS * T, 2)
    lin = float((e12 - (e1 + 2.0 * e2)).abs().max())
    assert lin < 2e-5, lin                                                  # 8-term fp32 interpolation sums: rounding only
    # adjointness through the binned (atomic-free) scatter at this size
    G = torch.randn(e1.shape, device="cuda", generator=g)
    with torch.no_grad():
        enc.embeddings.copy_(t1)
    out = enc.encode(x, bound=2.0, half=False)
    out.backward(G)
    lhs = float((out.detach().double() * G.double()).sum())
    rhs = float((t1.double() * enc.embeddings.grad.double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), abs(rhs), 1.0) + 1e-2, (lhs, rhs)


def test_fp16_scatter_is_bit_deterministic_and_adjoint_full_size():
    enc = _encoder()
    x = _ray_points(3)
    g = torch.Generator(device="cuda").manual_seed(2)
    with torch.no_grad():
        enc.embeddings.uniform_(-1, 1)
    G = (torch.randn(16, N_RAYS * T, 2, device="cuda", generator=g) * 0.05).half()
    grads = []
    for _ in range(2):
        enc.embeddings.grad = None
        out = enc.encode(x, bound=2.0, half=True)
        out.backward(G)
        grads.append(enc.embeddings.grad.clone())
    # exact 64-bit fixed-point sums, the split bins of the small dense levels included (fixed-point partial images, added exactly):
    # the whole table gradient is bit-reproducible, whatever order the records were written in
    assert torch.equal(grads[0], grads[1])
    # <enc, G> vs <T_half, scatter(G)>: the records are fp16-rounded w*g products (gridencoder.cu:328) -> 1e-3 relative
    lhs = float((out.detach().double() * G.double()).sum())
    rhs = float((enc.half_table().double() * grads[0].double()).sum())
    assert abs(lhs - rhs) <= 2e-3 * max(abs(lhs), abs(rhs)) + 1.0, (lhs, rhs)


def test_fp16_scatter_with_every_hashed_bin_split_is_deterministic_and_adjoint():
    """Four views' worth of samples in one backward pass (65536 rays x 128 samples = 8.4 M points, the multi-view edit step's size): the scatter's
    segment size is capped, EVERY hashed bin splits into two segments, and k_bin3_reduce_split's workgroups walk a split-bin list several times
    longer than its grid (round 6) — same invariants as at the benchmark size: bit-reproducible, adjoint to the gather"""
    enc = _encoder()
    x = torch.cat([_ray_points(10 + k) for k in range(4)], 0).contiguous()
    P = x.shape[0]
    g = torch.Generator(device="cuda").manual_seed(12)
    with torch.no_grad():
        enc.embeddings.uniform_(-1, 1)
    G = (torch.randn(16, P, 2, device="cuda", generator=g) * 0.05).half()
    grads = []
    for _ in range(2):
        enc.embeddings.grad = None
        out = enc.encode(x, bound=2.0, half=True)
        out.backward(G)
        grads.append(enc.embeddings.grad.clone())
    assert torch.equal(grads[0], grads[1])
    lhs = float((out.detach().double() * G.double()).sum())
    rhs = float((enc.half_table().double() * grads[0].double()).sum())
    assert abs(lhs - rhs) <= 2e-3 * max(abs(lhs), abs(rhs)) + 4.0, (lhs, rhs)
    del out, grads, G, x
    torch.cuda.empty_cache()                                           # (the scatter workspace of this size is ~14 GB: give it back)


def test_run_full_view_invariants():
    """one 128x128 view through run(): merged samples sorted per ray, weights in [0,1], sum <= 1, fg + bg weights partition by the
    soft mask, everything finite"""
    from customnerf_amd import scene as sc, tcnn
    from customnerf_amd.nerf.network_grid import NeRFNetwork
    from customnerf_amd.nerf.provider_utils import generate_rays
    tcnn.set_default_dtype(torch.float16)
    torch.manual_seed(0)
    opt = sc.make_opt(fp16=True)
    model = NeRFNetwork(opt).cuda().train()
    with torch.no_grad():
        model.pos_en.embeddings.uniform_(-0.3, 0.3)
    c2w = torch.from_numpy(sc.poses(8))[:1].cuda()
    o, d = generate_rays(c2w, *sc.intrinsics(128, 128), 128, 128, 1.0, 'nerfstudio')
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.float16):
        res = model.render(o.view(1, -1, 3), d.view(1, -1, 3), staged=False, perturb=True, force_all_rays=True, num_steps=64, upsample_steps=64)
    z = res['z_vals']
    assert z.shape == (N_RAYS, T) and bool((z[:, 1:] >= z[:, :-1]).all())
    for key in (None, 'fg', 'bg'):
        r = res if key is None else res[key]
        w = r['weights']
        assert torch.isfinite(w).all() and float(w.min()) >= 0.0 and float(w.max()) <= 1.0 + 1e-6
        assert float(r['weights_sum'].max()) <= 1.0 + 1e-4
        assert torch.isfinite(r['image']).all() and float(r['image'].min()) >= -1e-6 and float(r['image'].max()) <= 1.0 + 1e-4
    assert torch.isfinite(res['depth']).all() and float(res['depth'].min()) >= 0 and float(res['depth'].max()) <= 1.0 + 1e-4


def test_vae_sized_conv_is_adjoint_to_its_input_gradient():
    """<conv(x; W), g> = <x, dgrad(g; W)> at the VAE's largest shape (512x512x128, M = 262144 rows) and through a stride-2 downsample"""
    from customnerf_amd.sd import ops, pack
    g = torch.Generator().manual_seed(0)
    for (H, C, Co, stride, pad, out_hw) in ((512, 128, 128, 1, 1, None), (256, 128, 128, 2, 0, (128, 128))):
        x = torch.randn(1, H, H, C, generator=g).half().cuda()
        w = (torch.randn(Co, C, 3, 3, generator=g) / math.sqrt(9 * C)).half().float()
        y = ops.conv2d(x, pack.pack_conv(w).cuda(), None, 3, stride=stride, pad=pad, out_hw=out_hw)
        gy = torch.randn(y.shape, generator=g).half().cuda()
        dx = ops.conv2d(gy, pack.pack_conv_dgrad(w).cuda(), None, 3, stride=1, pad=3 - 1 - pad, tstride=stride, out_hw=(H, H))
        lhs = float((y.double() * gy.double()).sum())
        rhs = float((x.double() * dx.double()).sum())
        scale = float(y.double().norm() * gy.double().norm())
        assert abs(lhs - rhs) < 2e-3 * scale, (H, stride, lhs, rhs, scale)          # fp16 outputs on both sides


def test_unet_sized_gemm_linearity_and_split_k_consistency():
    """linear(a x1 + b x2) = a linear(x1) + b linear(x2) at the UNet's shapes (incl. the split-K 8x8 / 16x16 levels) within fp16 output
    rounding; the split-K and single-pass schedules of the same product agree"""
    from customnerf_amd.sd import ops
    g = torch.Generator().manual_seed(1)
    for (M, N, K) in ((8192, 320, 2880), (2048, 640, 5760), (512, 1280, 11520), (128, 1280, 23040), (8192, 2560, 320)):
        x1 = torch.randn(M, K, generator=g).half().cuda()
        x2 = torch.randn(M, K, generator=g).half().cuda()
        w = (torch.randn(N, K, generator=g) / math.sqrt(K)).half().cuda()
        y1, y2 = ops.linear(x1, w, out32=True), ops.linear(x2, w, out32=True)
        x12 = (x1.float() * 0.5 + x2.float() * 0.25).half()                       # exactly representable combination? not in general: compare to its own GEMM
        y12 = ops.linear(x12, w, out32=True)
        ref = (x12.float() @ w.float().t())
        assert float((y12 - ref).abs().max()) < 2e-3 * float(ref.abs().max()) + 1e-3
        lin = float((ops.linear((x1.float() * 2).half(), w, out32=True) - 2 * y1).abs().max())      # scaling by 2 is exact in fp16
        assert lin == 0.0, (M, N, K, lin)
        assert torch.isfinite(y2).all()
