"""GPU parity of the fused MFMA field kernels against the oracle field (oracle/torch_oracle.py FieldRef / mlp_forward)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import torch_oracle as to      # noqa: E402


def cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def make_case(L, n_geo, P, seed=0, scale=0.5):
    from customnerf_amd.gridencoder import GridEncoder
    ref = to.FieldRef(bound=2.0, num_levels=L, n_hidden_geo=n_geo, seed=seed)
    g = torch.Generator().manual_seed(seed + 7)
    with torch.no_grad():
        ref.pos_en.embeddings.copy_((torch.rand(ref.pos_en.embeddings.shape, generator=g) * 2 - 1) * scale)
    enc = GridEncoder(num_levels=L, log2_hashmap_size=19, desired_resolution=2048, gridtype='hash').cuda()
    with torch.no_grad():
        enc.embeddings.copy_(ref.pos_en.embeddings.cuda())
    rng = np.random.default_rng(seed)
    x = ((rng.random((P, 3)) * 2 - 1) * 1.9).astype(np.float32)
    x[:5] *= 0.05                                     # inside the gaussian density blob
    d = rng.standard_normal((P, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    return ref, enc, x, d


@pytest.mark.parametrize("L,n_geo", [(16, 2), (4, 1), (16, 1), (8, 2)])
@pytest.mark.parametrize("half", [False, True], ids=["f32", "f16"])
def test_field_forward(L, n_geo, half):
    from customnerf_amd.field import field_forward_raw
    P = 4133                                         # ragged vs the 32-sample tile and the persistent grid
    ref, enc, x, d = make_case(L, n_geo, P)
    ref.half = half
    ref.pos_en.half = half
    with torch.no_grad():
        s_ref, c_ref, _ = ref(torch.from_numpy(x), torch.from_numpy(d))
        e = enc.encode(cuda(x), bound=2.0, half=half)                 # [L,P,2] kernel layout
        s, c = field_forward_raw(e, cuda(x), cuda(d), 1, 2 * L, n_geo, 4, ref.network.cuda(), ref.density_network.cuda(), ref.rgb_network.cuda())
        s2, c2 = field_forward_raw(e, cuda(x), None, 1, 2 * L, n_geo, 4, ref.network.cuda(), ref.density_network.cuda(), None, with_rgb=False)
    assert c2 is None and torch.equal(s2, s)
    if half:
        # fp16 activations: the summation order differs (MFMA vs CPU matmul) so a hidden unit can round to the neighbouring
        # half; the log-density is compared at 2e-2 (sigma = exp(raw + blob), raw has ~1e-2 fp16 resolution near |raw|~8)
        np.testing.assert_allclose(np.log(s.cpu().numpy()), np.log(s_ref.numpy()), rtol=0, atol=3e-2)
        np.testing.assert_allclose(c.cpu().numpy(), c_ref.numpy(), rtol=0, atol=4e-3)
    else:
        np.testing.assert_allclose(s.cpu().numpy(), s_ref.numpy(), rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(c.cpu().numpy(), c_ref.numpy(), rtol=0, atol=2e-6)
    assert c.shape == (P, 4) and float(c.min()) > 0 and float(c.max()) < 1


def test_field_forward_dir_group_and_rgb3():
    """one direction per group of samples (samples of a ray share rays_d) and the 3-channel colour head"""
    from customnerf_amd.field import field_forward_raw
    L, n_geo, rays, spr = 16, 2, 37, 24
    P = rays * spr
    ref, enc, x, d = make_case(L, n_geo, P, seed=3)
    d_ray = d[:rays]
    with torch.no_grad():
        s_ref, c_ref, _ = ref(torch.from_numpy(x), torch.from_numpy(np.repeat(d_ray, spr, axis=0)))
        e = enc.encode(cuda(x), bound=2.0, half=False)
        s, c = field_forward_raw(e, cuda(x), cuda(d_ray), spr, 2 * L, n_geo, 3, ref.network.cuda(), ref.density_network.cuda(), ref.rgb_network.cuda())
    np.testing.assert_allclose(c.cpu().numpy()[:, :3], c_ref.numpy()[:, :3], rtol=0, atol=2e-6)
    assert torch.all(c[:, 3] == 0)
    np.testing.assert_allclose(s.cpu().numpy(), s_ref.numpy(), rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("L,n_geo", [(16, 2), (4, 1)])
@pytest.mark.parametrize("half", [False, True], ids=["f32", "f16"])
def test_field_backward(L, n_geo, half):
    """d(sigma, rgbc)/d(grid table, MLP weights) through the fused backward + the grid scatter, against autograd on the oracle."""
    from customnerf_amd.field import field
    P = 2077
    ref, enc, x, d = make_case(L, n_geo, P, seed=5)
    ref.half = half
    ref.pos_en.half = half
    rng = np.random.default_rng(9)
    gs = (rng.standard_normal(P) * 0.05).astype(np.float32)
    gc = rng.standard_normal((P, 4)).astype(np.float32)
    s_ref, c_ref, _ = ref(torch.from_numpy(x), torch.from_numpy(d))
    torch.autograd.backward([s_ref, c_ref], [torch.from_numpy(gs), torch.from_numpy(gc)])
    pn, pd, pr = (t.detach().clone().cuda().requires_grad_(True) for t in (ref.network, ref.density_network, ref.rgb_network))
    e = enc.encode(cuda(x), bound=2.0, half=half)
    s, c = field(e, cuda(x), cuda(d), 1, 2 * L, n_geo, 4, pn, pd, pr)
    torch.autograd.backward([s, c], [cuda(gs), cuda(gc)])
    # fp16: activations, dz and the weight-gradient GEMM operands are halves (tcnn numerics); the oracle keeps dz in fp32
    rt, at = (3e-2, 3e-3) if half else (1e-3, 1e-5)
    for name, a, b in (("net", pn.grad, ref.network.grad), ("den", pd.grad, ref.density_network.grad), ("rgb", pr.grad, ref.rgb_network.grad),
                       ("grid", enc.embeddings.grad, ref.pos_en.embeddings.grad)):
        a, b = a.cpu().numpy(), b.numpy()
        scale = float(np.abs(b).max())
        assert scale > 0, name
        err = np.abs(a - b).max() / scale
        assert err < (rt if name != "grid" else rt * 2), f"{name}: max|diff|/max|ref| = {err:.3e}"
        np.testing.assert_allclose(a, b, rtol=rt * 10, atol=max(at, rt * scale), err_msg=name)
    # padded parameter rows/columns (tcnn pads 1 -> 16 outputs, 91 -> 96 inputs) must get exactly zero gradient
    assert torch.all(pd.grad[4096 + 64:] == 0)
    assert torch.all(pr.grad[:64 * 96].view(64, 96)[:, 91:] == 0)


@pytest.mark.parametrize("n_geo,P,dir_group,L", [(2, 70003, 1, 16), (1, 4099, 1, 16), (2, 64 * 700, 64, 16), (1, 32 * 3 + 5, 32, 16), (2, 17, 1, 16),
                                                 (2, 5000, 1, 12), (1, 2500, 128, 9)])
def test_field_backward_pipeline_shapes(n_geo, P, dir_group, L):
    """k_field_bwd_x2 (fp16, 9..16 levels = feature width padded to 32): several tiles per wave pair (the software pipeline really cycles), a ragged last tile, tile
    counts below the pipeline depth, one / two hidden layers, and one direction per group of samples (the per-tile direction path when the
    group is a multiple of the 32-sample tile) — against autograd on the oracle field."""
    from customnerf_amd.field import field
    ref, enc, x, d = make_case(L, n_geo, P, seed=11)
    ref.half = True
    ref.pos_en.half = True
    n_dir = (P + dir_group - 1) // dir_group
    d_grp = d[:n_dir]
    d_full = np.repeat(d_grp, dir_group, axis=0)[:P]
    rng = np.random.default_rng(13)
    gs = (rng.standard_normal(P) * 0.05).astype(np.float32)
    gc = rng.standard_normal((P, 4)).astype(np.float32)
    s_ref, c_ref, _ = ref(torch.from_numpy(x), torch.from_numpy(d_full))
    torch.autograd.backward([s_ref, c_ref], [torch.from_numpy(gs), torch.from_numpy(gc)])
    pn, pd, pr = (t.detach().clone().cuda().requires_grad_(True) for t in (ref.network, ref.density_network, ref.rgb_network))
    e = enc.encode(cuda(x), bound=2.0, half=True)
    s, c = field(e, cuda(x), cuda(d_grp), dir_group, 2 * L, n_geo, 4, pn, pd, pr)
    torch.autograd.backward([s, c], [cuda(gs), cuda(gc)])
    np.testing.assert_allclose(c.detach().cpu().numpy(), c_ref.detach().numpy(), rtol=0, atol=4e-3)
    for name, a, b in (("net", pn.grad, ref.network.grad), ("den", pd.grad, ref.density_network.grad), ("rgb", pr.grad, ref.rgb_network.grad),
                       ("grid", enc.embeddings.grad, ref.pos_en.embeddings.grad)):
        a, b = a.cpu().numpy(), b.numpy()
        scale = float(np.abs(b).max())
        assert scale > 0, name
        # the sums run over up to 70 k samples of half-precision products: the budget grows with sqrt(P) from the 3e-2 of the 2 k-sample case
        assert np.abs(a - b).max() / scale < 3e-2 * (2 if name == "grid" else 1), f"{name}: max|diff|/max|ref| = {np.abs(a - b).max() / scale:.3e}"
    assert torch.all(pd.grad[4096 + 64:] == 0)
    assert torch.all(pr.grad[:64 * 96].view(64, 96)[:, 91:] == 0)


def test_field_backward_is_reproducible():
    """the per-pair partial sums and their reduction have a fixed order: two runs give the same bits"""
    from customnerf_amd.field import field
    L, n_geo, P = 16, 2, 40000
    ref, enc, x, d = make_case(L, n_geo, P, seed=2)
    pn0, pd0, pr0 = ref.network, ref.density_network, ref.rgb_network
    g = torch.Generator().manual_seed(0)
    gs, gc = torch.randn(P, generator=g).cuda() * 0.05, torch.randn(P, 4, generator=g).cuda()
    outs = []
    for _ in range(2):
        pn, pd, pr = (t.detach().clone().cuda().requires_grad_(True) for t in (pn0, pd0, pr0))
        with torch.no_grad():
            e = enc.encode(cuda(x), bound=2.0, half=True)
        e.requires_grad_(True)
        s, c = field(e, cuda(x), cuda(d), 1, 2 * L, n_geo, 4, pn, pd, pr)
        torch.autograd.backward([s, c], [gs, gc])
        outs.append((pn.grad.clone(), pd.grad.clone(), pr.grad.clone(), e.grad.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


# ---- packed weights (ABI 5: cnerf_field_pack_weights + the _img entry points) ----
def _field_run(enc_t, x, d, dir_group, L, n_geo, params, gs, gc, wimg):
    from customnerf_amd.field import field
    pn, pd, pr = (t.detach().clone().requires_grad_(True) for t in params)
    if wimg == "pack":
        from customnerf_amd.field import packed_weights
        wimg = packed_weights({}, 2 * L, n_geo, 4, pn, pd, pr)
    e = enc_t.detach().clone().requires_grad_(True)
    s, c = field(e, x, d, dir_group, 2 * L, n_geo, 4, pn, pd, pr, wimg=wimg)
    torch.autograd.backward([s, c], [gs, gc])
    return s.detach(), c.detach(), e.grad, pn.grad, pd.grad, pr.grad


@pytest.mark.parametrize("L,n_geo,P,dir_group", [(16, 2, 40000, 1), (16, 1, 64 * 300, 64), (12, 2, 5003, 1), (4, 1, 3000, 1)])
def test_packed_weights_are_bit_identical(L, n_geo, P, dir_group):
    """forward and backward reading the packed fp16 image (k_field_pack) against the same launches staging from the float32 parameters:
    the image holds the very halves the kernels stage, so every output is the same bits ((4, 1): the narrow-encoding backward accepts an
    image and ignores it)"""
    ref, enc, x, d = make_case(L, n_geo, P, seed=5)
    params = [t.detach().clone().cuda() for t in (ref.network, ref.density_network, ref.rgb_network)]
    n_dir = (P + dir_group - 1) // dir_group
    g = torch.Generator().manual_seed(1)
    gs, gc = torch.randn(P, generator=g).cuda() * 0.05, torch.randn(P, 4, generator=g).cuda()
    with torch.no_grad():
        e = enc.encode(cuda(x), bound=2.0, half=True)
    a = _field_run(e, cuda(x), cuda(d[:n_dir]), dir_group, L, n_geo, params, gs, gc, None)
    b = _field_run(e, cuda(x), cuda(d[:n_dir]), dir_group, L, n_geo, params, gs, gc, "pack")
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    # density-only launch (the image's rgb layers are not copied)
    from customnerf_amd.field import field_forward_raw, packed_weights
    s0, _ = field_forward_raw(e, cuda(x), None, 1, 2 * L, n_geo, 4, params[0], params[1], None, with_rgb=False)
    s1, _ = field_forward_raw(e, cuda(x), None, 1, 2 * L, n_geo, 4, params[0], params[1], None, with_rgb=False,
                              wimg=packed_weights({}, 2 * L, n_geo, 4, *params))
    assert torch.equal(s0, s1) and torch.equal(s0, a[0])


def test_packed_weights_follow_the_parameters():
    """the cache repacks when a parameter's version counter or fused-optimiser epoch moved, records a pack inside a stream capture (a replay
    after a parameter update then evaluates the NEW parameters), and drops its eager freshness after a capture"""
    from customnerf_amd.field import field_forward_raw, packed_weights
    L, n_geo, P = 16, 2, 4096
    ref, enc, x, d = make_case(L, n_geo, P, seed=9)
    pn, pd, pr = (t.detach().clone().cuda() for t in (ref.network, ref.density_network, ref.rgb_network))
    with torch.no_grad():
        e = enc.encode(cuda(x), bound=2.0, half=True)
    xs, ds = cuda(x), cuda(d)
    run = lambda w: field_forward_raw(e, xs, ds, 1, 2 * L, n_geo, 4, pn, pd, pr, wimg=w)
    cache = {}
    w = packed_weights(cache, 2 * L, n_geo, 4, pn, pd, pr)
    assert packed_weights(cache, 2 * L, n_geo, 4, pn, pd, pr).data_ptr() == w.data_ptr()
    s0, c0 = run(w)
    pn.mul_(1.25)                                                         # in-place op: the version counter moves
    s1, c1 = run(packed_weights(cache, 2 * L, n_geo, 4, pn, pd, pr))
    sr, cr = run(None)
    assert torch.equal(s1, sr) and torch.equal(c1, cr) and not torch.equal(c1, c0)
    pr.data.mul_(0.5)                                                     # a raw write + the fused optimiser's epoch (optim.FusedAdam.step)
    pr._cnerf_epoch = getattr(pr, '_cnerf_epoch', 0) + 1
    s2, c2 = run(packed_weights(cache, 2 * L, n_geo, 4, pn, pd, pr))
    sr, cr = run(None)
    assert torch.equal(s2, sr) and torch.equal(c2, cr)
    # capture: forward through the image; the pack is part of the graph
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        cache_s = {}
        run(packed_weights(cache_s, 2 * L, n_geo, 4, pn, pd, pr))        # warm-up on the capture stream (allocations)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            sg, cg = run(packed_weights(cache_s, 2 * L, n_geo, 4, pn, pd, pr))
            sg2, cg2 = run(packed_weights(cache_s, 2 * L, n_geo, 4, pn, pd, pr))      # second use inside the capture: no second pack needed
        slot = next(iter(cache_s.values()))
        assert slot['key'] is None                                       # eager freshness dropped: the replays rewrite the image
        pd.mul_(0.75)
        graph.replay()
        torch.cuda.synchronize()
        sr, cr = run(None)
        assert torch.equal(sg, sr) and torch.equal(cg, cr) and torch.equal(sg2, sr) and torch.equal(cg2, cr)
        pn.mul_(0.9)
        s3, c3 = run(packed_weights(cache_s, 2 * L, n_geo, 4, pn, pd, pr))            # eager again: repacked
        sr, cr = run(None)
        assert torch.equal(s3, sr) and torch.equal(c3, cr)
    torch.cuda.current_stream().wait_stream(side)


def test_pack_weights_argument_validation():
    import ctypes
    from customnerf_amd._lib import lib, ptr
    need = ctypes.c_uint64(0)
    assert lib.cnerf_field_weight_image_bytes(32, 2, 4, ctypes.addressof(need)) == 0 and need.value == 2 * (64 * 32 + 3 * 4096 + 2048 + 64 * 96 + 2048)
    assert lib.cnerf_field_weight_image_bytes(33, 2, 4, ctypes.addressof(need)) < 0
    p = torch.zeros(8192, device='cuda')
    img = torch.empty(need.value, dtype=torch.uint8, device='cuda')
    assert lib.cnerf_field_pack_weights(32, 2, 4, ptr(p), ptr(p), ptr(p), ptr(img), need.value - 16, None) < 0         # image too small
    assert lib.cnerf_field_pack_weights(32, 2, 4, ptr(p), ptr(p), None, ptr(img), need.value, None) < 0                # NULL parameter vector
    assert lib.cnerf_field_pack_weights(32, 2, 4, ptr(p), ptr(p), ptr(p), img.data_ptr() + 8, need.value, None) < 0    # misaligned image
