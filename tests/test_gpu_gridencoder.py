"""GPU parity: grid encoder HIP kernels (Python surface -> ctypes -> C-ABI) against oracle/gridencoder_ref.c.
Forward is expected bit-exact in fp32 AND fp16 (same operation order, host-computed level geometry on both sides);
the scatter backward is compared with a tolerance because float atomics reorder the sums."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import c_oracle as co          # noqa: E402
from oracle import torch_oracle as to      # noqa: E402


def cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def make_inputs(B, D, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.random((B, D)).astype(np.float32)
    x[0] = 0.0            # exact lower corner
    x[1] = 1.0            # exact upper corner (pos_grid = resolution-1 .. +1)
    x[2, 0] = -0.01       # out of bounds -> zeros
    x[3, D - 1] = 1.001   # out of bounds
    x[4] = 0.5
    return x


CONFIGS = [
    # (name, ctor kwargs)
    ("hash_L16_T19", dict(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048, gridtype='hash')),
    ("tiled_L16_T19_8192", dict(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=8192, gridtype='tiled')),
    ("hash_L4", dict(input_dim=3, num_levels=4, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048, gridtype='hash')),
    ("hash_D2_C4", dict(input_dim=2, num_levels=8, level_dim=4, base_resolution=8, log2_hashmap_size=14, per_level_scale=1.5, gridtype='hash')),
    ("hash_D3_C1_align", dict(input_dim=3, num_levels=6, level_dim=1, base_resolution=4, log2_hashmap_size=12, per_level_scale=2, gridtype='hash', align_corners=True)),
    ("tiled_D3_C8_smooth", dict(input_dim=3, num_levels=5, level_dim=8, base_resolution=8, log2_hashmap_size=15, per_level_scale=1.7, gridtype='tiled', interpolation='smoothstep')),
    ("hash_D4_C2", dict(input_dim=4, num_levels=4, level_dim=2, base_resolution=4, log2_hashmap_size=13, per_level_scale=2, gridtype='hash')),
    # tiny tables: the tiled index drops a dimension once its running stride exceeds the table (gridencoder.cu:66-84) — 256 entries keep x, y;
    # 16 entries keep x only — and the hashed x-pair window of the specialised fp16 gather meets a table of 4 / 64 entries
    ("tiled_tiny_T8", dict(input_dim=3, num_levels=6, level_dim=2, base_resolution=8, log2_hashmap_size=8, per_level_scale=1.6, gridtype='tiled')),
    ("tiled_tiny_T4", dict(input_dim=3, num_levels=5, level_dim=2, base_resolution=8, log2_hashmap_size=4, per_level_scale=1.6, gridtype='tiled')),
    ("hash_tiny_T6", dict(input_dim=3, num_levels=6, level_dim=2, base_resolution=8, log2_hashmap_size=6, per_level_scale=1.6, gridtype='hash')),
    ("hash_tiny_T2", dict(input_dim=3, num_levels=4, level_dim=2, base_resolution=8, log2_hashmap_size=2, per_level_scale=1.6, gridtype='hash')),
]


def build(kw, scale=1.0, seed=0):
    from customnerf_amd.gridencoder import GridEncoder
    enc = GridEncoder(**kw).cuda()
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        enc.embeddings.copy_(((torch.rand(enc.embeddings.shape, generator=g) * 2 - 1) * scale).cuda())
    return enc


@pytest.mark.parametrize("name,kw", CONFIGS, ids=[c[0] for c in CONFIGS])
@pytest.mark.parametrize("half", [False, True], ids=["f32", "f16"])
def test_forward_bit_exact(name, kw, half):
    enc = build(kw)
    off_ref, pls_ref = to.grid_offsets(kw.get('input_dim', 3), kw['num_levels'], kw['level_dim'], kw.get('per_level_scale', 2),
                                       kw['base_resolution'], kw['log2_hashmap_size'], kw.get('desired_resolution'), kw.get('align_corners', False))
    np.testing.assert_array_equal(enc.offsets.cpu().numpy(), off_ref)
    B = 4099                                                  # ragged vs the 256-point block
    x = make_inputs(B, enc.input_dim)
    emb = enc.embeddings.detach().cpu().numpy()
    gid, iid = enc.gridtype_id, enc.interp_id
    ref, _ = co.grid_encode_forward(x, emb, off_ref, enc.per_level_scale, enc.base_resolution, False, gid, enc.align_corners, iid, None, half)
    # call the autograd function with inputs already in [0,1] (avoids the rounding of GridEncoder.forward's affine map)
    from customnerf_amd.gridencoder.grid import _grid_encode
    table = enc.half_table() if half else enc.embeddings.detach()
    out = _grid_encode.apply(cuda(x), enc.embeddings, table, enc._offsets_host, enc.per_level_scale, enc.base_resolution, False, gid,
                             enc.align_corners, iid, None)
    L, Bc, C = out.shape
    got = out.detach().permute(1, 0, 2).reshape(B, L * C).float().cpu().numpy()
    np.testing.assert_array_equal(got, ref)
    assert np.all(got[2] == 0) and np.all(got[3] == 0) and np.abs(got).max() > 0.1


def test_forward_max_level_and_empty():
    from customnerf_amd.gridencoder.grid import _grid_encode
    kw = CONFIGS[0][1]
    enc = build(kw)
    x = make_inputs(1000, 3, seed=3)
    ref, _ = co.grid_encode_forward(x, enc.embeddings.detach().cpu().numpy(), enc._offsets_host, enc.per_level_scale, 16, False, 0, False, 0, 5)
    out = _grid_encode.apply(cuda(x), enc.embeddings, enc.embeddings.detach(), enc._offsets_host, enc.per_level_scale, 16, False, 0, False, 0, 5)
    got = out.detach().permute(1, 0, 2).reshape(1000, -1).cpu().numpy()
    np.testing.assert_array_equal(got, ref)
    assert np.all(got[:, 10:] == 0)
    out = enc(torch.empty(0, 3).cuda())
    assert out.shape == (0, 32)


@pytest.mark.parametrize("name,kw", CONFIGS[:6], ids=[c[0] for c in CONFIGS[:6]])
@pytest.mark.parametrize("half", [False, True], ids=["f32", "f16"])
def test_dy_dx_and_input_grad(name, kw, half):
    from customnerf_amd.gridencoder.grid import _grid_encode
    enc = build(kw)
    B = 777
    x = make_inputs(B, enc.input_dim, seed=1)
    emb = enc.embeddings.detach().cpu().numpy()
    gid, iid = enc.gridtype_id, enc.interp_id
    ref, dy_ref = co.grid_encode_forward(x, emb, enc._offsets_host, enc.per_level_scale, enc.base_resolution, True, gid, enc.align_corners, iid, None, half)
    xin = cuda(x).requires_grad_(True)
    table = enc.half_table() if half else enc.embeddings.detach()
    out = _grid_encode.apply(xin, enc.embeddings, table, enc._offsets_host, enc.per_level_scale, enc.base_resolution, True, gid,
                             enc.align_corners, iid, None)
    L, _, C = out.shape
    g = np.random.default_rng(2).standard_normal((B, L * C)).astype(np.float32)
    if half:
        g = co.h2f(co.f2h(g))
    g_lbc = cuda(g).view(B, L, C).permute(1, 0, 2).contiguous().to(out.dtype)
    out.backward(g_lbc)
    # input gradient: sum_l,c grad * dy_dx  (gridencoder.cu:342-368)
    ge_ref, gi_ref = co.grid_encode_backward(g, x, emb.shape, enc._offsets_host, enc.per_level_scale, enc.base_resolution, dy_ref, gid,
                                             enc.align_corners, iid)
    scale = max(1.0, float(np.abs(gi_ref).max()))
    np.testing.assert_allclose(xin.grad.cpu().numpy(), gi_ref, rtol=2e-3 if half else 1e-5, atol=(2e-3 if half else 1e-5) * scale)
    np.testing.assert_allclose(enc.embeddings.grad.cpu().numpy(), ge_ref, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name,kw", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_backward_scatter(name, kw):
    enc = build(kw)
    B = 20011
    x = make_inputs(B, enc.input_dim, seed=4)
    L, C = enc.num_levels, enc.level_dim
    g = np.random.default_rng(5).standard_normal((B, L * C)).astype(np.float32)
    ge_ref, _ = co.grid_encode_backward(g, x, tuple(enc.embeddings.shape), enc._offsets_host, enc.per_level_scale, enc.base_resolution, None,
                                        enc.gridtype_id, enc.align_corners, enc.interp_id)
    for half in (False, True):
        enc.embeddings.grad = None
        out = enc.encode(cuda(x) * 2 - 1, bound=1, half=half)
        gg = co.h2f(co.f2h(g)) if half else g
        if half:
            ge_ref_h, _ = co.grid_encode_backward(gg, x, tuple(enc.embeddings.shape), enc._offsets_host, enc.per_level_scale,
                                                  enc.base_resolution, None, enc.gridtype_id, enc.align_corners, enc.interp_id)
        out.backward(cuda(gg).view(B, L, C).permute(1, 0, 2).contiguous().to(out.dtype))
        got = enc.embeddings.grad.cpu().numpy()
        ref = ge_ref_h if half else ge_ref
        # the affine map (x*2-1+1)/2 may move a point by 1 ulp: compare with a tolerance scaled to the accumulated magnitude
        np.testing.assert_allclose(got, ref, rtol=1e-3, atol=2e-3)
        assert np.abs(got).max() > 1.0
        assert got.dtype == np.float32


BINNED_CONFIGS = [CONFIGS[0], CONFIGS[1],
                  ("hash_L16_T19_smooth_align", dict(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19,
                                                     desired_resolution=2048, gridtype='hash', align_corners=True, interpolation='smoothstep')),
                  ("tiled_L12_T17_odd", dict(input_dim=3, num_levels=12, level_dim=2, base_resolution=12, log2_hashmap_size=17,
                                             per_level_scale=1.38, gridtype='tiled')),
                  # the reference's bear table (tiled, T = 2^21, desired 8192): 512 bins per level -> the WIDE form of the histogram-free emit (round 6)
                  ("tiled_L16_T21_8192_bear", dict(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=21,
                                                   desired_resolution=8192, gridtype='tiled')),
                  # align_corners on a wide table (256 bins per level): points on the upper boundary have a
                  # corner one grid line past the level — the dense index must wrap into the level as the reference's `% hashmap_size` does
                  # (an unwrapped index used an uninitialised bin cursor there and overwrote some other record)
                  ("hash_L16_T20_align", dict(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=20,
                                              desired_resolution=1024, gridtype='hash', align_corners=True)),
                  # hashed levels smaller than one 4096-entry bin: single-bin levels of the third form (round 6; the second form, which used to serve
                  # them — and the align_corners tables above, whose dense levels have odd sizes — was removed)
                  ("hash_L16_T10_small", dict(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=10,
                                              desired_resolution=512, gridtype='hash'))]


@pytest.mark.parametrize("half", [False, True], ids=["f32", "f16"])
@pytest.mark.parametrize("name,kw", BINNED_CONFIGS, ids=[c[0] for c in BINNED_CONFIGS])
def test_backward_binned_no_atomics(name, kw, half):
    """Large scatters take the atomic-free binned path (partition by table chunk -> LDS sums -> plain stores); it must
    agree with the oracle AND with the atomic kernel, accumulate into a pre-filled gradient table, and skip OOB points."""
    from customnerf_amd.gridencoder import grid as G
    enc = build(kw)
    B = 70001 if enc.num_levels >= 16 else 100003   # x levels > 2^20 (point, level) pairs -> binned path; ragged vs the point blocks
    x = make_inputs(B, 3, seed=11)
    x[100:200] = x[50]                         # many samples in one cell (coarse-level style duplication)
    L, C = enc.num_levels, enc.level_dim
    g = np.random.default_rng(12).standard_normal((B, L * C)).astype(np.float32)
    if half:
        g = co.h2f(co.f2h(g))
    ge_ref, _ = co.grid_encode_backward(g, x, tuple(enc.embeddings.shape), enc._offsets_host, enc.per_level_scale, enc.base_resolution, None,
                                        enc.gridtype_id, enc.align_corners, enc.interp_id)
    import ctypes
    need = ctypes.c_uint64(0)
    S = float(np.log2(enc.per_level_scale))
    G.lib.cnerf_grid_encode_backward_workspace_bytes(enc._offsets_host.ctypes.data, B, 3, C, L, L, S, enc.base_resolution, int(half), ctypes.addressof(need))
    assert need.value > 0
    table = enc.half_table() if half else enc.embeddings.detach()
    out = G._grid_encode.apply(cuda(x), enc.embeddings, table, enc._offsets_host, enc.per_level_scale, enc.base_resolution, False,
                               enc.gridtype_id, enc.align_corners, enc.interp_id, None)
    glbc = cuda(g).view(B, L, C).permute(1, 0, 2).contiguous().to(out.dtype)
    out.backward(glbc)
    got = enc.embeddings.grad.cpu().numpy()
    tol = 2e-3 if half else 1e-4                # fp16: each w*g product is rounded to half (as gridencoder.cu:328 does)
    np.testing.assert_allclose(got, ge_ref, rtol=tol, atol=tol * 10)
    # same launch through the C-ABI with workspace=NULL -> atomic kernel; and accumulation into a non-zero table
    from customnerf_amd._lib import lib, ptr, stream, check
    ga = torch.full(enc.embeddings.shape, 0.5, device='cuda')
    check(lib.cnerf_grid_encode_backward(ptr(glbc), ptr(cuda(x)), enc._offsets_host.ctypes.data, ptr(ga), B, 3, C, L, L, S, enc.base_resolution,
                                         None, None, enc.gridtype_id, int(enc.align_corners), enc.interp_id, int(half), None, 0, stream()))
    gb = torch.full(enc.embeddings.shape, 0.5, device='cuda')
    ws = torch.empty(int(need.value) + 256, dtype=torch.uint8, device='cuda')
    check(lib.cnerf_grid_encode_backward(ptr(glbc), ptr(cuda(x)), enc._offsets_host.ctypes.data, ptr(gb), B, 3, C, L, L, S, enc.base_resolution,
                                         None, None, enc.gridtype_id, int(enc.align_corners), enc.interp_id, int(half), ptr(ws), ws.numel(), stream()))
    np.testing.assert_allclose(gb.cpu().numpy(), ga.cpu().numpy(), rtol=tol, atol=tol * 10)
    np.testing.assert_allclose(gb.cpu().numpy() - 0.5, ge_ref, rtol=tol, atol=tol * 10)
    if half:
        # fp16 records are summed as 64-bit fixed point, split bins included: the scatter is bit-reproducible whatever order the records land in
        gc = torch.full(enc.embeddings.shape, 0.5, device='cuda')
        check(lib.cnerf_grid_encode_backward(ptr(glbc), ptr(cuda(x)), enc._offsets_host.ctypes.data, ptr(gc), B, 3, C, L, L, S, enc.base_resolution,
                                             None, None, enc.gridtype_id, int(enc.align_corners), enc.interp_id, int(half), ptr(ws), ws.numel(), stream()))
        gd = torch.full(enc.embeddings.shape, 0.5, device='cuda')
        check(lib.cnerf_grid_encode_backward(ptr(glbc), ptr(cuda(x)), enc._offsets_host.ctypes.data, ptr(gd), B, 3, C, L, L, S, enc.base_resolution,
                                             None, None, enc.gridtype_id, int(enc.align_corners), enc.interp_id, int(half), ptr(ws), ws.numel(), stream()))
        assert torch.equal(gc, gd)


def test_scatter_skips_zero_rows_and_recomputes_overflowing_bins():
    """Round 5 scatter (no histogram pre-pass: block-local counting, one run reservation per (block, bin) in fixed-capacity bin regions on the
    hashed levels, run tables on the dense ones).  (a) Rows whose gradient is exactly zero emit no records: the table gradient is BIT-identical to
    the scatter of the list with those rows removed.  (b) A sample distribution that defeats the hash — every sample in one tiny cube, so a few bins
    of each hashed level receive all records and overflow their regions — is recomputed exactly by the fallback sweep: same result as the atomic
    kernel, and the same bits on every run."""
    from customnerf_amd._lib import lib, ptr, stream, check
    import ctypes
    enc = build(CONFIGS[0][1])
    L, C = enc.num_levels, enc.level_dim
    S = float(np.log2(enc.per_level_scale))

    def scatter(x, g_lbc, binned=True):
        B = x.shape[0]
        need = ctypes.c_uint64(0)
        lib.cnerf_grid_encode_backward_workspace_bytes(enc._offsets_host.ctypes.data, B, 3, C, L, L, S, enc.base_resolution, 1, ctypes.addressof(need))
        assert need.value > 0
        ws = torch.empty(int(need.value) + 256, dtype=torch.uint8, device='cuda') if binned else None
        out = torch.zeros(enc.embeddings.shape, device='cuda')
        check(lib.cnerf_grid_encode_backward(ptr(g_lbc), ptr(x), enc._offsets_host.ctypes.data, ptr(out), B, 3, C, L, L, S, enc.base_resolution, None, None,
                                             enc.gridtype_id, int(enc.align_corners), enc.interp_id, 1, ptr(ws), ws.numel() if binned else 0, stream()))
        return out
    # (a)
    B = 300001
    rng = np.random.default_rng(21)
    x = cuda(make_inputs(B, 3, seed=22))
    g = torch.from_numpy(rng.standard_normal((L, B, C)).astype(np.float32)).cuda().half()
    dead = torch.from_numpy(rng.random(B) < 0.7).cuda()
    g[:, dead] = 0
    full = scatter(x, g.contiguous())
    keep = (~dead).nonzero().squeeze(1)
    compact = scatter(x[keep].contiguous(), g[:, keep].contiguous())
    assert int(keep.numel()) * L >= (1 << 20)                     # the compacted list still takes the binned path
    assert torch.equal(full, compact)
    assert float(full.abs().max()) > 1.0
    # (b)
    B = 120001
    xb = (torch.rand(B, 3, device='cuda', generator=torch.Generator(device='cuda').manual_seed(3)) * (1.0 / 2048) + 0.3137).contiguous()
    gb = torch.from_numpy(rng.standard_normal((L, B, C)).astype(np.float32)).cuda().half().contiguous()
    r1 = scatter(xb, gb)
    r2 = scatter(xb, gb)
    ra = scatter(xb, gb, binned=False)                             # float atomics: order-dependent rounding, so a tolerance scaled to the sums
    assert torch.equal(r1, r2)
    scale = float(ra.abs().max())
    assert scale > 100.0 and float((r1 - ra).abs().max()) < 2e-3 * scale


@pytest.mark.parametrize("max_level", [1, 6, 9])
def test_binned_scatter_with_fewer_active_levels(max_level):
    """coarse-to-fine training hands `max_level < L` to the backward (grid.py: only the first max_level levels receive gradient): the histogram-free
    scatter plans its slots for those levels only — same table gradient as the atomic kernel, the inactive levels' rows untouched, bit-reproducible"""
    from customnerf_amd._lib import lib, ptr, stream, check
    import ctypes
    enc = build(CONFIGS[0][1])
    L, C = enc.num_levels, enc.level_dim
    S = float(np.log2(enc.per_level_scale))
    B = 90001
    x = cuda(make_inputs(B, 3, seed=31))
    g = torch.from_numpy(np.random.default_rng(32).standard_normal((L, B, C)).astype(np.float32)).cuda().half().contiguous()
    need = ctypes.c_uint64(0)
    lib.cnerf_grid_encode_backward_workspace_bytes(enc._offsets_host.ctypes.data, B, 3, C, L, max_level, S, enc.base_resolution, 1, ctypes.addressof(need))

    def scatter(ws):
        out = torch.full(enc.embeddings.shape, 0.25, device='cuda')
        check(lib.cnerf_grid_encode_backward(ptr(g), ptr(x), enc._offsets_host.ctypes.data, ptr(out), B, 3, C, L, max_level, S, enc.base_resolution, None, None,
                                             enc.gridtype_id, int(enc.align_corners), enc.interp_id, 1, ptr(ws), ws.numel() if ws is not None else 0, stream()))
        return out
    atomic = scatter(None)
    if need.value == 0:                                    # too little work for the binned path at this level count: nothing more to compare
        return
    ws = torch.empty(int(need.value) + 256, dtype=torch.uint8, device='cuda')
    a, b = scatter(ws), scatter(ws)
    assert torch.equal(a, b)
    np.testing.assert_allclose(a.cpu().numpy(), atomic.cpu().numpy(), rtol=2e-3, atol=2e-2)
    first_inactive = int(enc._offsets_host[max_level])
    assert torch.all(a[first_inactive:] == 0.25) and not torch.all(a[:first_inactive] == 0.25)


def test_grad_total_variation():
    kw = CONFIGS[0][1]
    enc = build(kw, scale=0.5, seed=7)
    B = 5000
    x = make_inputs(B, 3, seed=8)
    emb = enc.embeddings.detach().cpu().numpy()
    grad = np.zeros_like(emb)
    co.grad_total_variation(x, emb, grad, enc._offsets_host, 1e-3, enc.per_level_scale, 16, 0, False)
    enc.embeddings.grad = torch.zeros_like(enc.embeddings)
    enc.grad_total_variation(weight=1e-3, inputs=cuda(x) * 2 - 1, bound=1)
    np.testing.assert_allclose(enc.embeddings.grad.cpu().numpy(), grad, rtol=1e-3, atol=1e-6)
    assert np.abs(grad).max() > 0


def test_cast_and_half_table_refresh():
    kw = CONFIGS[2][1]
    enc = build(kw, scale=3.0)
    h = enc.half_table()
    assert torch.equal(h, enc.embeddings.detach().half())
    ref = co.f2h(enc.embeddings.detach().cpu().numpy())
    np.testing.assert_array_equal(h.cpu().view(torch.int16).numpy().view(np.uint16), ref)       # RNE, bit-exact vs the oracle's f2h
    with torch.no_grad():
        enc.embeddings.mul_(0.5)                                                                # bumps the version counter
    assert torch.equal(enc.half_table(), enc.embeddings.detach().half())


def test_rejects_bad_arguments():
    from customnerf_amd.gridencoder import GridEncoder
    with pytest.raises(ValueError):
        GridEncoder(level_dim=3)
    with pytest.raises(ValueError):
        GridEncoder(input_dim=6)
    enc = build(CONFIGS[2][1])
    with pytest.raises(RuntimeError):
        enc(torch.rand(8, 3))           # CPU tensor: no CPU path


@pytest.mark.gpu
@pytest.mark.parametrize("bad", [float("inf"), float("-inf"), float("nan")])
def test_backward_propagates_non_finite_gradients(bad):
    """The loss scaler detects overflow by looking for inf / NaN in the gradients (GradScaler; the reference's half2 atomics propagate them,
    gridencoder.cu:324-337).  The binned fp16 backward sums in fixed point, which cannot carry them: a non-finite incoming gradient must still
    leave a non-finite entry in the table gradient."""
    kw = CONFIGS[0][1]
    enc = build(kw)
    B = 70001                                                 # B x levels >= 2^20: the binned (fixed-point) path, not the atomic kernel
    x = torch.from_numpy(make_inputs(B, 3)).cuda()
    with torch.autocast('cuda', dtype=torch.float16):
        out = enc(x * 2 - 1, bound=1.0)
    assert out.dtype == torch.float16
    g = torch.randn(out.shape, device='cuda').half() * 1e-3
    g[1234, 17] = bad
    out.backward(g)
    grad = enc.embeddings.grad
    assert not bool(torch.isfinite(grad).all()), "a non-finite output gradient vanished in the scatter"
    # and a clean gradient stays clean
    enc.embeddings.grad = None
    with torch.autocast('cuda', dtype=torch.float16):
        out = enc(x * 2 - 1, bound=1.0)
    out.backward(torch.randn(out.shape, device='cuda').half() * 1e-3)
    assert bool(torch.isfinite(enc.embeddings.grad).all())


@pytest.mark.parametrize("name,kw", [c for c in CONFIGS if c[1].get('input_dim', 3) == 3 and c[1]['level_dim'] == 2], ids=[c[0] for c in CONFIGS if c[1].get('input_dim', 3) == 3 and c[1]['level_dim'] == 2])
@pytest.mark.parametrize("B,row0", [(4099, 0), (1 << 16, 256), (40961, 7), (5, 0), (255, 3)])
def test_sample_major_traversal_is_bit_identical(name, kw, B, row0):
    """the sample-major gather (cnerf_grid_encode_forward_ordered, traversal 1 — what the TraversalTuner may pick for the
    importance pass) must write exactly what the level-major kernel writes: ragged sizes, a row offset into a larger feature buffer,
    out-of-range and boundary samples, every fp16 D = 3 / C = 2 table of the suite (the kernel is specialised; other tables ignore the switch)."""
    from customnerf_amd.gridencoder.grid import LEVEL_MAJOR, SAMPLE_MAJOR, TraversalTuner
    enc = build(kw)
    x = cuda(make_inputs(B, 3, seed=B))
    L = enc.num_levels
    P = row0 + B + 5
    outs = []
    for trav in (LEVEL_MAJOR, SAMPLE_MAJOR):
        out = torch.full((L, P, 2), 7.0, dtype=torch.float16, device="cuda")
        enc.encode_into(x, out, row0, half=True, traversal=trav)
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    assert torch.all(outs[1][:, :row0] == 7.0) and torch.all(outs[1][:, row0 + B:] == 7.0)          # nothing outside the rows asked for
    assert not torch.isnan(outs[1].float()).any()
    # a tuner's trial call runs both forms back to back into the same rows: the result is the same buffer again
    tuner = TraversalTuner(first=0, period=4)
    for _ in range(6):
        out = torch.full((L, P, 2), 7.0, dtype=torch.float16, device="cuda")
        enc.encode_into(x, out, row0, half=True, traversal=tuner)
        assert torch.equal(out, outs[0])
    torch.cuda.synchronize()
    tuner.plan()
    assert len(tuner.history) >= 1 and tuner.choice in (LEVEL_MAJOR, SAMPLE_MAJOR)


# ---- the table's optimiser step inside the backward scatter (ABI 5: cnerf_grid_backward_adam) ----
def _adam_reference_and_fused(enc, x, g_lbc, scaler_state, p_half, max_level=None, lr=1e-2, zero_grad=1, prefill=None):
    """-> (reference: backward into g, then cnerf_adam_step_scaled), (fused: the same backward with the step armed), consumed"""
    from customnerf_amd._lib import lib, ptr, stream, check, GridAdam
    import ctypes
    L, C = enc.num_levels, enc.level_dim
    ml = L if max_level is None else max_level
    S = float(np.log2(enc.per_level_scale))
    B = x.shape[0]
    need = ctypes.c_uint64(0)
    lib.cnerf_grid_encode_backward_workspace_bytes(enc._offsets_host.ctypes.data, B, 3, C, L, ml, S, enc.base_resolution, 1, ctypes.addressof(need))
    ws = torch.empty(int(need.value) + 256, dtype=torch.uint8, device='cuda')
    gen = torch.Generator(device='cuda').manual_seed(5)
    m0 = torch.rand(enc.embeddings.shape, device='cuda', generator=gen) * 1e-3
    v0 = torch.rand(enc.embeddings.shape, device='cuda', generator=gen) * 1e-6
    outs = []
    consumed = None
    for fused in (False, True):
        p = enc.embeddings.detach().clone()
        m, v = m0.clone(), v0.clone()
        g = torch.zeros_like(p) if prefill is None else prefill.clone()
        ph = torch.zeros(p.shape, dtype=torch.half, device='cuda') if p_half else None
        st = scaler_state.clone()
        if fused:
            cfg = GridAdam()
            cfg.p, cfg.g, cfg.m, cfg.v = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
            cfg.p_half = ph.data_ptr() if ph is not None else None
            cfg.n = p.numel()
            cfg.lr, cfg.beta1, cfg.beta2, cfg.eps = lr, 0.9, 0.99, 1e-15
            cfg.scaler_state, cfg.extra_inv, cfg.zero_grad = st.data_ptr(), 1.0, zero_grad
            check(lib.cnerf_grid_backward_adam(ctypes.addressof(cfg)))
        check(lib.cnerf_grid_encode_backward(ptr(g_lbc), ptr(x), enc._offsets_host.ctypes.data, ptr(g), B, 3, C, L, ml, S, enc.base_resolution, None, None,
                                             enc.gridtype_id, int(enc.align_corners), enc.interp_id, 1, ptr(ws), ws.numel(), stream()))
        if fused:
            done = ctypes.c_int(0)
            check(lib.cnerf_grid_backward_adam_consumed(ctypes.addressof(done)))
            check(lib.cnerf_grid_backward_adam(None))
            consumed = bool(done.value)
        if not fused or not consumed:
            check(lib.cnerf_adam_step_scaled(ptr(p), ptr(g), ptr(m), ptr(v), ptr(ph), p.numel(), lr, 0.9, 0.99, 1e-15, ptr(st), 1.0, zero_grad, stream()))
        torch.cuda.synchronize()
        outs.append((p, m, v, g, ph))
    return outs[0], outs[1], consumed


@pytest.mark.parametrize("case", ["uniform", "corner", "overflow_skip", "accumulated_grad", "no_shadow_keep_grad"])
def test_table_adam_inside_the_scatter_is_bit_identical(case):
    """cnerf_grid_backward_adam: the Adam update applied by the scatter's flush (sole-owner bins, the split bins' reduction, bins without a single record)
    leaves parameter, moments, gradient table and fp16 shadow exactly as the backward pass followed by cnerf_adam_step_scaled does — uniform samples,
    samples confined to a corner of the volume (most bins of the dense levels see no record; a few are crowded and split), a step the loss scaler
    skips (found_inf set: no update, gradients cleared), a gradient table that already holds a contribution, and without shadow / without clearing"""
    enc = build(CONFIGS[0][1], scale=0.5)
    L, C = enc.num_levels, enc.level_dim
    B = 150001
    rng = np.random.default_rng(31)
    x = make_inputs(B, 3, seed=32)
    if case == "corner":
        x = x * 0.07 + 0.01
    x = cuda(x)
    g = torch.from_numpy(rng.standard_normal((L, B, C)).astype(np.float32)).cuda().half().contiguous()
    state = torch.tensor([128.0, 3.0, 1.0 if case == "overflow_skip" else 0.0, 17.0], device='cuda')
    prefill = None
    if case == "accumulated_grad":
        prefill = torch.from_numpy(rng.standard_normal(tuple(enc.embeddings.shape)).astype(np.float32)).cuda()
    ref, fus, consumed = _adam_reference_and_fused(enc, x, g, state, p_half=case != "no_shadow_keep_grad", zero_grad=0 if case == "no_shadow_keep_grad" else 1,
                                                   prefill=prefill)
    assert consumed
    names = ("p", "m", "v", "g", "p_half")
    for n, a, b in zip(names, ref, fus):
        if a is None:
            assert b is None
            continue
        assert torch.equal(a, b), n
    p_ref = ref[0]
    if case == "overflow_skip":
        assert torch.equal(p_ref, enc.embeddings.detach()) and float(ref[3].abs().max()) == 0.0
    else:
        assert not torch.equal(p_ref, enc.embeddings.detach())
        assert torch.equal(ref[4], p_ref.half()) if ref[4] is not None else float(ref[3].abs().max()) > 0.0


def test_table_adam_is_not_applied_by_a_partial_or_foreign_backward():
    """the armed step is one shot and all-or-nothing: a backward pass over fewer levels than the table has (max_level), or into another gradient
    table, applies nothing and reports so — the caller's own optimiser launch then does the update"""
    from customnerf_amd._lib import lib, ptr, stream, check, GridAdam
    import ctypes
    enc = build(CONFIGS[0][1], scale=0.5)
    L, C = enc.num_levels, enc.level_dim
    B = 100001
    rng = np.random.default_rng(41)
    x = cuda(make_inputs(B, 3, seed=42))
    g = torch.from_numpy(rng.standard_normal((L, B, C)).astype(np.float32)).cuda().half().contiguous()
    state = torch.tensor([64.0, 0.0, 0.0, 5.0], device='cuda')
    ref, fus, consumed = _adam_reference_and_fused(enc, x, g, state, p_half=True, max_level=12)
    assert consumed is False
    for a, b in zip(ref, fus):
        assert torch.equal(a, b)
    # armed for another table: this backward leaves it armed and untouched, the explicit disarm clears it
    other = torch.zeros_like(enc.embeddings)
    cfg = GridAdam()
    cfg.p, cfg.g, cfg.m, cfg.v, cfg.p_half, cfg.n = other.data_ptr(), other.data_ptr(), other.data_ptr(), other.data_ptr(), None, other.numel()
    cfg.lr, cfg.beta1, cfg.beta2, cfg.eps, cfg.scaler_state, cfg.extra_inv, cfg.zero_grad = 1e-2, 0.9, 0.99, 1e-15, state.data_ptr(), 1.0, 1
    check(lib.cnerf_grid_backward_adam(ctypes.addressof(cfg)))
    ref2, fus2, consumed2 = None, None, None
    S = float(np.log2(enc.per_level_scale))
    need = ctypes.c_uint64(0)
    lib.cnerf_grid_encode_backward_workspace_bytes(enc._offsets_host.ctypes.data, B, 3, C, L, L, S, enc.base_resolution, 1, ctypes.addressof(need))
    ws = torch.empty(int(need.value) + 256, dtype=torch.uint8, device='cuda')
    out = torch.zeros_like(enc.embeddings)
    check(lib.cnerf_grid_encode_backward(ptr(g), ptr(x), enc._offsets_host.ctypes.data, ptr(out), B, 3, C, L, L, S, enc.base_resolution, None, None,
                                         enc.gridtype_id, int(enc.align_corners), enc.interp_id, 1, ptr(ws), ws.numel(), stream()))
    done = ctypes.c_int(1)
    check(lib.cnerf_grid_backward_adam_consumed(ctypes.addressof(done)))
    check(lib.cnerf_grid_backward_adam(None))
    assert done.value == 0 and float(other.abs().max()) == 0.0 and float(out.abs().max()) > 0.0
    # argument validation
    bad = GridAdam()
    assert lib.cnerf_grid_backward_adam(ctypes.addressof(bad)) < 0                       # NULL pointers
    cfg.n = other.numel() + 2
    assert lib.cnerf_grid_backward_adam(ctypes.addressof(cfg)) < 0                       # n not a multiple of 4
