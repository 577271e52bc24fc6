"""CPU: the C restatement of the two CUDA kernels (oracle/gridencoder_ref.c, oracle/raymarching_ref.c — which cannot be pinned by running
the reference here) against the independent numpy restatement oracle/np_restatement.py.  Two separately written statements of the same
source that agree bit for bit (index arithmetic, float32 / binary16 accumulation order, the march's double promotions) leave little
room for a transcription slip in either."""
import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import np_restatement as nr
from oracle import torch_oracle as to


def test_fma32_is_single_rounding():
    rng = np.random.default_rng(0)
    a, b, c = (rng.standard_normal(20000).astype(np.float32) for _ in range(3))
    from fractions import Fraction
    out = nr.fma32(a, b, c)
    for i in range(0, 20000, 97):
        exact = Fraction(float(a[i])) * Fraction(float(b[i])) + Fraction(float(c[i]))
        lo, hi = np.nextafter(out[i], np.float32(-np.inf)), np.nextafter(out[i], np.float32(np.inf))
        assert abs(Fraction(float(out[i])) - exact) <= min(abs(Fraction(float(lo)) - exact), abs(Fraction(float(hi)) - exact))
    # a constructed float32 tie of the float64 sum: 1 + 2^-24 + tiny -> must round UP (exact), not to even
    one, eps = np.float32(1.0), np.float32(2.0 ** -24)
    assert nr.fma32(np.float32(2.0 ** -60), np.float32(1.0), np.float32(1.0)) == one
    assert nr.fma32(eps, one, one) == one                                                  # exact tie -> even
    assert nr.fma32(np.float32(1 + 2.0 ** -23), eps, one) == np.nextafter(one, np.float32(2))   # just above the tie


CFGS = {
    "hash_L16_T19": dict(num_levels=16, log2_hashmap_size=19, desired_resolution=2048, gridtype='hash'),
    "tiled_L16_T21_8192": dict(num_levels=16, log2_hashmap_size=21, desired_resolution=8192, gridtype='tiled'),
    "hash_L4_T8_tiny": dict(num_levels=4, log2_hashmap_size=8, desired_resolution=64, gridtype='hash'),
    "hash_L6_align": dict(num_levels=6, log2_hashmap_size=14, desired_resolution=256, gridtype='hash', align_corners=True),
}


@pytest.mark.parametrize("name", list(CFGS))
@pytest.mark.parametrize("half", [False, True], ids=["f32", "f16"])
def test_grid_encode_forward_two_restatements_agree(name, half):
    c = dict(CFGS[name])
    ac = c.pop("align_corners", False)
    offsets, pls = to.grid_offsets(3, c["num_levels"], 2, 2, 16, c["log2_hashmap_size"], c["desired_resolution"], align_corners=ac)[:2]
    offsets = np.asarray(offsets, np.int64)
    rng = np.random.default_rng(1)
    B = 700
    x = rng.random((B, 3)).astype(np.float32)
    x[:5] = [[0, 0, 0], [1, 1, 1], [1, 0, 0.5], [-0.1, 0.5, 0.5], [0.5, 1.0001, 0.5]]            # boundary corners and two out-of-range rows
    emb = ((rng.random((int(offsets[-1]), 2)) * 2 - 1)).astype(np.float32)
    gt = 0 if c["gridtype"] == 'hash' else 1
    ref, _ = co.grid_encode_forward(x, emb, offsets, pls, 16, gridtype=gt, align_corners=ac, half=half)
    table = emb.astype(np.float16) if half else emb
    out = nr.grid_encode_forward(x, table, offsets, float(np.log2(pls)), 16, gt, ac)
    out = out.astype(np.float32).transpose(1, 0, 2).reshape(B, -1)
    np.testing.assert_array_equal(out, ref)
    assert np.all(out[3:5] == 0) and np.abs(out[5:]).max() > 0.1


def test_grid_encode_backward_two_restatements_agree():
    offsets, pls = to.grid_offsets(3, 8, 2, 2, 16, 12, 256)[:2]
    offsets = np.asarray(offsets, np.int64)
    rng = np.random.default_rng(2)
    B = 400
    x = rng.random((B, 3)).astype(np.float32)
    x[0] = [1.5, 0.5, 0.5]
    grad = rng.standard_normal((B, 16)).astype(np.float32)
    ref, _ = co.grid_encode_backward(grad, x, (int(offsets[-1]), 2), offsets, pls, 16)
    out = nr.grid_encode_backward(np.ascontiguousarray(grad.reshape(B, 8, 2).transpose(1, 0, 2)), x, offsets, float(np.log2(pls)), 16, int(offsets[-1]))
    # both add float32 products in point order per level; numpy's add.at visits duplicates in index order too
    np.testing.assert_allclose(out, ref, rtol=1e-6, atol=1e-6)
    assert np.abs(ref).max() > 0.5


@pytest.mark.parametrize("dt_gamma,seed", [(0.0, 0), (1.0 / 128, 1)])
def test_march_rays_train_two_restatements_agree(dt_gamma, seed):
    from customnerf_amd import scene as sc
    H, C, bound, max_steps = 32, 2, 2.0, 256
    rng = np.random.default_rng(seed)
    grid = (rng.random((C, H ** 3)) < 0.08).astype(np.float32) * 20.0
    grid[:, :64] = 20.0
    bitfield = co.packbits(grid, 10.0)
    c2w = sc.poses(4)[seed]
    o, d = to.generate_rays(__import__("torch").from_numpy(c2w[None]), *sc.intrinsics(12, 12), 12, 12)
    o, d = o.reshape(-1, 3).numpy(), d.reshape(-1, 3).numpy()
    d[3] = [0.0, 0.0, -1.0]                                                                  # an axis-aligned ray: 1/0 in the voxel-skip distances
    o[3] = [0.1, 0.2, 3.0]
    nears, fars = co.near_far_from_aabb(o, d, np.array([-bound] * 3 + [bound] * 3, np.float32), 0.2)
    noises = rng.random(o.shape[0]).astype(np.float32)
    counter = np.zeros(2, np.int32)
    xyzs, dirs, deltas, rays = co.march_rays_train(o, d, bound, bitfield, C, H, nears, fars, counter, -1, noises, -1, True, dt_gamma, max_steps)
    counts, pts, dls = nr.march_rays_train(o, d, bitfield, bound, dt_gamma, max_steps, C, H, nears, fars, noises)
    assert counts.sum() > 200 and (counts == 0).any()
    order = np.argsort(rays[:, 0])
    np.testing.assert_array_equal(rays[order, 2], counts)
    for n in range(o.shape[0]):
        r = rays[order[n]]
        if r[2] == 0:
            continue
        np.testing.assert_array_equal(xyzs[r[1]:r[1] + r[2]], pts[n], err_msg=f"ray {n} positions")
        np.testing.assert_array_equal(deltas[r[1]:r[1] + r[2]], dls[n], err_msg=f"ray {n} deltas")
