"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product path).

A SECOND, independent restatement of the two CUDA kernels whose C restatement (gridencoder_ref.c, raymarching_ref.c) cannot be pinned by
running the reference (no nvcc / no NVIDIA GPU here — DESIGN.md §2): written in numpy from the reference sources with a different
structure (vectorised over points / one Python loop per ray, arithmetic in numpy float32 scalars), so that a transcription slip in either
restatement shows up as a disagreement in tests/test_oracle_independent.py.

  grid_encode_forward   kernel_grid            gridencoder/src/gridencoder.cu:87-200 (+ get_grid_index :66-84, fast_hash :50-63)
  grid_encode_backward  kernel_grid_backward   gridencoder/src/gridencoder.cu:247-339 (float32 accumulation of the w * grad products)
  march_rays_train      kernel_march_rays_train raymarching/src/raymarching.cu:311-480 (helpers :30-81), ray-ordered slots

Shared modelling decision (both restatements, stated so it can be challenged): nvcc contracts `a * b + c` into one fused multiply-add
(its default -fmad=true), so `inputs[d] * scale + 0.5f`, `results += w * g`, `ox + t * dx` etc. are single-rounding FMAs; expressions with a
double literal (`0.5 * (x * mip_rbound + 1) * H`, `dt * H * 0.5`) are evaluated in double from the point where the literal enters.
"""
from fractions import Fraction

import numpy as np

f32 = np.float32
PRIMES = np.array([1, 2654435761, 805459861, 3674653429, 2097192037, 1434869437, 2165219737], dtype=np.uint64)


def fma32(a, b, c):
    """fmaf(a, b, c) on float32 arrays / scalars: the product of two float32 is exact in float64; the float64 sum may round, which can only
    matter when it lands exactly on a float32 rounding tie — those (rare) elements are redone in exact rational arithmetic."""
    a, b, c = np.broadcast_arrays(np.asarray(a, f32), np.asarray(b, f32), np.asarray(c, f32))
    s = a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)
    out = s.astype(f32)
    bits = s.view(np.uint64) if s.ndim else np.array(s).view(np.uint64)
    tie = (bits & np.uint64(0x1FFFFFFF)) == np.uint64(0x10000000)
    if np.any(tie):
        out = np.array(out, copy=True)
        for i in np.argwhere(np.atleast_1d(tie)):
            i = tuple(i) if out.ndim else ()
            exact = Fraction(float(a[i])) * Fraction(float(b[i])) + Fraction(float(c[i]))
            lo, hi = np.nextafter(f32(s[i]), f32(-np.inf)), np.nextafter(f32(s[i]), f32(np.inf))
            best = min((f32(s[i]), lo, hi), key=lambda v: (abs(Fraction(float(v)) - exact), int(np.array(v).view(np.uint32)) & 1))
            out[i] = best
    return out if out.ndim else f32(out)


def level_geometry(level, S, H):
    """exp2f(level * S) * H - 1.0f (:138-139).  exp2f here = the correctly rounded float32 value (double-precision power, rounded once):
    what glibc's exp2f returns (the C restatement and the HIP library's host side use it) in all but vanishingly rare cases.  numpy's own
    float32 exp2 is 1 ulp off at several levels of the L16 / 2048 configuration — and so may CUDA's device exp2f be (documented 2 ulp):
    the last bit of a level's scale is NOT pinned by the reference's source."""
    e = f32(f32(level) * f32(S))
    scale = f32(f32(2.0 ** float(e)) * f32(H)) - f32(1.0)
    resolution = np.uint32(np.ceil(scale)) + np.uint32(1)
    return f32(scale), int(resolution)


def grid_index(pos_grid, gridtype, align_corners, hashmap_size, resolution):
    """get_grid_index (:66-84) for an [n, D] uint32 array of vertex coordinates -> entry index (before the * C)"""
    n, D = pos_grid.shape
    pg = pos_grid.astype(np.uint64)
    stride, index = 1, np.zeros(n, np.uint64)
    step = resolution if align_corners else resolution + 1
    for d in range(D):
        if stride > hashmap_size:
            break
        index = (index + pg[:, d] * np.uint64(stride)) & np.uint64(0xFFFFFFFF)
        stride = (stride * step) & 0xFFFFFFFF
    if gridtype == 0 and stride > hashmap_size:
        index = np.zeros(n, np.uint64)
        for d in range(D):
            index ^= (pg[:, d] * PRIMES[d]) & np.uint64(0xFFFFFFFF)
    return (index % np.uint64(hashmap_size)).astype(np.int64)


def _cell(inputs, scale, align_corners):
    pos = fma32(inputs, scale, f32(0.0 if align_corners else 0.5))
    pos_grid = np.floor(pos).astype(np.uint32)
    return (pos - pos_grid.astype(f32)).astype(f32), pos_grid


def _corner(frac, pos_grid, idx):
    """weight and vertex of corner idx: `w *= 1 - pos[d]` / `w *= pos[d]` for d = 0 .. D-1 (:165-178)"""
    n, D = frac.shape
    w = np.ones(n, f32)
    v = pos_grid.copy()
    for d in range(D):
        if idx & (1 << d):
            w = (w * frac[:, d]).astype(f32)
            v[:, d] += 1
        else:
            w = (w * (f32(1) - frac[:, d]).astype(f32)).astype(f32)
    return w, v


def grid_encode_forward(inputs, embeddings, offsets, S, H, gridtype=0, align_corners=False):
    """inputs [B, D] float32 in [0, 1] (others -> zeros), embeddings [T, C] float32 or float16 -> outputs [L, B, C] in the table's dtype"""
    inputs = np.asarray(inputs, f32)
    B, D = inputs.shape
    C = embeddings.shape[1]
    L = len(offsets) - 1
    half = embeddings.dtype == np.float16
    out = np.zeros((L, B, C), embeddings.dtype)
    ok = np.all((inputs >= 0) & (inputs <= 1), axis=1)
    x = inputs[ok]
    for level in range(L):
        scale, res = level_geometry(level, S, H)
        size = int(offsets[level + 1] - offsets[level])
        table = embeddings[offsets[level]:offsets[level + 1]]
        frac, pg = _cell(x, scale, align_corners)
        acc = np.zeros((x.shape[0], C), np.float16 if half else f32)
        for idx in range(1 << D):
            w, v = _corner(frac, pg, idx)
            g = table[grid_index(v, gridtype, align_corners, size, res)]
            if half:                                           # Half += (float * Half): the product is converted to Half, then a Half add
                prod = (w[:, None] * g.astype(f32)).astype(f32).astype(np.float16)
                acc = (acc.astype(f32) + prod.astype(f32)).astype(np.float16)
            else:
                acc = fma32(w[:, None], g, acc)
        out[level, ok] = acc
    return out


def grid_encode_backward(grad, inputs, offsets, S, H, n_entries, gridtype=0, align_corners=False):
    """grad [L, B, C] (float32) -> grad_embeddings [n_entries, C] float32 (the atomics' sums in point order, float32 adds)"""
    inputs = np.asarray(inputs, f32)
    B, D = inputs.shape
    L, _, C = grad.shape
    out = np.zeros((n_entries, C), f32)
    ok = np.all((inputs >= 0) & (inputs <= 1), axis=1)
    x = inputs[ok]
    for level in range(L):
        scale, res = level_geometry(level, S, H)
        size = int(offsets[level + 1] - offsets[level])
        frac, pg = _cell(x, scale, align_corners)
        g = np.asarray(grad[level], f32)[ok]
        dst = out[offsets[level]:offsets[level + 1]]
        for idx in range(1 << D):
            w, v = _corner(frac, pg, idx)
            np.add.at(dst, grid_index(v, gridtype, align_corners, size, res), (w[:, None] * g).astype(f32))
    return out


# ------------------------------------------------------------------------------------------------ march_rays_train
def _expand_bits(v):
    v = (v * 0x00010001) & 0xFF0000FF
    v = (v * 0x00000101) & 0x0F00F00F
    v = (v * 0x00000011) & 0xC30C30C3
    v = (v * 0x00000005) & 0x49249249
    return v & 0xFFFFFFFF


def _morton(x, y, z):
    return _expand_bits(x) | (_expand_bits(y) << 1) | (_expand_bits(z) << 2)


def _clampf(x, lo, hi):
    return f32(np.fmin(f32(hi), np.fmax(f32(lo), f32(x))))          # fminf(max, fmaxf(min, x)): NaN-ignoring, as the C library's


def _frexp_exp(x):
    return int(np.frexp(f32(x))[1])


def march_rays_train(rays_o, rays_d, bitfield, bound, dt_gamma, max_steps, C, H, nears, fars, noises):
    """-> (num_steps [N], xyzs list per ray [n,3], deltas list per ray [n,2]) in ray order"""
    rays_o, rays_d = np.asarray(rays_o, f32).reshape(-1, 3), np.asarray(rays_d, f32).reshape(-1, 3)
    N = rays_o.shape[0]
    bound, dt_gamma = f32(bound), f32(dt_gamma)
    SQRT3 = f32(1.7320508075688772)
    dt_min = f32(f32(2) * SQRT3) / f32(max_steps)
    dt_max = f32(f32(f32(2) * SQRT3) * f32(1 << (C - 1))) / f32(H)
    rH = f32(1) / f32(H)
    H3 = f32(H * H * H)
    counts, pts, dls = np.zeros(N, np.int64), [], []
    for n in range(N):
        o, d = rays_o[n], rays_d[n]
        with np.errstate(divide='ignore'):
            rd = (f32(1) / d).astype(f32)
        sgn = np.copysign(f32(1), d).astype(f32)
        far = f32(fars[n])
        t0 = f32(nears[n])
        t0 = fma32(_clampf(t0 * dt_gamma, dt_min, dt_max), f32(noises[n]), t0)            # t0 += clamp(...) * noise
        t, last_t = t0, t0
        xs, ds = [], []
        while t < far and len(xs) < max_steps:
            p = [_clampf(fma32(t, d[k], o[k]), -bound, bound) for k in range(3)]
            dt = _clampf(t * dt_gamma, dt_min, dt_max)
            mx = np.fmax(abs(p[0]), np.fmax(abs(p[1]), abs(p[2])))
            lvl_pos = int(np.fmin(f32(C - 1), np.fmax(f32(0), f32(_frexp_exp(mx)))))
            mxd = f32(np.float64(f32(dt * f32(H))) * 0.5)                                  # dt * H * 0.5: float product, double literal, back to float
            lvl_dt = int(np.fmin(f32(C - 1), np.fmax(f32(0), f32(_frexp_exp(mxd)))))
            level = max(lvl_pos, lvl_dt)
            mip_bound = f32(np.fmin(f32(np.ldexp(f32(1), level)), bound))
            mip_rbound = f32(1) / mip_bound
            nxyz = []
            for k in range(3):
                inner = fma32(p[k], mip_rbound, f32(1))                                    # x * mip_rbound + 1 (float)
                v = f32(0.5 * np.float64(inner) * np.float64(H))                           # double from the literal on, float at the call
                nxyz.append(int(_clampf(v, f32(0), f32(H - 1))))
            index = np.uint32(f32(f32(level) * H3) + f32(_morton(*nxyz)))                  # `level * H3 + morton`: a FLOAT sum (H3 is float)
            occ = (int(bitfield[int(index) // 8]) >> (int(index) % 8)) & 1
            if occ:
                xs.append(p)
                t = f32(t + dt)
                ds.append((dt, f32(t - last_t)))
                last_t = t
            else:
                tt3 = []
                for k in range(3):
                    a = fma32(f32(0.5), sgn[k], f32(f32(nxyz[k]) + f32(0.5)))              # nx + 0.5f + 0.5f * signf(dx)
                    a = fma32(f32(a * rH), f32(2), f32(-1))                                # * rH * 2 - 1
                    a = fma32(a, mip_bound, -p[k])                                         # * mip_bound - x
                    tt3.append(f32(a * rd[k]))
                tt = f32(t + np.fmax(f32(0), np.fmin(tt3[0], np.fmin(tt3[1], tt3[2]))))
                while True:
                    t = f32(t + _clampf(t * dt_gamma, dt_min, dt_max))
                    if not (t < tt):
                        break
        counts[n] = len(xs)
        pts.append(np.array(xs, f32).reshape(-1, 3))
        dls.append(np.array(ds, f32).reshape(-1, 2))
    return counts, pts, dls
