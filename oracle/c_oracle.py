"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product path).

numpy/ctypes front-end of ``liboracle.so`` (plain-C restatement of the reference's
``raymarching/src/raymarching.cu`` and ``gridencoder/src/gridencoder.cu``).
Array allocation rules follow the reference's Python wrappers
(``raymarching/raymarching.py``, ``gridencoder/grid.py``), cited per function.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


u32, f32 = C.c_uint32, C.c_float


# ---------------------------------------------------------------- raymarching
def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
    """raymarching.py:23-50 -> raymarching.cu:91-145"""
    rays_o, rays_d, aabb = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3), _f32(aabb)
    N = rays_o.shape[0]
    nears, fars = np.empty(N, np.float32), np.empty(N, np.float32)
    lib().orc_near_far_from_aabb(_p(rays_o), _p(rays_d), _p(aabb), u32(N), f32(min_near), _p(nears), _p(fars))
    return nears, fars


def sph_from_ray(rays_o, rays_d, radius):
    """raymarching.py:56-81 -> raymarching.cu:162-198"""
    rays_o, rays_d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    N = rays_o.shape[0]
    coords = np.empty((N, 2), np.float32)
    lib().orc_sph_from_ray(_p(rays_o), _p(rays_d), f32(radius), u32(N), _p(coords))
    return coords


def morton3D(coords):
    """raymarching.py:86-105 -> raymarching.cu:214-226"""
    coords = np.ascontiguousarray(coords, dtype=np.int32)
    N = coords.shape[0]
    out = np.empty(N, np.int32)
    lib().orc_morton3D(_p(coords), u32(N), _p(out))
    return out


def morton3D_invert(indices):
    """raymarching.py:109-127 -> raymarching.cu:237-254"""
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    N = indices.shape[0]
    out = np.empty((N, 3), np.int32)
    lib().orc_morton3D_invert(_p(indices), u32(N), _p(out))
    return out


def packbits(grid, thresh, bitfield=None):
    """raymarching.py:133-156 -> raymarching.cu:267-289"""
    grid = _f32(grid)
    N = grid.shape[0] * grid.shape[1] // 8
    if bitfield is None:
        bitfield = np.empty(N, np.uint8)
    lib().orc_packbits(_p(grid), u32(N), f32(thresh), _p(bitfield))
    return bitfield


def march_rays_train(rays_o, rays_d, bound, density_bitfield, Cc, H, nears, fars, step_counter=None, mean_count=-1,
                     noises=None, align=-1, force_all_rays=False, dt_gamma=0, max_steps=1024):
    """raymarching.py:165-236 -> raymarching.cu:311-480.  ``noises`` replaces the wrapper's
    torch.rand (perturb) so both sides can be fed the same numbers; None = zeros."""
    rays_o, rays_d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    N = rays_o.shape[0]
    M = N * max_steps
    if not force_all_rays and mean_count > 0:
        if align > 0:
            mean_count += align - mean_count % align
        M = mean_count
    xyzs, dirs = np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32)
    deltas = np.zeros((M, 2), np.float32)
    rays = np.empty((N, 3), np.int32)
    if step_counter is None:
        step_counter = np.zeros(2, np.int32)
    noises = np.zeros(N, np.float32) if noises is None else _f32(noises)
    bf = np.ascontiguousarray(density_bitfield, dtype=np.uint8)
    lib().orc_march_rays_train(_p(rays_o), _p(rays_d), _p(bf), f32(bound), f32(dt_gamma), u32(max_steps), u32(N), u32(Cc),
                               u32(H), u32(M), _p(_f32(nears)), _p(_f32(fars)), _p(xyzs), _p(dirs), _p(deltas), _p(rays),
                               _p(step_counter), _p(noises))
    if force_all_rays or mean_count <= 0:
        m = int(step_counter[0])
        if align > 0:
            m += align - m % align
        xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
    return xyzs, dirs, deltas, rays


def composite_rays_train_forward(sigmas, rgbs, deltas, rays, T_thresh=1e-4):
    """raymarching.py:242-270 -> raymarching.cu:500-577"""
    sigmas, rgbs, deltas = _f32(sigmas), _f32(rgbs), _f32(deltas)
    rays = np.ascontiguousarray(rays, np.int32)
    M, N = sigmas.shape[0], rays.shape[0]
    ws, depth, image = np.empty(N, np.float32), np.empty(N, np.float32), np.empty((N, 3), np.float32)
    lib().orc_composite_rays_train_forward(_p(sigmas), _p(rgbs), _p(deltas), _p(rays), u32(M), u32(N), f32(T_thresh),
                                           _p(ws), _p(depth), _p(image))
    return ws, depth, image


def composite_rays_train_backward(grad_ws, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, T_thresh=1e-4):
    """raymarching.py:272-289 -> raymarching.cu:691-772"""
    sigmas, rgbs, deltas = _f32(sigmas), _f32(rgbs), _f32(deltas)
    rays = np.ascontiguousarray(rays, np.int32)
    M, N = sigmas.shape[0], rays.shape[0]
    gs, gc = np.zeros_like(sigmas), np.zeros_like(rgbs)
    lib().orc_composite_rays_train_backward(_p(_f32(grad_ws)), _p(_f32(grad_image)), _p(sigmas), _p(rgbs), _p(deltas),
                                            _p(rays), _p(_f32(weights_sum)), _p(_f32(image)), u32(M), u32(N),
                                            f32(T_thresh), _p(gs), _p(gc))
    return gs, gc


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, Cc, H, near, far,
               align=-1, noises=None, dt_gamma=0, max_steps=1024):
    """raymarching.py:358-405 -> raymarching.cu:884-989"""
    rays_o, rays_d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    M = n_alive * n_step
    if align > 0:
        M += align - (M % align)
    xyzs, dirs = np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32)
    deltas = np.zeros((M, 2), np.float32)
    noises = np.zeros(n_alive, np.float32) if noises is None else _f32(noises)
    bf = np.ascontiguousarray(density_bitfield, dtype=np.uint8)
    lib().orc_march_rays(u32(n_alive), u32(n_step), _p(np.ascontiguousarray(rays_alive, np.int32)), _p(_f32(rays_t)),
                         _p(rays_o), _p(rays_d), f32(bound), f32(dt_gamma), u32(max_steps), u32(Cc), u32(H), _p(bf),
                         _p(_f32(near)), _p(_f32(far)), _p(xyzs), _p(dirs), _p(deltas), _p(noises))
    return xyzs, dirs, deltas


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh=1e-2):
    """raymarching.py:411-427 -> raymarching.cu:1002-1089.  In place on rays_alive/rays_t/weights_sum/depth/image
    (all must be C-contiguous numpy arrays of the right dtype)."""
    for a, dt in ((rays_alive, np.int32), (rays_t, np.float32), (weights_sum, np.float32), (depth, np.float32), (image, np.float32)):
        assert a.dtype == dt and a.flags.c_contiguous
    lib().orc_composite_rays(u32(n_alive), u32(n_step), f32(T_thresh), _p(rays_alive), _p(rays_t), _p(_f32(sigmas)),
                             _p(_f32(rgbs)), _p(_f32(deltas)), _p(weights_sum), _p(depth), _p(image))


# ---------------------------------------------------------------- gridencoder
def f2h(a):
    a = _f32(a)
    out = np.empty(a.shape, np.uint16)
    lib().orc_f2h(_p(a), _p(out), u32(a.size))
    return out


def h2f(a):
    a = np.ascontiguousarray(a, np.uint16)
    out = np.empty(a.shape, np.float32)
    lib().orc_h2f(_p(a), _p(out), u32(a.size))
    return out


def level_geometry(level, S, H):
    sc, res = f32(), u32()
    lib().orc_level_geometry(u32(level), f32(S), u32(H), C.byref(sc), C.byref(res))
    return sc.value, res.value


def grid_encode_forward(inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False,
                        gridtype=0, align_corners=False, interpolation=0, max_level=None, half=False):
    """grid.py:27-69 -> gridencoder.cu:87-244.  Returns (outputs [B, L*C], dy_dx or None) as float32 arrays
    (half=True: embeddings are rounded to binary16 first and all arithmetic follows the at::Half path; the
    float32 arrays returned then hold exactly-representable half values)."""
    inputs = _f32(inputs)
    B, D = inputs.shape
    offsets = np.ascontiguousarray(offsets, np.int32)
    L = offsets.shape[0] - 1
    Cc = embeddings.shape[1]
    S = float(np.log2(per_level_scale))
    max_level = L if max_level is None else min(max_level, L)
    emb = f2h(embeddings) if half else _f32(embeddings)
    odt = np.uint16 if half else np.float32
    outputs = np.zeros((L, B, Cc), odt)
    dy_dx = np.zeros((B, L * D * Cc), odt) if calc_grad_inputs else None
    lib().orc_grid_encode_forward(_p(inputs), _p(emb), _p(offsets), _p(outputs), u32(B), u32(D), u32(Cc), u32(L), u32(max_level),
                                  f32(S), u32(base_resolution), _p(dy_dx), u32(gridtype), C.c_int(int(align_corners)),
                                  u32(interpolation), C.c_int(int(half)))
    if half:
        outputs = h2f(outputs)
        dy_dx = None if dy_dx is None else h2f(dy_dx)
    return outputs.transpose(1, 0, 2).reshape(B, L * Cc).copy(), dy_dx


def grid_encode_backward(grad, inputs, embeddings_shape, offsets, per_level_scale, base_resolution, dy_dx=None,
                         gridtype=0, align_corners=False, interpolation=0, max_level=None):
    """grid.py:74-95 -> gridencoder.cu:247-368.  grad: [B, L*C] float32.  Returns (grad_embeddings f32, grad_inputs|None)."""
    inputs = _f32(inputs)
    B, D = inputs.shape
    offsets = np.ascontiguousarray(offsets, np.int32)
    L = offsets.shape[0] - 1
    Cc = embeddings_shape[1]
    S = float(np.log2(per_level_scale))
    max_level = L if max_level is None else min(max_level, L)
    g = np.ascontiguousarray(_f32(grad).reshape(B, L, Cc).transpose(1, 0, 2))
    ge = np.zeros(embeddings_shape, np.float32)
    gi = np.zeros((B, D), np.float32) if dy_dx is not None else None
    lib().orc_grid_encode_backward(_p(g), _p(inputs), _p(offsets), _p(ge), u32(B), u32(D), u32(Cc), u32(L), u32(max_level),
                                   f32(S), u32(base_resolution), _p(None if dy_dx is None else _f32(dy_dx)), _p(gi),
                                   u32(gridtype), C.c_int(int(align_corners)), u32(interpolation), C.c_int(0))
    return ge, gi


def grad_total_variation(inputs, embeddings, grad, offsets, weight, per_level_scale, base_resolution, gridtype=0,
                         align_corners=False):
    """grid.py:171-192 -> gridencoder.cu:505-609.  In place on ``grad`` (float32, C-contiguous)."""
    inputs = _f32(inputs)
    B, D = inputs.shape
    offsets = np.ascontiguousarray(offsets, np.int32)
    L = offsets.shape[0] - 1
    Cc = embeddings.shape[1]
    assert grad.dtype == np.float32 and grad.flags.c_contiguous
    lib().orc_grad_total_variation(_p(inputs), _p(_f32(embeddings)), _p(grad), _p(offsets), f32(weight), u32(B), u32(D),
                                   u32(Cc), u32(L), f32(float(np.log2(per_level_scale))), u32(base_resolution),
                                   u32(gridtype), C.c_int(int(align_corners)))
