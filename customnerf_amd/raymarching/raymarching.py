"""Python surface of the ray-marching ops — same names, arguments and return values as the reference's
`raymarching/raymarching.py` (cited per function), backed by libcustomnerf_hip.so through ctypes.

Differences that are deliberate:
  * march_rays_train lays samples out in ray order (deterministic) instead of atomics order, never zero-fills the
    N*max_steps worst case of xyzs / dirs / deltas, and synchronises with the host only to size its outputs (the reference
    does the same `.item()` at raymarching.py:225).  The exact-size path keeps an UNINITIALISED per-device scratch of
    N*max_steps*8 bytes for the occupied-probe list (capped at _HITS_MAX_BYTES, released when requests shrink; larger calls
    re-march instead); in the fixed-budget path a call that overflows its budget drops a run of rays that starts at a
    jitter-dependent ray (see cnerf_march_rays_train), not always the last image rows;
  * composite ops accept rgbs with 3 or 4 channels per sample (the 4th, confidence, is ignored) so the renderer can
    pass the field output without a slice copy;
  * there is no CPU path: tensors are moved to the GPU exactly like the reference does (raymarching.py:35-36).
"""
import torch
from torch.autograd import Function

from .._lib import lib, check, ptr, stream, scratch_key, scratch_reallocated

__all__ = ["near_far_from_aabb", "sph_from_ray", "morton3D", "morton3D_invert", "packbits", "march_rays_train",
           "composite_rays_train", "composite_rays_train_sdf", "march_rays", "composite_rays", "compact_rays_alive"]


def _cuda_f32(t):
    if not t.is_cuda:
        t = t.cuda()
    return t.contiguous().float()


# ---------------------------------------------------------------------------------------------- utils
def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
    """raymarching.py:20-50.  rays_o/d [N,3] float, aabb [6] -> nears, fars [N]."""
    rays_o = _cuda_f32(rays_o).view(-1, 3)
    rays_d = _cuda_f32(rays_d).view(-1, 3)
    aabb = _cuda_f32(aabb)
    N = rays_o.shape[0]
    nears = torch.empty(N, dtype=torch.float32, device=rays_o.device)
    fars = torch.empty(N, dtype=torch.float32, device=rays_o.device)
    check(lib.cnerf_near_far_from_aabb(ptr(rays_o), ptr(rays_d), ptr(aabb), N, float(min_near), ptr(nears), ptr(fars), stream()),
          "near_far_from_aabb")
    return nears, fars


def sph_from_ray(rays_o, rays_d, radius):
    """raymarching.py:53-81 -> coords [N,2] in [-1,1]."""
    rays_o = _cuda_f32(rays_o).view(-1, 3)
    rays_d = _cuda_f32(rays_d).view(-1, 3)
    N = rays_o.shape[0]
    coords = torch.empty(N, 2, dtype=torch.float32, device=rays_o.device)
    check(lib.cnerf_sph_from_ray(ptr(rays_o), ptr(rays_d), float(radius), N, ptr(coords), stream()), "sph_from_ray")
    return coords


def morton3D(coords):
    """raymarching.py:84-105.  coords [N,3] int -> indices [N] int32."""
    if not coords.is_cuda:
        coords = coords.cuda()
    coords = coords.int().contiguous()
    N = coords.shape[0]
    indices = torch.empty(N, dtype=torch.int32, device=coords.device)
    check(lib.cnerf_morton3D(ptr(coords), N, ptr(indices), stream()), "morton3D")
    return indices


def morton3D_invert(indices):
    """raymarching.py:107-127.  indices [N] int -> coords [N,3] int32."""
    if not indices.is_cuda:
        indices = indices.cuda()
    indices = indices.int().contiguous()
    N = indices.shape[0]
    coords = torch.empty(N, 3, dtype=torch.int32, device=indices.device)
    check(lib.cnerf_morton3D_invert(ptr(indices), N, ptr(coords), stream()), "morton3D_invert")
    return coords


def packbits(grid, thresh, bitfield=None):
    """raymarching.py:130-156.  grid [C, H^3] float -> bitfield uint8 [C*H^3/8]."""
    grid = _cuda_f32(grid)
    C, H3 = grid.shape[0], grid.shape[1]
    N = C * H3 // 8
    if bitfield is None:
        bitfield = torch.empty(N, dtype=torch.uint8, device=grid.device)
    check(lib.cnerf_packbits(ptr(grid), N, float(thresh), ptr(bitfield), stream()), "packbits")
    return bitfield


# ---------------------------------------------------------------------------------------------- training
_HITS = {}
_HITS_MAX_BYTES = 512 << 20          # above this the write pass re-marches (cnerf_march_rays_train_write) instead of keeping a probe list


def _hits_scratch(N, max_steps, device):
    """Per-(device, stream) scratch (scratch_key: concurrent calls on two streams must not share it, nor have it freed under them) for the
    occupied-probe list of march_rays_train: N * max_steps (t, dt) pairs, uninitialised.  Grown on demand,
    released when a request needs less than a quarter of what is held (one full-image call must not pin its worst case for the life of
    the process), and never larger than _HITS_MAX_BYTES: returns None beyond that and the caller takes the re-marching writer."""
    need = N * max_steps * 2
    key = scratch_key(device)
    if need * 4 > _HITS_MAX_BYTES:
        if _HITS.pop(key, None) is not None:
            scratch_reallocated()
        return None
    buf = _HITS.get(key)
    if buf is None or buf.numel() < need or buf.numel() > 4 * need:
        _HITS.pop(key, None)
        buf = None                                                  # drop the old block before asking for the new one
        buf = _HITS[key] = torch.empty(need, dtype=torch.float32, device=device)
        scratch_reallocated()
    return buf


def march_rays_train(rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter=None, mean_count=-1,
                     perturb=False, align=-1, force_all_rays=False, dt_gamma=0, max_steps=1024, noises=None):
    """raymarching.py:162-236 -> (xyzs [M,3], dirs [M,3], deltas [M,2], rays [N,3] int32).

    `noises` (optional [N] tensor) replaces the wrapper's torch.rand draw so tests can replay it."""
    rays_o = _cuda_f32(rays_o).view(-1, 3)
    rays_d = _cuda_f32(rays_d).view(-1, 3)
    if not density_bitfield.is_cuda:
        density_bitfield = density_bitfield.cuda()
    density_bitfield = density_bitfield.contiguous()
    nears, fars = _cuda_f32(nears), _cuda_f32(fars)
    dev = rays_o.device
    N = rays_o.shape[0]
    rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
    if step_counter is None:
        step_counter = torch.zeros(2, dtype=torch.int32, device=dev)
    if noises is None:
        noises = torch.rand(N, dtype=torch.float32, device=dev) if perturb else torch.zeros(N, dtype=torch.float32, device=dev)
    else:
        noises = _cuda_f32(noises)
    args = (ptr(rays_o), ptr(rays_d), ptr(density_bitfield), float(bound), float(dt_gamma), int(max_steps), N, int(C), int(H))

    if not force_all_rays and mean_count > 0:
        # fixed budget, no host sync (raymarching.py:201-204): rays whose segment overflows M are dropped
        if align > 0:
            mean_count += align - mean_count % align
        M = int(mean_count)
        xyzs = torch.zeros(M, 3, dtype=torch.float32, device=dev)
        dirs = torch.zeros(M, 3, dtype=torch.float32, device=dev)
        deltas = torch.zeros(M, 2, dtype=torch.float32, device=dev)
        check(lib.cnerf_march_rays_train(*args, M, ptr(nears), ptr(fars), ptr(xyzs), ptr(dirs), ptr(deltas), ptr(rays),
                                         ptr(step_counter), ptr(noises), stream()), "march_rays_train")
        return xyzs, dirs, deltas, rays

    # count -> size -> write: allocates exactly the samples that exist (+ alignment padding, zero-filled as in the
    # reference where the tail of the zero-initialised buffers is returned, raymarching.py:206-208,226-230)
    base = step_counter[0:1].clone()
    # the counting march records (t, dt) of every occupied probe: the write pass then needs no second march (hits: caller-owned scratch,
    # N x max_steps x 8 bytes, uninitialised; only the first num_steps entries of a ray's row are written and read)
    hits = _hits_scratch(N, int(max_steps), dev)
    if hits is not None:
        check(lib.cnerf_march_rays_train_count_hits(*args, ptr(nears), ptr(fars), ptr(rays), ptr(step_counter), ptr(noises), ptr(hits), stream()),
              "march_rays_train_count_hits")
    else:
        check(lib.cnerf_march_rays_train_count(*args, ptr(nears), ptr(fars), ptr(rays), ptr(step_counter), ptr(noises), stream()),
              "march_rays_train_count")
    b0, m = torch.cat([base, step_counter[0:1]]).tolist()          # ONE D2H sync (same point as raymarching.py:225)
    if b0 != 0:
        raise ValueError("march_rays_train: step_counter must be zeroed by the caller (renderer.py:619-620)")
    m_alloc = m + (align - m % align) if align > 0 else m
    m_alloc = min(m_alloc, N * int(max_steps))          # the reference slices a buffer of N * max_steps rows (raymarching.py:196,226-230)
    xyzs = torch.empty(m_alloc, 3, dtype=torch.float32, device=dev)
    dirs = torch.empty(m_alloc, 3, dtype=torch.float32, device=dev)
    deltas = torch.empty(m_alloc, 2, dtype=torch.float32, device=dev)
    if m_alloc > m:
        xyzs[m:].zero_(); dirs[m:].zero_(); deltas[m:].zero_()
    if hits is not None:
        check(lib.cnerf_march_rays_train_write_hits(ptr(rays_o), ptr(rays_d), float(bound), float(dt_gamma), int(max_steps), N, int(C), int(H), m_alloc,
                                                    ptr(nears), ptr(noises), ptr(hits), ptr(rays), ptr(xyzs), ptr(dirs), ptr(deltas), stream()),
              "march_rays_train_write_hits")
    else:
        check(lib.cnerf_march_rays_train_write(*args, m_alloc, ptr(nears), ptr(fars), ptr(xyzs), ptr(dirs), ptr(deltas), ptr(rays), ptr(noises), stream()),
              "march_rays_train_write")
    return xyzs, dirs, deltas, rays


class _composite_rays_train(Function):
    """raymarching.py:239-289."""

    @staticmethod
    def forward(ctx, sigmas, rgbs, deltas, rays, T_thresh=1e-4):
        sigmas = sigmas.contiguous().float()
        rgbs = rgbs.contiguous().float()
        deltas = deltas.contiguous().float()
        rays = rays.contiguous()
        M, N = sigmas.shape[0], rays.shape[0]
        rs = rgbs.shape[-1]
        dev = sigmas.device
        weights_sum = torch.empty(N, dtype=torch.float32, device=dev)
        depth = torch.empty(N, dtype=torch.float32, device=dev)
        image = torch.empty(N, 3, dtype=torch.float32, device=dev)
        check(lib.cnerf_composite_rays_train_forward(ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(rays), M, N, float(T_thresh),
                                                     ptr(weights_sum), ptr(depth), ptr(image), rs, stream()), "composite_rays_train_forward")
        ctx.save_for_backward(sigmas, rgbs, deltas, rays, weights_sum, depth, image)
        ctx.dims = [M, N, T_thresh, rs]
        return weights_sum, depth, image

    @staticmethod
    def backward(ctx, grad_weights_sum, grad_depth, grad_image):
        # grad_depth is not propagated (raymarching.py:276)
        grad_weights_sum = grad_weights_sum.contiguous().float()
        grad_image = grad_image.contiguous().float()
        sigmas, rgbs, deltas, rays, weights_sum, depth, image = ctx.saved_tensors
        M, N, T_thresh, rs = ctx.dims
        grad_sigmas = torch.zeros_like(sigmas)
        grad_rgbs = torch.zeros_like(rgbs)
        check(lib.cnerf_composite_rays_train_backward(ptr(grad_weights_sum), ptr(grad_image), ptr(sigmas), ptr(rgbs), ptr(deltas),
                                                      ptr(rays), ptr(weights_sum), ptr(image), M, N, float(T_thresh),
                                                      ptr(grad_sigmas), ptr(grad_rgbs), rs, stream()), "composite_rays_train_backward")
        return grad_sigmas, grad_rgbs, None, None, None


def composite_rays_train(sigmas, rgbs, deltas, rays, T_thresh=1e-4):
    return _composite_rays_train.apply(sigmas, rgbs, deltas, rays, T_thresh)


# the reference's `_sdf` variant is a byte-identical duplicate (raymarching.py:295-348, raymarching.cu:579-657, 776-857)
composite_rays_train_sdf = composite_rays_train


# ---------------------------------------------------------------------------------------------- inference
def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far, align=-1,
               perturb=False, dt_gamma=0, max_steps=1024, noises=None):
    """raymarching.py:355-405 -> (xyzs, dirs [n_alive*n_step (+pad), 3], deltas [.., 2])."""
    rays_o = _cuda_f32(rays_o).view(-1, 3)
    rays_d = _cuda_f32(rays_d).view(-1, 3)
    dev = rays_o.device
    M = n_alive * n_step
    if align > 0:
        M += align - (M % align)
    xyzs = torch.zeros(M, 3, dtype=torch.float32, device=dev)
    dirs = torch.zeros(M, 3, dtype=torch.float32, device=dev)
    deltas = torch.zeros(M, 2, dtype=torch.float32, device=dev)
    if noises is None:
        noises = torch.rand(n_alive, dtype=torch.float32, device=dev) if perturb else torch.zeros(n_alive, dtype=torch.float32, device=dev)
    check(lib.cnerf_march_rays(int(n_alive), int(n_step), ptr(rays_alive), ptr(rays_t), ptr(rays_o), ptr(rays_d), float(bound),
                               float(dt_gamma), int(max_steps), int(C), int(H), ptr(density_bitfield), ptr(near), ptr(far),
                               ptr(xyzs), ptr(dirs), ptr(deltas), ptr(noises), stream()), "march_rays")
    return xyzs, dirs, deltas


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh=1e-2):
    """raymarching.py:408-427.  In place on rays_alive / rays_t / weights_sum / depth / image."""
    sigmas = sigmas.contiguous().float()
    rgbs = rgbs.contiguous().float()
    check(lib.cnerf_composite_rays(int(n_alive), int(n_step), float(T_thresh), ptr(rays_alive), ptr(rays_t), ptr(sigmas), ptr(rgbs),
                                   ptr(deltas), ptr(weights_sum), ptr(depth), ptr(image), rgbs.shape[-1], stream()), "composite_rays")
    return tuple()


def compact_rays_alive(rays_alive, n_alive=None, out=None, count=None):
    """Device-side, order-preserving `rays_alive[rays_alive >= 0]` (renderer.py:685).  Returns (out, count_tensor):
    the first count entries of `out` are the surviving ray ids.  No host sync."""
    n = rays_alive.shape[0] if n_alive is None else int(n_alive)
    if out is None:
        out = torch.empty_like(rays_alive)
    if count is None:
        count = torch.zeros(1, dtype=torch.int32, device=rays_alive.device)
    check(lib.cnerf_compact_rays_alive(ptr(rays_alive), n, ptr(out), ptr(count), stream()), "compact_rays_alive")
    return out, count
