"""Hot-path helpers of the reference's nerf/provider_utils.py and the ray generators."""
import torch
from torch.autograd import Function

from .._lib import lib, check, ptr, stream, require_cuda


class _trunc_exp(Function):
    """provider_utils.py:16-29: forward exp(x) in float32, backward g * exp(clamp(x, -15, 15))."""

    @staticmethod
    def forward(ctx, x):
        x = x.float()
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        x = ctx.saved_tensors[0]
        return g * torch.exp(x.clamp(-15, 15))


trunc_exp = _trunc_exp.apply


def safe_normalize(x, eps=1e-20):
    """provider_utils.py:125-126"""
    return x / torch.sqrt(torch.clamp(torch.sum(x * x, -1, keepdim=True), min=eps))


def custom_meshgrid(*args):
    """provider_utils.py:118-123"""
    return torch.meshgrid(*args, indexing='ij')


def generate_rays(c2w, fx, fy, cx, cy, H, W, level=1.0, convention='nerfstudio', distortion=None):
    """Per-pixel rays for V cameras on the device (HIP kernel k_generate_rays).
    distortion = [k1, k2, k3, k4, p1, p2] selects the OPENCV_FISHEYE branch of the nerfstudio convention (provider.py:421-433).

    convention 'nerfstudio': reference nerf/provider.py:402-464 pinhole branch (x = linspace(0, W*level-1, W) + .5,
    dir = normalize(R [ (x-cx)/fx, -(y-cy)/fy, -1 ]), origin = c2w[:, 3], output [V, H, W, 3]);
    convention 'ngp': reference nerf/provider_utils.py:239-302 get_rays (all-pixels branch).
    c2w: [V, 3, 4] (or [V, 4, 4]) float tensor on the GPU.  Returns origins, directions [V, H, W, 3]."""
    c2w = c2w[:, :3, :4].contiguous().float()
    require_cuda(c2w)
    V = c2w.shape[0]
    origins = torch.empty(V, H, W, 3, dtype=torch.float32, device=c2w.device)
    directions = torch.empty_like(origins)
    if distortion is not None:
        import ctypes
        if convention != 'nerfstudio' or len(distortion) != 6:
            raise ValueError("distortion = [k1, k2, k3, k4, p1, p2] goes with the nerfstudio convention")
        k = (ctypes.c_float * 6)(*[float(v) for v in distortion])
        check(lib.cnerf_generate_rays_fisheye(ptr(c2w), V, int(H), int(W), float(fx), float(fy), float(cx), float(cy), float(level),
                                              ctypes.cast(k, ctypes.c_void_p), ptr(origins), ptr(directions), stream()), "generate_rays_fisheye")
        return origins, directions
    conv = {'nerfstudio': 0, 'ngp': 1}[convention]
    check(lib.cnerf_generate_rays(ptr(c2w), V, int(H), int(W), float(fx), float(fy), float(cx), float(cy), float(level), conv,
                                  ptr(origins), ptr(directions), stream()), "generate_rays")
    return origins, directions


def get_rays(poses, intrinsics, H, W, N=-1, error_map=None, offset=(0.5, 0.5)):
    """provider_utils.py:239-302 call surface (the all-pixels branch, which is what a full-view step uses).
    poses [B,4,4] cam2world, intrinsics (fx, fy, cx, cy) -> {'rays_o','rays_d'} [B, H*W, 3]."""
    if N > 0 or error_map is not None or tuple(offset) != (0.5, 0.5):
        raise NotImplementedError("get_rays: only the all-pixels branch is on CustomNeRF's path (the reference never calls the others)")
    fx, fy, cx, cy = intrinsics
    o, d = generate_rays(poses, fx, fy, cx, cy, H, W, 1.0, 'ngp')
    B = poses.shape[0]
    return {'rays_o': o.view(B, H * W, 3), 'rays_d': d.view(B, H * W, 3)}
