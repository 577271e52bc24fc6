"""Seeded synthetic scene of SURVEY.md §8(d) / BASELINE.md §2 (no files, no network): cameras on a circle looking at
the origin, random rgb targets with a centred disc mask, a unit-sphere occupancy grid.  Pure numpy/torch host code used
by bench.py, __graft_entry__.smoke() and the tests."""
import argparse
import math

import numpy as np
import torch


def camera_pose(view=0, n_views=8, radius=3.5, elev_deg=20.0, opencv=False):
    """[3,4] camera-to-world.  OpenGL axes (x right, y up, -z forward: what provider.py:435-438 expects) or
    OpenCV axes (x right, y down, +z forward: what get_rays expects)."""
    th = 2 * np.pi * view / n_views
    el = np.deg2rad(elev_deg)
    eye = np.array([radius * np.cos(el) * np.cos(th), radius * np.sin(el), radius * np.cos(el) * np.sin(th)])
    fwd = -eye / np.linalg.norm(eye)
    right = np.cross(fwd, np.array([0.0, 1.0, 0.0]))
    right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    if opencv:
        return np.stack([right, -up, fwd, eye], axis=1).astype(np.float32)
    return np.stack([right, up, -fwd, eye], axis=1).astype(np.float32)


def intrinsics(H, W, fovy_deg=50.0):
    f = 0.5 * H / np.tan(0.5 * np.deg2rad(fovy_deg))
    return float(f), float(f), W / 2.0, H / 2.0


def poses(n_views=8, **kw):
    return np.stack([camera_pose(v, n_views, **kw) for v in range(n_views)])


def targets(n_views, H, W, seed=0):
    """gt rgb U(0,1) [V,H*W,3] and mask = pixel within 0.4*H of the centre [V,H*W,1]."""
    g = torch.Generator().manual_seed(seed)
    rgb = torch.rand(n_views, H * W, 3, generator=g)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing='ij')
    m = (((xx + 0.5 - W / 2) ** 2 + (yy + 0.5 - H / 2) ** 2).sqrt() < 0.4 * H).float().reshape(1, H * W, 1).repeat(n_views, 1, 1)
    return rgb, m


def sphere_targets(rays_o, rays_d, radius=1.0):
    """Multi-view CONSISTENT targets (bench.py --prefit: a field fitted to them has a surface for the importance samples to cluster around):
    the analytic unit sphere at the origin, colour = 0.5 + 0.5 * surface normal where the ray hits it, black elsewhere; mask = hit.
    rays_o, rays_d [V, N, 3] (unit directions) -> rgb [V, N, 3], mask [V, N, 1] on the rays' device."""
    b = (rays_o * rays_d).sum(-1)
    c = (rays_o * rays_o).sum(-1) - radius * radius
    disc = b * b - c
    hit = (disc > 0) & (-b - disc.clamp_min(0).sqrt() > 0)
    t = -b - disc.clamp_min(0).sqrt()
    n = (rays_o + t.unsqueeze(-1) * rays_d) / radius
    rgb = torch.where(hit.unsqueeze(-1), 0.5 + 0.5 * n, torch.zeros_like(n))
    return rgb.contiguous(), hit.unsqueeze(-1).to(rgb.dtype).contiguous()


def sphere_density_grid(cascade=2, grid_size=128, bound=2.0, radius=1.0, value=100.0):
    """density_grid [cascade, H^3] in MORTON order: `value` where the cell centre is inside the sphere, else 0."""
    H = grid_size
    c = np.arange(H, dtype=np.uint32)
    X, Y, Z = np.meshgrid(c, c, c, indexing='ij')

    def expand(v):
        v = (v * np.uint32(0x00010001)) & np.uint32(0xFF0000FF)
        v = (v * np.uint32(0x00000101)) & np.uint32(0x0F00F00F)
        v = (v * np.uint32(0x00000011)) & np.uint32(0xC30C30C3)
        v = (v * np.uint32(0x00000005)) & np.uint32(0x49249249)
        return v
    morton = (expand(X) | (expand(Y) << np.uint32(1)) | (expand(Z) << np.uint32(2))).reshape(-1).astype(np.int64)
    grid = np.zeros((cascade, H ** 3), np.float32)
    for cas in range(cascade):
        b = min(2 ** cas, bound)
        ctr = ((np.stack([X, Y, Z], -1).reshape(-1, 3).astype(np.float32) + 0.5) / H * 2 - 1) * b
        inside = (ctr ** 2).sum(-1) < radius ** 2
        grid[cas, morton] = np.where(inside, value, 0.0)
    return grid


def make_opt(**kw):
    """The reference's argparse defaults that reach the hot path (main.py:11-146), as a Namespace."""
    o = argparse.Namespace(
        bound=2.0, min_near=0.01, num_steps=64, upsample_steps=64, max_steps=1024, cuda_ray=False, fp16=False,
        density_thresh=10, update_extra_interval=100, train_conf=0.01, train_rgb=1.0, soft_mask=True, conf_thr=0.5,
        detach_bg=False, detach_mask_from_field=False, mask_no_dir=False, mask_no_dir_nodetach=False, keyword2=None,
        batch_rays=0, lambda_sd=0.01, cfg=100, max_ratio=0.98, stage_time=False, global_ratio=0.5, local_t_ratio=0.5,
        keep_bg=0, lr=5e-4, iters=3000, backbone='grid', bg_color=None, dt_gamma=0,
        # grid geometry of the synthetic benchmark scene; the reference's own field uses
        # grid_type='tiledgrid', log2_hashmap_size=21, desired_resolution=8192 (network_grid.py:89-96)
        grid_type='hashgrid', log2_hashmap_size=19, desired_resolution=2048, num_levels=16, level_dim=2, base_resolution=16,
        n_hidden_geo=2)
    o.__dict__.update(kw)
    return o
