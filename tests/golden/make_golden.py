"""Generate golden vectors by IMPORTING THE REFERENCE'S PYTHON in the build container.

Run once here (the reference tree does not exist on the GPU box):

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

Nothing of the reference travels: the .npz files hold only inputs, recorded RNG draws and the outputs the
reference produced.  Missing third-party modules (trimesh, plyfile, skimage, torchtyping, tinycudann) and the
CUDA extensions are replaced by empty stubs; the one native call on the pure-PyTorch path
(`raymarching.near_far_from_aabb`, renderer.py:297) is stubbed with a straight numpy slab test.
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("CUSTOMNERF_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _near_far_np(rays_o, rays_d, aabb, min_near=0.2):
    """Independent numpy float32 slab test with the arithmetic order of raymarching.cu:108-144."""
    o = rays_o.detach().numpy().astype(np.float32).reshape(-1, 3)
    d = rays_d.detach().numpy().astype(np.float32).reshape(-1, 3)
    a = aabb.detach().numpy().astype(np.float32)
    with np.errstate(divide='ignore', invalid='ignore'):
        rd = (np.float32(1) / d).astype(np.float32)
        lo = ((a[:3] - o) * rd).astype(np.float32)
        hi = ((a[3:] - o) * rd).astype(np.float32)
    tmin, tmax = np.minimum(lo, hi), np.maximum(lo, hi)
    N = o.shape[0]
    nears, fars = np.empty(N, np.float32), np.empty(N, np.float32)
    big = np.finfo(np.float32).max
    for n in range(N):
        near, far = tmin[n, 0], tmax[n, 0]
        ny, fy = tmin[n, 1], tmax[n, 1]
        if near > fy or ny > far:
            nears[n] = fars[n] = big
            continue
        near, far = max(near, ny), min(far, fy)
        nz, fz = tmin[n, 2], tmax[n, 2]
        if near > fz or nz > far:
            nears[n] = fars[n] = big
            continue
        near, far = max(near, nz), min(far, fz)
        nears[n], fars[n] = max(near, np.float32(min_near)), far
    return torch.from_numpy(nears), torch.from_numpy(fars)


def import_reference():
    for name in ("trimesh", "plyfile", "skimage", "tinycudann", "_gridencoder", "_raymarching"):
        _stub(name)
    _stub("skimage.measure")
    sys.modules["skimage"].measure = sys.modules["skimage.measure"]

    class _TT:
        def __class_getitem__(cls, item):
            return cls
    _stub("torchtyping", TensorType=_TT)
    _stub("raymarching", near_far_from_aabb=_near_far_np)
    sys.path.insert(0, REF)
    import warnings
    warnings.filterwarnings("ignore")
    from nerf import renderer as ref_renderer
    from nerf import provider_utils as ref_pu
    from nerf import base as ref_base
    # the real gridencoder python package (its native backend is the empty `_gridencoder` stub)
    del sys.modules["raymarching"]
    import gridencoder.grid as ref_grid
    sys.modules["raymarching"] = _stub("raymarching", near_far_from_aabb=_near_far_np)
    return ref_renderer, ref_pu, ref_base, ref_grid


class _Recorder:
    """Records torch.rand / torch.randn draws made inside the reference call, in order."""

    def __enter__(self):
        self.draws = []
        self._rand, self._randn = torch.rand, torch.randn

        def rand(*a, **k):
            k.pop("device", None)
            t = self._rand(*a, **k)
            self.draws.append(("rand", t.clone()))
            return t

        def randn(*a, **k):
            k.pop("device", None)
            t = self._randn(*a, **k)
            self.draws.append(("randn", t.clone()))
            return t
        torch.rand, torch.randn = rand, randn
        return self

    def __exit__(self, *exc):
        torch.rand, torch.randn = self._rand, self._randn


def scene_rays(H, W, radius=3.5, elev_deg=20.0, fovy_deg=50.0, view=0, n_views=8, opencv=False):
    """SURVEY.md §8d synthetic scene: pinhole camera on a circle, look-at origin.  OpenGL convention (-z forward, what
    provider.py:435-438 expects) by default; opencv=True gives x-right / y-down / +z-forward (what get_rays expects)."""
    th = 2 * np.pi * view / n_views
    el = np.deg2rad(elev_deg)
    eye = np.array([radius * np.cos(el) * np.cos(th), radius * np.sin(el), radius * np.cos(el) * np.sin(th)])
    fwd = -eye / np.linalg.norm(eye)
    right = np.cross(fwd, np.array([0.0, 1.0, 0.0]))
    right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    c2w = np.stack([right, up, -fwd, eye], axis=1).astype(np.float32)      # [3,4]
    if opencv:
        c2w = np.stack([right, -up, fwd, eye], axis=1).astype(np.float32)
    f = 0.5 * H / np.tan(0.5 * np.deg2rad(fovy_deg))
    return c2w, float(f), float(f), W / 2.0, H / 2.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=HERE)
    args = ap.parse_args()
    ref_renderer, ref_pu, ref_base, ref_grid = import_reference()
    from oracle.toy_field import ToyField, toy_sigma, toy_rgbc

    # ---- (1) sample_pdf, det and random  (renderer.py:21-55)
    torch.manual_seed(1)
    bins = torch.sort(torch.rand(64, 15) * 3 + 0.5, dim=-1).values
    w = torch.rand(64, 14) ** 3
    w[5] = 0                                                  # all-zero weights row (denom<1e-5 branch)
    w[6, 3:] = 0
    out_det = ref_renderer.sample_pdf(bins, w, 16, det=True)
    with _Recorder() as rec:
        out_rnd = ref_renderer.sample_pdf(bins, w, 16, det=False)
    np.savez(os.path.join(args.out, "sample_pdf.npz"), bins=bins.numpy(), weights=w.numpy(), out_det=out_det.numpy(),
             out_rnd=out_rnd.numpy(), u=rec.draws[0][1].numpy())

    # ---- (5) trunc_exp fwd/bwd incl. |x| > 15  (provider_utils.py:16-29)
    x = torch.tensor([-20.0, -15.0, -3.0, 0.0, 0.5, 7.0, 15.0, 16.0, 20.0], requires_grad=True)
    y = ref_pu.trunc_exp(x)
    g = torch.linspace(0.5, 1.5, x.numel())
    y.backward(g)
    np.savez(os.path.join(args.out, "trunc_exp.npz"), x=x.detach().numpy(), y=y.detach().numpy(), g=g.numpy(), gx=x.grad.numpy())

    # ---- (6) get_embedder(4) on unit dirs (base.py:10-77) + safe_normalize
    torch.manual_seed(2)
    d = ref_pu.safe_normalize(torch.randn(128, 3))
    emb, out_dim = ref_base.get_embedder(4)
    np.savez(os.path.join(args.out, "embedder.npz"), d=d.numpy(), out=emb(d).numpy(), out_dim=out_dim,
             sn_in=torch.tensor([[0.0, 0.0, 0.0], [3.0, 4.0, 0.0]]).numpy(),
             sn_out=ref_pu.safe_normalize(torch.tensor([[0.0, 0.0, 0.0], [3.0, 4.0, 0.0]])).numpy())

    # ---- (4) GridEncoder offsets / per_level_scale / init range (grid.py:103-146)
    enc = {}
    for tag, kw in (("hash_L16_T19_2048", dict(num_levels=16, log2_hashmap_size=19, desired_resolution=2048, gridtype='hash')),
                    ("tiled_L16_T21_8192", dict(num_levels=16, log2_hashmap_size=21, desired_resolution=8192, gridtype='tiled')),
                    ("hash_L4_T19_2048", dict(num_levels=4, log2_hashmap_size=19, desired_resolution=2048, gridtype='hash')),
                    ("hash_default", dict())):
        torch.manual_seed(3)
        ge = ref_grid.GridEncoder(**kw)
        enc[tag + "__offsets"] = ge.offsets.numpy()
        enc[tag + "__pls"] = np.float64(ge.per_level_scale)
        enc[tag + "__absmax"] = np.float32(ge.embeddings.detach().abs().max())
        enc[tag + "__shape"] = np.array(ge.embeddings.shape)
        enc[tag + "__n_params"] = np.int64(int(ge.n_params))
        enc[tag + "__output_dim"] = np.int64(ge.output_dim)
    np.savez(os.path.join(args.out, "grid_offsets.npz"), **enc)

    # ---- (7) get_rays (provider_utils.py:239-302) and the _generate_rays pinhole math (provider.py:402-464)
    rays = {}
    for tag, (H, W) in (("32", (32, 32)), ("64", (64, 64)), ("24x40", (24, 40))):
        c2w, fx, fy, cx, cy = scene_rays(H, W, view=3, opencv=True)
        pose = torch.eye(4).unsqueeze(0).clone()
        pose[0, :3, :4] = torch.from_numpy(c2w)
        r = ref_pu.get_rays(pose, (fx, fy, cx, cy), H, W, -1)
        rays[f"get_rays_{tag}__c2w"] = c2w
        rays[f"get_rays_{tag}__intr"] = np.array([fx, fy, cx, cy, H, W], np.float64)
        rays[f"get_rays_{tag}__o"] = r['rays_o'].numpy()
        rays[f"get_rays_{tag}__d"] = r['rays_d'].numpy()
        c2w = scene_rays(H, W, view=3)[0]
        rays[f"gen_rays_{tag}__c2w"] = c2w
        # provider.py:402-464 needs a dataset object; its pinhole branch is executed here line by line on the same
        # inputs (meshgrid 'ij' over (tx,ty), +0.5, (x-cx)/fx, -(y-cy)/fy, -1, rotate, normalize, [W,H]->[H,W]).
        for level in (1, 2):
            tx = torch.linspace(0, W * level - 1, W)
            ty = torch.linspace(0, H * level - 1, H)
            x, y = torch.meshgrid(tx, ty)
            x = (x + 0.5).reshape(-1)
            y = (y + 0.5).reshape(-1)
            coord = torch.stack([(x - cx) / fx, -(y - cy) / fy], -1)
            dirs = torch.empty(coord.shape[0], 3)
            dirs[..., 0] = coord[..., 0].float()
            dirs[..., 1] = coord[..., 1].float()
            dirs[..., 2] = -1.0
            c = torch.from_numpy(c2w).unsqueeze(0).repeat(coord.shape[0], 1, 1)
            dirs = torch.sum(dirs[..., None, :] * c[..., :3, :3], dim=-1)
            dirs = torch.nn.functional.normalize(dirs, dim=-1)
            orig = c[..., :3, 3]
            rays[f"gen_rays_{tag}_l{level}__o"] = orig.reshape(W, H, 3).permute(1, 0, 2).numpy()
            rays[f"gen_rays_{tag}_l{level}__d"] = dirs.reshape(W, H, 3).permute(1, 0, 2).numpy()
    np.savez(os.path.join(args.out, "rays.npz"), **rays)

    # ---- (7b) the OPENCV_FISHEYE branch of _generate_rays (provider.py:415-433): the reference's own radial_and_tangential_undistort
    # (provider_utils.py:197-234) is CALLED here; the lines around it (pixel grid, coord stack, theta mapping, rotation, normalise, permute)
    # are executed on the same inputs as for the pinhole branch above.
    import math
    fish = {}
    for tag, (H, W), level, dist in (("24x40", (24, 40), 1, [-0.05, 0.02, -0.004, 0.0007, 1e-3, -2e-3]),
                                     ("32", (32, 32), 2, [0.11, -0.03, 0.0, 0.0, 0.0, 0.0])):
        c2w, fx, fy, cx, cy = scene_rays(H, W, view=5, fovy_deg=140.0)
        cx, cy = cx * level + 0.3, cy * level - 0.2                  # principal point in full-resolution pixels, off the pixel centres
        fx, fy = fx * level, fy * level
        distortion_params = torch.Tensor(dist)
        tx = torch.linspace(0, W * level - 1, W)
        ty = torch.linspace(0, H * level - 1, H)
        x, y = torch.meshgrid(tx, ty)
        x = (x + 0.5).reshape(-1)
        y = (y + 0.5).reshape(-1)
        coord = torch.stack([(x - cx) / fx, -(y - cy) / fy], -1)
        coord_x_offset = torch.stack([(x - cx + 1) / fx, -(y - cy) / fy], -1)
        coord_y_offset = torch.stack([(x - cx) / fx, -(y - cy + 1) / fy], -1)
        coord_stack = torch.stack([coord, coord_x_offset, coord_y_offset], dim=0)
        directions_stack = torch.empty((3,) + (coord_stack.shape[1],) + (3,))
        coord_stack = ref_pu.radial_and_tangential_undistort(coord_stack, distortion_params.squeeze().unsqueeze(0).repeat(coord_stack.shape[1], 1))
        theta = torch.sqrt(torch.sum(coord_stack ** 2, dim=-1))
        theta = torch.clip(theta, 0.0, math.pi)
        sin_theta = torch.sin(theta)
        directions_stack[..., 0] = coord_stack[..., 0] * sin_theta / theta
        directions_stack[..., 1] = coord_stack[..., 1] * sin_theta / theta
        directions_stack[..., 2] = -torch.cos(theta)
        c = torch.from_numpy(c2w).unsqueeze(0).repeat(coord_stack.shape[1], 1, 1)
        directions_stack = torch.sum(directions_stack[..., None, :] * c[..., :3, :3], dim=-1)
        directions_stack = torch.nn.functional.normalize(directions_stack, dim=-1)
        fish[f"{tag}__c2w"] = c2w
        fish[f"{tag}__intr"] = np.array([fx, fy, cx, cy, H, W, level], np.float64)
        fish[f"{tag}__dist"] = np.array(dist, np.float32)
        fish[f"{tag}__undist"] = coord_stack[0].numpy()
        fish[f"{tag}__o"] = c[..., :3, 3].reshape(W, H, 3).permute(1, 0, 2).numpy()
        fish[f"{tag}__d"] = directions_stack[0].reshape(W, H, 3).permute(1, 0, 2).numpy()
    np.savez(os.path.join(args.out, "rays_fisheye.npz"), **fish)

    # ---- (2)+(3) weights_sum_i and the full run() dict with a closed-form field plugged in (renderer.py:278-474)
    class ToyRenderer(ref_renderer.NeRFRenderer):
        def __init__(self, opt):
            super().__init__(opt)
            self.f = ToyField()

        def forward(self, x, d):
            return self.f(x, d)

        def density(self, x):
            return self.f.density(x)

    def make_opt(**kw):
        o = argparse.Namespace(bound=2.0, cuda_ray=False, min_near=0.01, density_thresh=10, train_conf=0.01, soft_mask=True,
                               conf_thr=0.5, detach_bg=False, detach_mask_from_field=False, backbone='grid')
        o.__dict__.update(kw)
        return o

    H = W = 32
    c2w, fx, fy, cx, cy = scene_rays(H, W, view=1, opencv=True)
    pose = torch.eye(4).unsqueeze(0).clone()
    pose[0, :3, :4] = torch.from_numpy(c2w)
    r = ref_pu.get_rays(pose, (fx, fy, cx, cy), H, W, -1)
    rays_o, rays_d = r['rays_o'].contiguous(), r['rays_d'].contiguous()      # [1, 1024, 3]

    run_out = {"rays_o": rays_o.numpy(), "rays_d": rays_d.numpy()}
    cases = {
        "train_T8": dict(opt={}, training=True, kw=dict(num_steps=8, upsample_steps=8, perturb=True)),
        "train_T64": dict(opt={}, training=True, stride=4, kw=dict(num_steps=64, upsample_steps=64, perturb=True)),
        "eval_T64": dict(opt={}, training=False, stride=4, kw=dict(num_steps=64, upsample_steps=64, perturb=False)),
        "train_T16_hardmask": dict(opt=dict(soft_mask=False), training=True, stride=4, kw=dict(num_steps=16, upsample_steps=16, perturb=True)),
        "train_T16_detach": dict(opt=dict(detach_bg=True, detach_mask_from_field=True), training=True, stride=4,
                                 kw=dict(num_steps=16, upsample_steps=16, perturb=True)),
    }
    import io
    import contextlib
    for tag, c in cases.items():
        with contextlib.redirect_stdout(io.StringIO()):
            model = ToyRenderer(make_opt(**c["opt"]))
        model.train(c["training"])
        torch.manual_seed(11)
        st = c.get("stride", 1)
        run_out[f"{tag}__stride"] = np.int64(st)
        with _Recorder() as rec:
            res = model.run(rays_o[:, ::st].contiguous(), rays_d[:, ::st].contiguous(), **c["kw"])
        kinds = [k for k, _ in rec.draws]
        assert kinds[0] == "randn"
        run_out[f"{tag}__light"] = rec.draws[0][1].numpy()
        if c["kw"]["perturb"]:
            run_out[f"{tag}__z"] = rec.draws[1][1].numpy()
        if c["training"]:
            run_out[f"{tag}__u"] = rec.draws[-1][1].numpy()
        for k in ("image", "depth", "render_mask", "weights_sum", "weights", "mask", "sigma", "rgbs", "edit_mask"):
            run_out[f"{tag}__{k}"] = res[k].detach().numpy()
        for sub in ("fg", "bg"):
            for k in ("image", "depth", "render_mask", "weights_sum", "weights"):
                run_out[f"{tag}__{sub}_{k}"] = res[sub][k].detach().numpy()
    np.savez_compressed(os.path.join(args.out, "run.npz"), **run_out)

    # weights_sum_i stand-alone (random sigma / rgb / mask / z incl. soft_mask-style sigmas and detach_bg)
    torch.manual_seed(5)
    N, T = 96, 24
    sig = (torch.rand(N, T, 1) * 8) ** 2
    rgb = torch.rand(N, T, 3)
    msk = torch.rand(N, T, 1)
    z = torch.sort(torch.rand(N, T) * 3 + 0.2, dim=-1).values
    nears, fars = z[:, :1] - 0.1, z[:, -1:] + 0.3
    sd = (fars - nears) / 12
    ws = {"sigmas": sig.numpy(), "rgbs": rgb.numpy(), "masks": msk.numpy(), "z": z.numpy(), "nears": nears.numpy(),
          "fars": fars.numpy(), "sample_dist": sd.numpy()}
    for tag, o in (("plain", {}), ("detach", dict(detach_bg=True, detach_mask_from_field=True))):
        with contextlib.redirect_stdout(io.StringIO()):
            model = ToyRenderer(make_opt(**o))
        s, c_ = sig.clone().requires_grad_(True), rgb.clone().requires_grad_(True)
        res = model.weights_sum_i(sd, s, None, None, None, z, nears, fars, c_, (1, N), masks=msk, is_all=True)
        loss = (res['image'] ** 2).sum() + res['weights_sum'].sum() + (res['render_mask'] * 0.3).sum() + res['depth'].sum()
        loss.backward()
        for k in ("image", "depth", "render_mask", "weights_sum", "weights", "mask"):
            ws[f"{tag}__{k}"] = res[k].detach().numpy()
        ws[f"{tag}__grad_sigmas"] = s.grad.numpy()
        ws[f"{tag}__grad_rgbs"] = c_.grad.numpy()
    np.savez_compressed(os.path.join(args.out, "weights_sum_i.npz"), **ws)

    # ---- dataset front-end: pose normalisation of NerfstudioData._load_renderings (provider.py:226-236; provider_utils.py:35-115)
    g = torch.Generator().manual_seed(11)
    V = 23
    ang = torch.rand(V, generator=g) * 6.28
    eye = torch.stack([3.0 * torch.cos(ang), 0.8 + 0.3 * torch.rand(V, generator=g), 3.0 * torch.sin(ang)], -1) + torch.tensor([0.7, -0.4, 1.1])
    fwd = torch.nn.functional.normalize(-eye + 0.2 * torch.randn(V, 3, generator=g), dim=-1)
    right = torch.nn.functional.normalize(torch.cross(fwd, torch.tensor([0.05, 1.0, 0.1]).expand(V, 3), dim=-1), dim=-1)
    up = torch.cross(right, fwd, dim=-1)
    poses = torch.eye(4).repeat(V, 1, 1)
    poses[:, :3, 0], poses[:, :3, 1], poses[:, :3, 2], poses[:, :3, 3] = right, up, -fwd, eye
    orient = {"poses": poses.numpy()}
    for method in ("up", "none"):
        for center in (True, False):
            out, tr = ref_pu.auto_orient_and_center_poses(poses.clone(), method=method, center_poses=center)
            orient[f"{method}_{int(center)}__poses"] = out.numpy()
            orient[f"{method}_{int(center)}__transform"] = tr.numpy()
    np.savez(os.path.join(args.out, "orient.npz"), **orient)

    # ---- occupancy-grid refresh: the reference's own update_extra_state loop (renderer.py:1658-1715) on a 16^3 grid (grid_size is an attribute;
    # the 128 of the constructor would need 50 MB of recorded jitter).  The two native calls of the loop are bit manipulation only and are
    # stubbed in numpy: morton3D = 10-bit interleave (raymarching.cu:56-69), packbits = bit i of byte n is grid[8 n + i] > thresh (:267-289).
    def _expand(v):
        v = v.astype(np.uint32)
        v = (v * np.uint32(0x00010001)) & np.uint32(0xFF0000FF)
        v = (v * np.uint32(0x00000101)) & np.uint32(0x0F00F00F)
        v = (v * np.uint32(0x00000011)) & np.uint32(0xC30C30C3)
        v = (v * np.uint32(0x00000005)) & np.uint32(0x49249249)
        return v

    def _morton3D(coords):
        c = coords.numpy().astype(np.uint32)
        return torch.from_numpy((_expand(c[:, 0]) | (_expand(c[:, 1]) << np.uint32(1)) | (_expand(c[:, 2]) << np.uint32(2))).astype(np.int32))

    def _packbits(grid, thresh, bitfield=None):
        bits = (grid.reshape(-1, 8).numpy() > np.float32(thresh)).astype(np.uint8)
        return torch.from_numpy((bits << np.arange(8, dtype=np.uint8)).sum(-1).astype(np.uint8))

    ref_renderer.raymarching.morton3D = _morton3D
    ref_renderer.raymarching.packbits = _packbits
    Hg = 16
    with contextlib.redirect_stdout(io.StringIO()):
        occ = ToyRenderer(make_opt(cuda_ray=True, density_thresh=10))
    occ.grid_size = Hg
    g = torch.Generator().manual_seed(21)
    grid0 = torch.rand(occ.cascade, Hg ** 3, generator=g) * 20.0
    grid0[torch.rand(occ.cascade, Hg ** 3, generator=g) < 0.1] = -1.0           # invalid cells stay untouched (:1707)
    grid0[torch.rand(occ.cascade, Hg ** 3, generator=g) < 0.3] = 0.0
    occ.density_grid = grid0.clone()
    occ.density_bitfield = torch.zeros(occ.cascade * Hg ** 3 // 8, dtype=torch.uint8)
    occ_out = {"grid0": grid0.numpy(), "grid_size": np.int64(Hg), "cascade": np.int64(occ.cascade), "bound": np.float32(occ.bound), "decay": np.float32(0.95),
               "density_thresh": np.float32(occ.density_thresh)}
    rand_like = torch.rand_like
    for rnd in range(2):                                                          # two refreshes: the second one sees the first one's grid
        draws = []

        def rec_rand_like(t, *a, **k):
            r = rand_like(t, *a, **k)
            draws.append(r.clone())
            return r
        torch.manual_seed(30 + rnd)
        occ.local_step = 3 + rnd
        occ.step_counter[:4, 0] = torch.tensor([100, 200, 301, 77], dtype=torch.int32)
        torch.rand_like = rec_rand_like
        try:
            occ.update_extra_state(decay=0.95, S=Hg)
        finally:
            torch.rand_like = rand_like
        assert len(draws) == occ.cascade
        occ_out[f"r{rnd}__rand"] = torch.stack(draws).numpy()
        occ_out[f"r{rnd}__grid"] = occ.density_grid.numpy().copy()
        occ_out[f"r{rnd}__mean_density"] = np.float64(occ.mean_density)
        occ_out[f"r{rnd}__bitfield"] = occ.density_bitfield.numpy().copy()
        occ_out[f"r{rnd}__mean_count"] = np.int64(occ.mean_count)
    np.savez_compressed(os.path.join(args.out, "occupancy.npz"), **occ_out)

    # ---- SDS guidance: the reference's own StableDiffusion.train_step (nerf/sd.py:115-155).  The class is built without its __init__ (which
    # downloads the pipeline); third-party pieces are stand-ins with the PUBLISHED definitions: the scheduler's scaled-linear betas and
    # add_noise(x, n, t) = sqrt(abar_t) x + sqrt(1 - abar_t) n; the UNet is the closed-form toy_eps of oracle/toy_field.py.
    for name in ("transformers", "diffusers", "torchvision"):
        _stub(name)
    sys.modules["transformers"].logging = types.SimpleNamespace(set_verbosity_error=lambda: None)
    sys.modules["diffusers"].DiffusionPipeline = object
    sys.modules["torchvision"].transforms = types.SimpleNamespace()
    from nerf import sd as ref_sd
    from oracle.toy_field import toy_eps
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2
    abar = torch.cumprod(1.0 - betas, dim=0)

    class _Sched:
        alphas_cumprod = abar

        def add_noise(self, x, n, t):
            a = abar[t].reshape(-1, 1, 1, 1)
            return a.sqrt() * x + (1 - a).sqrt() * n

    poison = {}

    class _UNet:
        def __call__(self, x, t, encoder_hidden_states=None, class_labels=None):
            e = toy_eps(x, t.float(), encoder_hidden_states)
            if poison:
                e = e.clone()
                e[1, 0, 0, 0], e[1, 1, 2, 3], e[0, 2, 1, 1] = float("nan"), float("inf"), float("-inf")
            return types.SimpleNamespace(sample=e)

    class _UNetHalfIO(_UNet):
        """the same closed form with HALF-PRECISION input and output — what the real UNet has under the reference's fp16 autocast (main.py:152-153:
        its first convolution casts the input to half, its last one emits half).  The product carries exactly these two roundings, so against these
        cases (tags ending in `_h`) it can be held to 1e-3 where the float32-epsilon cases need cfg x half-epsilon = a few per cent."""
        def __call__(self, x, t, encoder_hidden_states=None, class_labels=None):
            return types.SimpleNamespace(sample=super().__call__(x.half().float(), t, encoder_hidden_states).sample.half().float())

    sds_out = {"alphas_cumprod": abar.numpy()}
    g = torch.Generator().manual_seed(44)
    text = torch.randn(2, 77, 16, generator=g)
    sds_out["text"] = text.numpy()
    cases = {"plain": dict(t_ratio=1), "local": dict(t_ratio=0.4), "stage_late": dict(t_ratio=1, stage_time=True, step=900),
             "stage_early": dict(t_ratio=1, stage_time=True, step=100), "nonfinite": dict(t_ratio=1, poison=True),
             "plain_h": dict(t_ratio=1, half_io=True), "local_h": dict(t_ratio=0.4, half_io=True)}
    for tag, c in cases.items():
        guide = ref_sd.StableDiffusion.__new__(ref_sd.StableDiffusion)
        torch.nn.Module.__init__(guide)
        guide.device = "cpu"
        guide.opt = argparse.Namespace(cfg=100.0, stage_time=bool(c.get("stage_time")), iters=1000, lambda_sd=0.01, max_ratio=0.98)
        guide.scheduler, guide.unet = _Sched(), (_UNetHalfIO() if c.get("half_io") else _UNet())
        guide.num_train_timesteps = 1000
        guide.min_step, guide.max_step = int(1000 * 0.02), int(1000 * 0.98)                 # sd.py:69-70
        guide.alphas = abar
        poison.clear()
        if c.get("poison"):
            poison["on"] = True
        lat = (torch.randn(1, 4, 8, 8, generator=g) * 0.8).requires_grad_(True)
        torch.manual_seed(70 + len(sds_out))
        drawn = []
        randint, randn_like = torch.randint, torch.randn_like

        def rec_randint(*a, **k):
            k.pop("device", None)
            r = randint(*a, **k)
            drawn.append(("randint", a[0], a[1], r.clone()))
            return r

        def rec_randn_like(t_, *a, **k):
            r = randn_like(t_, *a, **k)
            drawn.append(("randn_like", r.clone()))
            return r
        torch.randint, torch.randn_like = rec_randint, rec_randn_like
        try:
            loss, ld = guide.train_step(lat, text, system=types.SimpleNamespace(global_step=c.get("step", 0)), t_ratio=c["t_ratio"])
        finally:
            torch.randint, torch.randn_like = randint, randn_like
        loss.backward()
        assert [d[0] for d in drawn] == ["randint", "randn_like"]
        sds_out[f"{tag}__latents"] = lat.detach().numpy()
        sds_out[f"{tag}__t_ratio"] = np.float64(c["t_ratio"])
        sds_out[f"{tag}__stage_time"] = np.int64(bool(c.get("stage_time")))
        sds_out[f"{tag}__global_step"] = np.int64(c.get("step", 0))
        sds_out[f"{tag}__randint_lo_hi"] = np.array([drawn[0][1], drawn[0][2]], np.int64)   # the range the reference drew from (hi exclusive)
        sds_out[f"{tag}__t_draw"] = drawn[0][3].numpy()
        sds_out[f"{tag}__noise"] = drawn[1][1].numpy()
        sds_out[f"{tag}__loss"] = np.float64(loss.item())
        sds_out[f"{tag}__loss_sds"] = np.float64(ld["loss_sds"])
        sds_out[f"{tag}__grad_latents"] = lat.grad.numpy()
    np.savez_compressed(os.path.join(args.out, "sds.npz"), **sds_out)

    # ---- the composed editing step: the reference's own Trainer_Nerf.train_step_editing / train_step_sd / get_pt (nerf/utils_init_nerf.py:243-308,
    # 353-394), the object built without its __init__; model / pretrained model = the reference renderer with the toy field (six parameters `theta`
    # for a parameter gradient), guidance = the reference StableDiffusion of the section above + the closed-form toy VAE.
    for name in ("cv2", "imageio", "tensorboardX", "clip", "torch_ema", "lpips"):
        _stub(name)
    _stub("torchvision.transforms")
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    from nerf import utils_init_nerf as ref_tr
    from oracle.toy_field import toy_vae_latents

    class _VAE:
        def encode(self, x):
            return types.SimpleNamespace(latent_dist=types.SimpleNamespace(sample=lambda: toy_vae_latents(x)))

    class ToyParamRenderer(ref_renderer.NeRFRenderer):
        def __init__(self, opt, theta):
            super().__init__(opt)
            self.theta = torch.nn.Parameter(theta.clone())
            self.f = ToyField(self.theta)

        def forward(self, x, d):
            return self.f(x, d)

        def density(self, x):
            return self.f.density(x)

    He = We = 16
    c2w, fx, fy, cx, cy = scene_rays(He, We, view=2, opencv=True)
    pose = torch.eye(4).unsqueeze(0).clone()
    pose[0, :3, :4] = torch.from_numpy(c2w)
    r = ref_pu.get_rays(pose, (fx, fy, cx, cy), He, We, -1)
    e_o, e_d = r['rays_o'].contiguous(), r['rays_d'].contiguous()
    g = torch.Generator().manual_seed(91)
    e_rgbs, e_mask = torch.rand(1, He * We, 3, generator=g), (torch.rand(1, He * We, 1, generator=g) > 0.5).float()
    theta_edit, theta_pre = torch.tensor([0.10, 0.6, 0.2, -0.3, 0.1, 0.4]), torch.tensor([-0.05, 0.3, -0.1, 0.2, 0.0, -0.2])
    tz, tz_fg = torch.randn(2, 77, 16, generator=g), torch.randn(2, 77, 16, generator=g)
    ed_out = {"rays_o": e_o.numpy(), "rays_d": e_d.numpy(), "rgbs": e_rgbs.numpy(), "mask": e_mask.numpy(), "H": np.int64(He), "W": np.int64(We),
              "theta_edit": theta_edit.numpy(), "theta_pre": theta_pre.numpy(), "text_z": tz.numpy(), "text_z_fg": tz_fg.numpy(),
              "alphas_cumprod": abar.numpy()}
    ed_base = dict(bound=2.0, cuda_ray=False, min_near=0.01, density_thresh=10, train_conf=0.01, soft_mask=True, conf_thr=0.5, detach_bg=False,
                   detach_mask_from_field=False, backbone='grid', num_steps=8, upsample_steps=8, random_bg_c=False, black_bg_c=False, white_bg_c=False,
                   ori_bg=False, lambda_sd=0.01, keep_bg=1000.0, g_only=False, l_only=False, local_t_ratio=0.4, global_ratio=0.5, clip_view=False,
                   cfg=100.0, stage_time=False, iters=1000, max_ratio=0.98)
    ed_out["opt_keys"] = np.array(sorted(ed_base.keys()))
    for k in ("num_steps", "upsample_steps", "lambda_sd", "keep_bg", "local_t_ratio", "cfg", "train_conf", "conf_thr", "min_near", "bound"):
        ed_out[f"opt__{k}"] = np.float64(ed_base[k])
    for tag, kw in (("g_only", dict(g_only=True)), ("l_only", dict(l_only=True)), ("ori_bg", dict(g_only=True, ori_bg=True)),
                    ("g_only_h", dict(g_only=True)), ("l_only_h", dict(l_only=True))):                  # `_h`: epsilon predictor with half-precision I/O
        o = argparse.Namespace(**dict(ed_base, **kw))
        tr = ref_tr.Trainer_Nerf.__new__(ref_tr.Trainer_Nerf)
        with contextlib.redirect_stdout(io.StringIO()):
            tr.model, tr.model_pretrained = ToyParamRenderer(o, theta_edit), ToyParamRenderer(o, theta_pre)
        tr.model.train(); tr.model_pretrained.train()
        tr.opt, tr.pt_dict, tr.global_step, tr.log_ptr = o, {}, 0, None
        guide = ref_sd.StableDiffusion.__new__(ref_sd.StableDiffusion)
        torch.nn.Module.__init__(guide)
        guide.device, guide.opt = "cpu", o
        guide.scheduler, guide.unet, guide.vae = _Sched(), (_UNetHalfIO() if tag.endswith("_h") else _UNet()), _VAE()
        guide.num_train_timesteps = 1000
        guide.min_step, guide.max_step = int(1000 * 0.02), int(1000 * o.max_ratio)
        guide.alphas = abar
        poison.clear()
        tr.guidance = guide
        tr.text_z, tr.text_z_fg, tr.text_z_norm, tr.text_z_norm_fg = tz, tz_fg, tz, tz_fg
        torch.manual_seed(123)
        drawn = []
        randint, randn_like = torch.randint, torch.randn_like

        def rec_randint(*a, **k):
            k.pop("device", None)
            rr = randint(*a, **k)
            drawn.append(rr.clone())
            return rr

        def rec_randn_like(t_, *a, **k):
            rr = randn_like(t_, *a, **k)
            drawn.append(rr.clone())
            return rr
        torch.randint, torch.randn_like = rec_randint, rec_randn_like
        err = None
        try:
            with _Recorder() as rec:
                try:
                    pred_rgb, pred_ws, loss, ld = tr.train_step_editing((e_rgbs, e_mask, e_o, e_d, He, We, "view2"))
                except RuntimeError as ex:
                    err = str(ex)
        finally:
            torch.randint, torch.randn_like = randint, randn_like
        if tag == "ori_bg":
            # utils_init_nerf.py:378-380 multiplies rgbs [B, 3, H, W] by non_edit [B, H, W, 1]: not broadcastable for H != 3 — the reference raises
            ed_out["ori_bg__raises"] = np.int64(err is not None)
            ed_out["ori_bg__message"] = np.array(err or "")
            continue
        assert err is None, err
        loss.backward()
        kinds = [k for k, _ in rec.draws]
        assert kinds == ["randn", "rand", "rand", "randn", "rand", "rand"], kinds               # edited render, then the pretrained render (get_pt)
        for i, nm in enumerate(("light", "z", "u", "pt_light", "pt_z", "pt_u")):
            ed_out[f"{tag}__{nm}"] = rec.draws[i][1].numpy()
        ed_out[f"{tag}__t_draw"] = drawn[0].numpy()
        ed_out[f"{tag}__noise"] = drawn[1].numpy()
        ed_out[f"{tag}__pred_rgb"] = pred_rgb.detach().numpy()
        ed_out[f"{tag}__pred_ws"] = pred_ws.detach().numpy()
        ed_out[f"{tag}__loss"] = np.float64(loss.item())
        ed_out[f"{tag}__loss_sds"] = np.float64(ld["loss_sds"])
        ed_out[f"{tag}__loss_bg"] = np.float64(ld["loss_bg"])
        ed_out[f"{tag}__grad_theta"] = tr.model.theta.grad.numpy().copy()
    # ---- the reconstruction step: the reference's own Trainer_Nerf.train_step_pretrain (utils_init_nerf.py:194-241), same rays / toy field
    # ("batch": --batch_rays subsampling, :210-215 — the index draw of np.random.choice is recorded)
    for tag, kw in (("conf", dict(train_rgb=1.0, train_conf=0.01, batch_rays=0)), ("conf2", dict(train_rgb=2.5, train_conf=0.05, batch_rays=0)),
                    ("batch", dict(train_rgb=1.0, train_conf=0.01, batch_rays=100))):
        o = argparse.Namespace(**dict(ed_base, **kw))
        tr = ref_tr.Trainer_Nerf.__new__(ref_tr.Trainer_Nerf)
        with contextlib.redirect_stdout(io.StringIO()):
            tr.model = ToyParamRenderer(o, theta_edit)
        tr.model.train()
        tr.opt, tr.device, tr.log_ptr = o, "cpu", None
        torch.manual_seed(321)
        np.random.seed(4242)
        picked = []
        np_choice = np.random.choice

        def rec_choice(*a, **k):
            sel = np_choice(*a, **k)
            picked.append(np.asarray(sel).copy())
            return sel
        np.random.choice = rec_choice
        try:
            with _Recorder() as rec:
                pred_rgb, mask_volume, loss, ld = tr.train_step_pretrain((e_rgbs, e_mask, e_o, e_d, He, We, "view2"))
        finally:
            np.random.choice = np_choice
        loss.backward()
        assert [k for k, _ in rec.draws] == ["randn", "rand", "rand"]
        assert len(picked) == (1 if kw["batch_rays"] else 0)
        ed_out[f"pre_{tag}__batch_rays"] = np.int64(kw["batch_rays"])
        ed_out[f"pre_{tag}__select_inds"] = picked[0].astype(np.int64) if picked else np.zeros(0, np.int64)
        for i, nm in enumerate(("light", "z", "u")):
            ed_out[f"pre_{tag}__{nm}"] = rec.draws[i][1].numpy()
        ed_out[f"pre_{tag}__train_rgb"], ed_out[f"pre_{tag}__train_conf"] = np.float64(kw["train_rgb"]), np.float64(kw["train_conf"])
        ed_out[f"pre_{tag}__pred_rgb"] = pred_rgb.detach().numpy()
        ed_out[f"pre_{tag}__mask_volume"] = mask_volume.detach().numpy()
        ed_out[f"pre_{tag}__loss"] = np.float64(loss.item())
        ed_out[f"pre_{tag}__loss_c"] = np.float64(ld["loss_c"])
        ed_out[f"pre_{tag}__loss_m"] = np.float64(ld.get("loss_m", 0.0))
        ed_out[f"pre_{tag}__grad_theta"] = tr.model.theta.grad.numpy().copy()
    np.savez_compressed(os.path.join(args.out, "editing.npz"), **ed_out)

    # ---- the field's glue: the reference's own NeRFNetwork.forward / density (nerf/network_grid.py:66-193) with its real GridEncoder wrapper
    # (gridencoder/grid.py), get_encoder (encoding.py), get_embedder (base.py) and trunc_exp — input mapping, [L, B, C] permute, gaussian blob, the
    # [dir embedding, features] concatenation order, output activations.  The two native pieces underneath are stand-ins BY THIS BUILD (so this
    # section pins the glue, not them): `_gridencoder.grid_encode_forward / _backward` = the C restatement (oracle/gridencoder_ref.c),
    # `tinycudann.Network` = oracle.torch_oracle.mlp_forward on one flat parameter vector (the layout assumption of DESIGN.md section 2).
    from oracle import c_oracle as co_
    from oracle import torch_oracle as to_

    class _TcnnNetwork(torch.nn.Module):
        def __init__(self, n_input_dims, n_output_dims, network_config):
            super().__init__()
            self.cfg = (n_input_dims, n_output_dims, network_config["n_neurons"], network_config["n_hidden_layers"])
            self.act = network_config["output_activation"]
            self.params = torch.nn.Parameter(torch.zeros(to_.mlp_n_params(*self.cfg)))

        def forward(self, x):
            return to_.mlp_forward(x.float(), self.params, *self.cfg, self.act, False)
    sys.modules["tinycudann"].Network = _TcnnNetwork

    def _ge_fwd(inputs, embeddings, offsets, outputs, B, D, C, L, max_level, S, Hb, dy_dx, gridtype, align_corners, interp):
        out, _ = co_.grid_encode_forward(inputs.detach().numpy(), embeddings.detach().numpy(), offsets.numpy(), float(2.0 ** S), int(Hb), False, int(gridtype),
                                         bool(align_corners), int(interp), int(max_level))
        outputs.copy_(torch.from_numpy(out.reshape(B, L, C).transpose(1, 0, 2).copy()))

    def _ge_bwd(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, max_level, S, Hb, dy_dx, grad_inputs, gridtype, align_corners, interp):
        gflat = grad.detach().numpy().transpose(1, 0, 2).reshape(B, L * C)               # the wrapper hands over [L, B, C]
        ge, _ = co_.grid_encode_backward(gflat, inputs.detach().numpy(), tuple(embeddings.shape), offsets.numpy(), float(2.0 ** S), int(Hb), None,
                                         int(gridtype), bool(align_corners), int(interp), int(max_level))
        grad_embeddings.add_(torch.from_numpy(ge))
    import gridencoder.grid as _rg
    _rg._backend.grid_encode_forward, _rg._backend.grid_encode_backward = _ge_fwd, _ge_bwd
    from nerf import network_grid as ref_ng
    fo = argparse.Namespace(bound=2.0, cuda_ray=False, min_near=0.01, density_thresh=10, train_conf=0.01, detach_mask_from_field=False, mask_no_dir=False,
                            keyword2=None)
    with contextlib.redirect_stdout(io.StringIO()):
        net = ref_ng.NeRFNetwork(fo, device=torch.device("cpu"))
    n_emb = net.pos_en.embeddings.shape[0]
    with torch.no_grad():                                                               # a closed-form table: the tests rebuild it from the index
        idx = torch.arange(n_emb, dtype=torch.float64)
        net.pos_en.embeddings.copy_(torch.stack([torch.sin(idx * 0.37) * 0.5, torch.cos(idx * 0.11 + 1.3) * 0.5], -1).float())
    g = torch.Generator().manual_seed(77)
    fld = {"n_embeddings": np.int64(n_emb), "offsets": net.pos_en.offsets.numpy(), "per_level_scale": np.float64(net.pos_en.per_level_scale),
           "gridtype": np.array(net.pos_en.gridtype), "bound": np.float32(fo.bound)}
    for nm, mod in (("network", net.network), ("density_network", net.density_network), ("rgb_network", net.rgb_network)):
        with torch.no_grad():
            mod.params.copy_(to_.xavier_params(*mod.cfg, generator=g))
        fld[f"{nm}__params"] = mod.params.detach().numpy().copy()
        fld[f"{nm}__cfg"] = np.array(mod.cfg, np.int64)
    P = 192
    fx_ = (torch.rand(P, 3, generator=g) * 2 - 1) * 1.9
    fx_[:8] *= 0.05                                                                     # points inside the gaussian blob
    fd_ = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1)
    w_s, w_r = torch.rand(P, generator=g), torch.rand(P, 4, generator=g)
    sigma, rad, _ = net(fx_, fd_)
    dens = net.density(fx_)["sigma"]
    loss = (sigma * w_s * 0.01).sum() + (rad * w_r).sum()
    loss.backward()
    ge = net.pos_en.embeddings.grad
    nz = torch.nonzero(ge.abs().sum(-1)).reshape(-1)
    fld.update({"x": fx_.numpy(), "d": fd_.numpy(), "w_sigma": w_s.numpy(), "w_rad": w_r.numpy(), "sigma": sigma.detach().numpy(), "radiances": rad.detach().numpy(),
                "density_sigma": dens.detach().numpy(), "loss": np.float64(loss.item()), "grad_emb_idx": nz.numpy(), "grad_emb_val": ge[nz].numpy()})
    for nm, mod in (("network", net.network), ("density_network", net.density_network), ("rgb_network", net.rgb_network)):
        fld[f"{nm}__grad"] = mod.params.grad.numpy().copy()
    np.savez_compressed(os.path.join(args.out, "field.npz"), **fld)
    print("golden vectors written to", args.out)


if __name__ == "__main__":
    main()
