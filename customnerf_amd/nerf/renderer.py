"""NeRFRenderer — counterpart of the hot-path part of the reference's nerf/renderer.py.

Implemented (reference lines): sample_pdf (21-55), NeRFRenderer.__init__ (199-243), reset_extra_state (266-276),
run (278-405), weights_sum_i (407-474), run_cuda (597-718), update_extra_state (1658-1715), render (1719-1733).
Out of scope: the SDF/NeuS paths, mesh export, the unused run_cuda2 / render_cuda duplicates.

Reference defects handled deliberately (SURVEY.md "Known reference defects"):
  * run_cuda reads `opt.bg_color` which argparse never defines -> read with getattr(..., None);
  * run_cuda feeds 4-channel rgb+confidence rows to a stride-3 compositor -> our compositor takes the row stride;
  * run() with upsample_steps == 0 referenced an undefined name -> handled;
  * the output-dead density pass over the fine samples (renderer.py:353) is skipped unless
    `opt.eval_fine_density` asks for it (output-identical, tests/test_oracle_golden.py proves it on the reference).
"""
import math

import numpy as np
import torch
import torch.nn as nn

from .. import raymarching
from .provider_utils import custom_meshgrid, safe_normalize
from . import render_ops


def sample_pdf(bins, weights, n_samples, det=False, u=None):
    """renderer.py:21-55.  bins [B,T], weights [B,T-1] -> [B,n_samples].  `u` replays the random draw of :37."""
    weights = weights + 1e-5
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    if det:
        u = torch.linspace(0. + 0.5 / n_samples, 1. - 0.5 / n_samples, steps=n_samples, device=weights.device)
        u = u.expand(list(cdf.shape[:-1]) + [n_samples])
    elif u is None:
        u = torch.rand(list(cdf.shape[:-1]) + [n_samples], device=weights.device)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    cdf_b, cdf_a = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    bins_b, bins_a = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    denom = cdf_a - cdf_b
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_b) / denom
    return bins_b + t * (bins_a - bins_b)


class _Lazy:
    """a result-dict entry that is computed on first access (entries the training loop never reads cost no launches)"""
    __slots__ = ('fn', 'value', 'done')

    def __init__(self, fn):
        self.fn, self.value, self.done = fn, None, False

    def get(self):
        if not self.done:
            self.value, self.done, self.fn = self.fn(), True, None
        return self.value


class _LazyResults(dict):
    """dict whose _Lazy values are materialised by [] / get / items / values (keys, `in` and len behave as for a plain dict)"""

    def __getitem__(self, k):
        v = dict.__getitem__(self, k)
        if isinstance(v, _Lazy):
            v = v.get()
            dict.__setitem__(self, k, v)
        return v

    def get(self, k, default=None):
        return self[k] if k in self else default

    def items(self):
        return [(k, self[k]) for k in dict.keys(self)]

    def values(self):
        return [self[k] for k in dict.keys(self)]


class NeRFRenderer(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.bound = opt.bound
        self.cascade = 1 + math.ceil(math.log2(opt.bound))
        self.grid_size = 128
        self.cuda_ray = opt.cuda_ray
        self.min_near = opt.min_near
        self.density_thresh = opt.density_thresh

        aabb_train = torch.FloatTensor([-opt.bound, -opt.bound, -opt.bound, opt.bound, opt.bound, opt.bound])
        self.register_buffer('aabb_train', aabb_train)
        self.register_buffer('aabb_infer', aabb_train.clone())
        self.register_buffer('aabb_train_bg', 2 * aabb_train.clone())
        self.register_buffer('aabb_infer_bg', 2 * aabb_train.clone())

        if self.cuda_ray:                                                        # renderer.py:230-243
            self.register_buffer('density_grid', torch.zeros([self.cascade, self.grid_size ** 3]))
            self.register_buffer('density_bitfield', torch.zeros(self.cascade * self.grid_size ** 3 // 8, dtype=torch.uint8))
            self.mean_density = 0
            self.iter_density = 0
            self.register_buffer('step_counter', torch.zeros(16, 2, dtype=torch.int32))
            self.mean_count = 0
            self.local_step = 0

    def forward(self, x, d):
        raise NotImplementedError()

    def density(self, x):
        raise NotImplementedError()

    def reset_extra_state(self):
        if not self.cuda_ray:
            return
        self.density_grid.zero_()
        self.mean_density = 0
        self.iter_density = 0
        self.step_counter.zero_()
        self.mean_count = 0
        self.local_step = 0

    # ------------------------------------------------------------------------------------------ run (pure-torch path)
    def run(self, rays_o, rays_d, num_steps=128, upsample_steps=128, light_d=None, ambient_ratio=1.0, shading='albedo',
            bg_color=None, perturb=False, _draws=None, **kwargs):
        """renderer.py:278-405.  rays_o, rays_d [B,N,3] (B == 1) -> result dict.
        `_draws` = dict(light, z, u) replays the RNG draws of :305, :317 and sample_pdf:37 (tests)."""
        if (getattr(self.opt, 'fused_render', True) and rays_o.is_cuda and getattr(self.opt, 'train_conf', 0) and upsample_steps >= 2
                and 3 <= num_steps <= 128 and upsample_steps <= 128 and getattr(self, 'supports_dir_group', False)):
            return self._run_fused(rays_o, rays_d, num_steps, upsample_steps, perturb, _draws)
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        N = rays_o.shape[0]
        device = rays_o.device
        draws = _draws or {}
        results = {}
        aabb = self.aabb_train if self.training else self.aabb_infer

        nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, aabb, self.min_near)
        nears = nears.unsqueeze(-1)
        fars = fars.unsqueeze(-1)

        if light_d is None:                                                       # :303-306 (consumes RNG; value unused)
            light_d = rays_o[0] + (draws['light'].to(device) if 'light' in draws else torch.randn(3, device=device, dtype=torch.float))
            light_d = safe_normalize(light_d)

        z_vals = torch.linspace(0.0, 1.0, num_steps, device=device).unsqueeze(0).expand((N, num_steps))
        z_vals = nears + (fars - nears) * z_vals
        sample_dist = (fars - nears) / num_steps
        if perturb:
            zr = draws['z'].to(device) if 'z' in draws else torch.rand(z_vals.shape, device=device)
            z_vals = z_vals + (zr - 0.5) * sample_dist

        xyzs = rays_o.unsqueeze(-2) + rays_d.unsqueeze(-2) * z_vals.unsqueeze(-1)
        xyzs = torch.min(torch.max(xyzs, aabb[:3]), aabb[3:])

        if upsample_steps > 0:
            with torch.no_grad():
                sig_c = self.density(xyzs.reshape(-1, 3))['sigma'].view(N, num_steps).float()
                deltas = z_vals[..., 1:] - z_vals[..., :-1]
                deltas = torch.cat([deltas, sample_dist * torch.ones_like(deltas[..., :1])], dim=-1)
                alphas = 1 - torch.exp(-deltas * sig_c)
                alphas_shifted = torch.cat([torch.ones_like(alphas[..., :1]), 1 - alphas + 1e-15], dim=-1)
                weights = alphas * torch.cumprod(alphas_shifted, dim=-1)[..., :-1]
                z_vals_mid = (z_vals[..., :-1] + 0.5 * deltas[..., :-1])
                u = draws['u'].to(device) if 'u' in draws else None
                new_z_vals = sample_pdf(z_vals_mid, weights[:, 1:-1], upsample_steps, det=not self.training, u=u).detach()
                new_xyzs = rays_o.unsqueeze(-2) + rays_d.unsqueeze(-2) * new_z_vals.unsqueeze(-1)
                new_xyzs = torch.min(torch.max(new_xyzs, aabb[:3]), aabb[3:])
                if getattr(self.opt, 'eval_fine_density', False):               # :353, output-dead in the reference
                    self.density(new_xyzs.reshape(-1, 3))
            z_vals = torch.cat([z_vals, new_z_vals], dim=1)
            z_vals, z_index = torch.sort(z_vals, dim=1)
            xyzs = torch.cat([xyzs, new_xyzs], dim=1)
            xyzs = torch.gather(xyzs, dim=1, index=z_index.unsqueeze(-1).expand_as(xyzs))

        dirs = rays_d.view(-1, 1, 3).expand_as(xyzs)
        if getattr(self, 'supports_dir_group', False):
            # the fused field reads one direction per ray (samples of a ray share rays_d): no [N*S,3] expansion copy
            sigmas, rgbs, normals = self(xyzs.reshape(-1, 3), rays_d, dir_group=xyzs.shape[1])
        else:
            sigmas, rgbs, normals = self(xyzs.reshape(-1, 3), dirs.reshape(-1, 3))
        if rgbs.shape[-1] > 3:
            n_dim = rgbs.shape[-1] - 3
            rgbs, masks = rgbs.split([3, n_dim], dim=-1)
            masks = masks.reshape(N, -1, n_dim).float()
        else:
            masks = None
        sigmas = sigmas.view(N, -1, 1)
        rgbs = rgbs.reshape(N, -1, 3).float()

        if getattr(self.opt, 'train_conf', 0):
            results = self.weights_sum_i(sample_dist, sigmas, None, dirs, None, z_vals, nears, fars, rgbs, prefix, masks=masks, is_all=True)
            if getattr(self.opt, 'soft_mask', False):                             # :386-389
                edit_mask = torch.sigmoid((masks - self.opt.conf_thr) * 100)
                sigmas_fg = sigmas * edit_mask
                sigmas_bg = sigmas * (1 - edit_mask)
            else:                                                                 # :391-395
                edit_mask = masks > 0.5
                sigmas_bg = torch.where(edit_mask, torch.zeros_like(sigmas), sigmas)
                sigmas_fg = torch.where(edit_mask, sigmas, torch.zeros_like(sigmas))
            results['sigma'] = sigmas
            results['rgbs'] = rgbs
            results['edit_mask'] = edit_mask
            results['fg'] = self.weights_sum_i(sample_dist, sigmas_fg, None, dirs, None, z_vals, nears, fars, rgbs, prefix, masks=masks, if_fg=True)
            results['bg'] = self.weights_sum_i(sample_dist, sigmas_bg, None, dirs, None, z_vals, nears, fars, rgbs, prefix, masks=masks)
        return results

    def _run_fused(self, rays_o, rays_d, num_steps, upsample_steps, perturb, _draws=None):
        """run() on the fused kernels: 2 sampling launches + 2 field launches (+1 gather each) + 1 composite launch.
        Same result dict as run(); RNG draws keep the reference's order (rand(N,T) then rand(N,t))."""
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3).float()
        rays_d = rays_d.contiguous().view(-1, 3).float()
        N = rays_o.shape[0]
        device = rays_o.device
        draws = _draws or {}
        aabb = self.aabb_train if self.training else self.aabb_infer
        nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, aabb, self.min_near)
        S = num_steps + upsample_steps
        soft, thr = bool(getattr(self.opt, 'soft_mask', False)), float(getattr(self.opt, 'conf_thr', 0.5))
        dbg, dmask = bool(getattr(self.opt, 'detach_bg', False)), bool(getattr(self.opt, 'detach_mask_from_field', False))
        split = (num_steps == upsample_steps and getattr(self.opt, 'split_eval', True) and getattr(self, 'supports_split_eval', False)
                 and not getattr(self.opt, 'eval_fine_density', False))
        grad_on = torch.is_grad_enabled()
        plan = None
        with torch.no_grad():
            noise = None
            both = None
            if perturb and self.training and 'z' not in draws and 'u' not in draws:
                # rand(N,T) then rand(N,t) (renderer.py:317, :37) drawn by one generator launch: the first N*T values are the jitter
                both = torch.rand(N * (num_steps + upsample_steps), device=device)
            if perturb:
                noise = (draws['z'].to(device).contiguous() if 'z' in draws else
                         (both[:N * num_steps].view(N, num_steps) if both is not None else torch.rand(N, num_steps, device=device)))
            if self.training:
                u_draw = lambda: (draws['u'].to(device).contiguous() if 'u' in draws else
                                  (both[N * num_steps:].view(N, upsample_steps) if both is not None else torch.rand(N, upsample_steps, device=device)))
            else:
                u_draw = lambda: None                                         # det=True (sample_pdf :33-35)
            if split:
                # sample list = [coarse block N*T | fine block N*t]; the coarse block's grid features are gathered once (density pass) and
                # stay in `enc` for the full evaluation; the sorted order only exists as an index (src) for the compositing kernels.
                Pc, P = N * num_steps, N * S
                xyz_list = torch.empty(P, 3, dtype=torch.float32, device=device)
                enc, unit = self.split_buffers(P, device)
                bnd = float(self.opt.bound)                                  # the samplers also write the grid's [0,1] coordinates (no elementwise pass)
                z_vals, xyz_c = render_ops.sample_coarse(rays_o, rays_d, nears, fars, aabb, num_steps, noise, xyz_out=xyz_list[:Pc].view(N, num_steps, 3),
                                                         unit_out=unit[:Pc].view(N, num_steps, 3), bound=bnd)
                self.split_encode(enc, unit, xyz_list[:Pc], 0, unit_ready=True)
                sig_c = self.split_density(enc, xyz_list[:Pc])
                z_all, xyz_f, src = render_ops.sample_fine_merge_split(rays_o, rays_d, nears, fars, aabb, z_vals, sig_c, upsample_steps, u_draw(),
                                                                        xyz_fine_out=xyz_list[Pc:].view(N, upsample_steps, 3),
                                                                        unit_fine_out=unit[Pc:].view(N, upsample_steps, 3), bound=bnd)
                plan = self.split_prepare(unit, grad_on)                     # all coordinates exist: the scatter's histogram runs beside the gather below
                self.split_encode(enc, unit, xyz_list[Pc:], Pc, unit_ready=True)
            else:
                z_vals, xyzs = render_ops.sample_coarse(rays_o, rays_d, nears, fars, aabb, num_steps, noise)
                sig_c = self.density(xyzs.view(-1, 3))['sigma'].float().contiguous()
                z_all, xyz_all = render_ops.sample_fine_merge(rays_o, rays_d, nears, fars, aabb, z_vals, sig_c, upsample_steps, u_draw())
                if getattr(self.opt, 'eval_fine_density', False):
                    self.density(xyz_all.view(-1, 3))
        if split:
            # both blocks hold num_steps samples per ray, so "one direction per num_steps consecutive samples" covers the list with [d | d]
            sig_l, rgbc_l = self.split_forward(enc, unit, xyz_list, torch.cat([rays_d, rays_d], 0), num_steps, plan=plan)
            out_ray = render_ops.composite_run_indexed(sig_l, rgbc_l, z_all, src, nears, fars, num_steps, soft, thr, dbg, dmask)
            # per-sample by-products (weights, sorted-order sigma / rgbc copies, detached): a second launch, only if somebody reads them
            aux = _Lazy(lambda: render_ops.composite_run_indexed_aux(sig_l, rgbc_l, z_all, src, nears, fars, num_steps, soft, thr))
            weights_of = lambda v: _Lazy(lambda: aux.get()[0][v])
            sigma_e = _Lazy(lambda: aux.get()[1].view(N, S, 1))
            rgbs_e = _Lazy(lambda: aux.get()[2].view(N, S, 4)[..., :3])
            conf_of = lambda: aux.get()[2].view(N, S, 4)[..., 3:4]
        else:
            sigmas, rgbc, _ = self(xyz_all.view(-1, 3), rays_d, dir_group=S)
            out_ray, out_w = render_ops.composite_run(sigmas.view(N, S), rgbc.view(N, S, 4), z_all, nears, fars, num_steps, soft, thr, dbg, dmask)
            weights_of = lambda v: out_w[v]
            sigma_e, rgbs_e = sigmas.view(N, S, 1), rgbc.view(N, S, 4)[..., :3]
            conf_of = lambda: rgbc.view(N, S, 4)[..., 3:4]
        mask = _Lazy(lambda: (nears < fars).reshape(*prefix))

        def pack(v):
            r = out_ray[v]
            return _LazyResults({'image': r[:, 0:3].reshape(*prefix, 3), 'depth': r[:, 3].reshape(*prefix), 'weights_sum': r[:, 4],
                                 'render_mask': r[:, 5].reshape(*prefix, 1), 'weights': weights_of(v), 'mask': mask})
        results = pack(0)
        results['sigma'] = sigma_e
        results['rgbs'] = rgbs_e
        if getattr(self.opt, 'soft_mask', False):
            thr_ = self.opt.conf_thr
            results['edit_mask'] = _Lazy(lambda: torch.sigmoid((conf_of().detach() - thr_) * 100))
        else:
            results['edit_mask'] = _Lazy(lambda: conf_of().detach() > 0.5)
        results['z_vals'] = z_all
        results['_out_ray'] = out_ray                                        # [3, N, 6] raw composite output (trainer.ReconTrainer's fused loss)
        results['fg'] = pack(1)
        results['bg'] = pack(2)
        return results

    def weights_sum_i(self, sample_dist, sigmas, normals, dirs, weights, z_vals, nears, fars, rgbs, prefix, masks=None,
                      bg_color=None, if_fg=False, is_all=False):
        """renderer.py:407-474 (normals are None on the grid backbone)."""
        if is_all and getattr(self.opt, 'detach_bg', False):                      # :409-418
            edit_points = masks.mean(-1, keepdims=True) >= 0.5
            sigmas = torch.where(edit_points, sigmas, sigmas.detach())
            rgbs = torch.where(edit_points, rgbs, rgbs.detach())
        deltas = z_vals[..., 1:] - z_vals[..., :-1]
        deltas = torch.cat([deltas, sample_dist * torch.ones_like(deltas[..., :1])], dim=-1)
        alphas = 1 - torch.exp(-deltas * sigmas.squeeze(-1))
        alphas_shifted = torch.cat([torch.ones_like(alphas[..., :1]), 1 - alphas + 1e-15], dim=-1)
        weights = alphas * torch.cumprod(alphas_shifted, dim=-1)[..., :-1]
        results = {}
        weights_sum = weights.sum(dim=-1)
        ori_z_vals = ((z_vals - nears) / (fars - nears)).clamp(0, 1)
        depth = torch.sum(weights * ori_z_vals, dim=-1)
        image = torch.sum(weights.unsqueeze(-1) * rgbs, dim=-2)
        image = image.view(*prefix, 3)
        depth = depth.view(*prefix)
        if if_fg and bg_color is not None:
            results['black_image'] = image
            image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
        mask = (nears < fars).reshape(*prefix)
        results['image'] = image
        if getattr(self.opt, 'train_conf', 0):
            w = weights.unsqueeze(-1).detach() if getattr(self.opt, 'detach_mask_from_field', False) else weights.unsqueeze(-1)
            results['render_mask'] = torch.sum(w * masks, dim=-2).view(*prefix, -1)
        results['depth'] = depth
        results['weights_sum'] = weights_sum
        results['weights'] = weights
        results['mask'] = mask
        return results

    # ------------------------------------------------------------------------------------------ run_cuda (occupancy march)
    def run_cuda(self, rays_o, rays_d, dt_gamma=0, light_d=None, ambient_ratio=1.0, shading='albedo', bg_color=None,
                 perturb=False, force_all_rays=False, max_steps=1024, T_thresh=1e-4, _noises=None, **kwargs):
        """renderer.py:597-718."""
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.cuda().contiguous().view(-1, 3)
        rays_d = rays_d.cuda().contiguous().view(-1, 3)
        N = rays_o.shape[0]
        device = rays_o.device
        # NB: the reference does not pass min_near here, so the wrapper default 0.2 applies (renderer.py:612-613)
        nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, self.aabb_train if self.training else self.aabb_infer)
        results = {}
        normals = None

        if self.training:
            counter = self.step_counter[self.local_step % 16]
            counter.zero_()
            self.local_step += 1
            xyzs, dirs, deltas, rays = raymarching.march_rays_train(rays_o, rays_d, self.bound, self.density_bitfield, self.cascade,
                                                                    self.grid_size, nears, fars, counter, self.mean_count, perturb,
                                                                    128, force_all_rays, dt_gamma, max_steps, noises=_noises)
            sigmas, rgbs, normals = self(xyzs, dirs)
            weights_sum, depth, image = raymarching.composite_rays_train(sigmas, rgbs.float(), deltas, rays, T_thresh)
            results['rays'] = rays
            results['num_points'] = xyzs.shape[0]
        else:
            dtype = torch.float32
            weights_sum = torch.zeros(N, dtype=dtype, device=device)
            depth = torch.zeros(N, dtype=dtype, device=device)
            image = torch.zeros(N, 3, dtype=dtype, device=device)
            n_alive = N
            rays_alive = torch.arange(n_alive, dtype=torch.int32, device=device)
            rays_alive_next = torch.empty_like(rays_alive)
            count = torch.zeros(1, dtype=torch.int32, device=device)
            rays_t = nears.clone()
            step = 0
            while step < max_steps:                                               # :667-688
                if n_alive <= 0:
                    break
                n_step = max(min(N // n_alive, 8), 1)
                xyzs, dirs, deltas = raymarching.march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, self.bound,
                                                            self.density_bitfield, self.cascade, self.grid_size, nears, fars, 128,
                                                            perturb if step == 0 else False, dt_gamma, max_steps)
                sigmas, rgbs, normals = self(xyzs, dirs, light_d, ratio=ambient_ratio, shading=shading)
                raymarching.composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh)
                raymarching.compact_rays_alive(rays_alive, n_alive, rays_alive_next, count)     # device-side `[rays_alive >= 0]`
                rays_alive, rays_alive_next = rays_alive_next, rays_alive
                n_alive = int(count.item())
                step += n_step

        opt_bg = getattr(self.opt, 'bg_color', None)
        if opt_bg:
            bg = torch.tensor([np.array(opt_bg)], dtype=torch.float32, device=device)
            image = image + (1 - weights_sum).unsqueeze(-1) * bg
        image = image.view(*prefix, 3)
        depth = depth.view(*prefix)
        weights_sum = weights_sum.reshape(*prefix)
        mask = (nears < fars).reshape(*prefix)
        results['image'] = image
        results['depth'] = depth
        results['weights_sum'] = weights_sum
        results['mask'] = mask
        return results

    # ------------------------------------------------------------------------------------------ occupancy grid refresh
    @torch.no_grad()
    def update_extra_state(self, decay=0.95, S=128):
        """renderer.py:1658-1715."""
        if not self.cuda_ray:
            return
        tmp_grid = - torch.ones_like(self.density_grid)
        dev = self.density_bitfield.device
        X = torch.arange(self.grid_size, dtype=torch.int32, device=dev).split(S)
        Y = torch.arange(self.grid_size, dtype=torch.int32, device=dev).split(S)
        Z = torch.arange(self.grid_size, dtype=torch.int32, device=dev).split(S)
        for xs in X:
            for ys in Y:
                for zs in Z:
                    xx, yy, zz = custom_meshgrid(xs, ys, zs)
                    coords = torch.cat([xx.reshape(-1, 1), yy.reshape(-1, 1), zz.reshape(-1, 1)], dim=-1)
                    indices = raymarching.morton3D(coords).long()
                    xyzs = 2 * coords.float() / (self.grid_size - 1) - 1
                    for cas in range(self.cascade):
                        bound = min(2 ** cas, self.bound)
                        half_grid_size = bound / self.grid_size
                        cas_xyzs = xyzs * (bound - half_grid_size)
                        cas_xyzs += (torch.rand_like(cas_xyzs) * 2 - 1) * half_grid_size
                        sigmas = self.density(cas_xyzs)['sigma'].reshape(-1).detach()
                        tmp_grid[cas, indices] = sigmas.float()
        valid_mask = self.density_grid >= 0
        self.density_grid[valid_mask] = torch.maximum(self.density_grid[valid_mask] * decay, tmp_grid[valid_mask])
        self.mean_density = torch.mean(self.density_grid[valid_mask]).item()
        self.iter_density += 1
        density_thresh = min(self.mean_density, self.density_thresh)
        self.density_bitfield = raymarching.packbits(self.density_grid, density_thresh, self.density_bitfield)
        total_step = min(16, self.local_step)
        if total_step > 0:
            self.mean_count = int(self.step_counter[:total_step, 0].sum().item() / total_step)
        self.local_step = 0

    def render(self, rays_o, rays_d, staged=False, max_ray_batch=2048, **kwargs):
        """renderer.py:1719-1733."""
        _run = self.run_cuda if self.cuda_ray else self.run
        return _run(rays_o, rays_d, **kwargs)
