"""NeRFRenderer — counterpart of the hot-path part of the reference's nerf/renderer.py.

Implemented (reference lines): sample_pdf (21-55), NeRFRenderer.__init__ (199-243), reset_extra_state (266-276),
run (278-405), weights_sum_i (407-474), run_cuda (597-718), update_extra_state (1658-1715), render (1719-1733).
Out of scope: the SDF/NeuS paths, mesh export, the unused run_cuda2 / render_cuda duplicates.
Every function here ends in HIP kernels (render.hip, raymarching.hip, occupancy.hip): there is no torch restatement of the
reference's bodies in the product — that lives in oracle/torch_oracle.py, for the tests.

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
from .._lib import lib, check, ptr, stream, require_cuda
from . import render_ops


def sample_pdf(bins, weights, n_samples, det=False, u=None):
    """The reference's `sample_pdf(bins, weights, n_samples, det)` (renderer.py:21-55) on the HIP kernel k_sample_pdf: bins [B, T],
    weights [B, T-1] -> [B, n_samples].  `u` [B, n_samples] replays the random draw of :37 (drawn here with torch.rand when training).
    run() does not come through here — its resampling is fused with the merge (render_ops.sample_fine_merge)."""
    require_cuda(bins, weights, u)
    bins, weights = bins.contiguous().float(), weights.contiguous().float()
    B, T = bins.shape
    if weights.shape != (B, T - 1):
        raise ValueError(f"sample_pdf: weights must be [B, {T - 1}] for bins [B, {T}]")
    if not det and u is None:
        u = torch.rand(B, n_samples, device=bins.device)
    out = torch.empty(B, n_samples, dtype=torch.float32, device=bins.device)
    check(lib.cnerf_sample_pdf(ptr(bins), ptr(weights), None if det else ptr(u.contiguous().float()), B, T, int(n_samples), ptr(out), stream()), "sample_pdf")
    return out


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

    # ------------------------------------------------------------------------------------------ run (the path `-O2` takes)
    def run(self, rays_o, rays_d, num_steps=128, upsample_steps=128, light_d=None, ambient_ratio=1.0, shading='albedo',
            bg_color=None, perturb=False, _draws=None, **kwargs):
        """renderer.py:278-405.  rays_o, rays_d [B,N,3] (B == 1) -> result dict, on the fused HIP kernels (no torch fallback: CPU
        tensors raise).  `_draws` = dict(light, z, u) replays the RNG draws of :305, :317 and sample_pdf:37 (tests).
        The reference fills its result dict only under `opt.train_conf` (:383-405): without it the dict is empty here too."""
        require_cuda(rays_o, rays_d)
        if not getattr(self.opt, 'train_conf', 0):
            return {}
        if not (3 <= num_steps <= 128 and 2 <= upsample_steps <= 128):
            # (upsample_steps == 0 fails in the reference as well: `weights` is bound only inside `if upsample_steps > 0`, renderer.py:333-384)
            raise ValueError(f"run(): the sampling kernels take 3..128 coarse and 2..128 importance samples per ray (got {num_steps} + {upsample_steps}; "
                             "the reference's recipe is 64 + 64)")
        return self._run_fused(rays_o, rays_d, num_steps, upsample_steps, perturb, _draws, fg_bg=bool(kwargs.get('fg_bg', True)))

    def _dirs_twice(self, rays_d):
        """[d | d] for the split sample list, cached per view: a dataset's ray directions are resident tensors that come back every epoch
        (provider.py keeps them on the GPU), so the concatenation launch is paid once per view, not once per step"""
        if torch.cuda.is_current_stream_capturing():
            # hipGraph capture (trainer.train_step_graphed): the concatenation becomes a node of the graph and its result lives in the graph's
            # own memory pool — a cached tensor's address would be baked into the graph and outlive its eviction from the 64-entry cache
            return torch.cat([rays_d, rays_d], 0)
        cache = self.__dict__.setdefault('_dirs2_cache', {})
        key = (rays_d.data_ptr(), rays_d._version, tuple(rays_d.shape))
        ent = cache.get(key)
        if ent is None:
            if len(cache) >= 64:
                cache.pop(next(iter(cache)))
            ent = (rays_d, torch.cat([rays_d, rays_d], 0))               # the source reference pins the key's storage
            cache[key] = ent
        return ent[1]

    def _field_all(self, xyz, rays_d, S):
        """sigma [P], rgb+confidence [P, 4] of P = N*S samples (S consecutive samples share a ray direction) from a subclass forward()"""
        if getattr(self, 'supports_dir_group', False):
            sigmas, rgbc, _ = self(xyz, rays_d, dir_group=S)
        else:
            sigmas, rgbc, _ = self(xyz, rays_d.repeat_interleave(S, dim=0))
        if rgbc.shape[-1] != 4:
            raise ValueError("run() with opt.train_conf needs forward() to return rgb + 1 confidence channel (network_grid.py:126-129)")
        return sigmas.reshape(-1), rgbc.reshape(-1, 4)

    def _run_fused(self, rays_o, rays_d, num_steps, upsample_steps, perturb, _draws=None, fg_bg=True):
        """run() on the fused kernels: 2 sampling launches + 2 field launches (+1 gather each) + 1 composite launch.
        Same result dict as run(); RNG draws keep the reference's order (rand(N,T) then rand(N,t)).
        fg_bg=False (the reconstruction trainer, whose loss reads the first composite only): the edit-region / background composites are not
        computed and the result dict has no 'fg' / 'bg' entries."""
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3).float()
        rays_d = rays_d.contiguous().view(-1, 3).float()
        N = rays_o.shape[0]
        device = rays_o.device
        draws = _draws or {}
        aabb = self.aabb_train if self.training else self.aabb_infer
        S = num_steps + upsample_steps
        soft, thr = bool(getattr(self.opt, 'soft_mask', False)), float(getattr(self.opt, 'conf_thr', 0.5))
        dbg, dmask = bool(getattr(self.opt, 'detach_bg', False)), bool(getattr(self.opt, 'detach_mask_from_field', False))
        split = (num_steps == upsample_steps and getattr(self.opt, 'split_eval', True) and getattr(self, 'supports_split_eval', False)
                 and not getattr(self.opt, 'eval_fine_density', False))
        grad_on = torch.is_grad_enabled()
        plan = None
        if not split:
            nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, aabb, self.min_near)
        with torch.no_grad():
            noise = None
            both = None
            blockwise = False
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
                # near_far_from_aabb (:297) rides on the coarse sampler's launch
                nears, fars, z_vals, xyz_c = render_ops.sample_coarse_aabb(rays_o, rays_d, aabb.contiguous().float(), self.min_near, num_steps, noise,
                                                                           xyz_list[:Pc].view(N, num_steps, 3), unit[:Pc].view(N, num_steps, 3), bnd)
                # the scatter plan's histogram in two pieces: the coarse block's rows now (beside the coarse gather), the fine block's after the
                # importance sampling — all of it is then done before the field backward starts
                pstate, piecewise = self.split_prepare_rows(None, unit, grad_on, 0, Pc, False)
                self.split_encode(enc, unit, xyz_list[:Pc], 0, unit_ready=True)
                blockwise = bool(getattr(self.opt, 'blockwise_field', True)) and self._fused_cfg()[2] == 4
                if blockwise:
                    # full evaluation of the coarse block right away: its sigma drives the importance sampling and its colours are the
                    # coarse half of the final evaluation — no density-only pass, no second visit of these rows
                    sig_all, rgbc_all = self.split_outputs(P, device)
                    self.split_forward_rows(enc, 0, xyz_list[:Pc], rays_d, num_steps, sig_all, rgbc_all)
                    sig_c = sig_all[:Pc]
                else:
                    sig_c = self.split_density(enc, xyz_list[:Pc])
                z_all, xyz_f, src = render_ops.sample_fine_merge_split(rays_o, rays_d, nears, fars, aabb, z_vals, sig_c, upsample_steps, u_draw(),
                                                                        xyz_fine_out=xyz_list[Pc:].view(N, upsample_steps, 3),
                                                                        unit_fine_out=unit[Pc:].view(N, upsample_steps, 3), bound=bnd)
                if piecewise and pstate is not None:
                    plan, _ = self.split_prepare_rows(pstate, unit, grad_on, Pc, P - Pc, True)
                elif not piecewise:
                    plan = self.split_prepare(unit, grad_on)                 # all coordinates exist: the scatter's histogram runs beside the gather below
                self.split_encode(enc, unit, xyz_list[Pc:], Pc, unit_ready=True, importance=True)
                if blockwise:
                    self.split_forward_rows(enc, Pc, xyz_list[Pc:], rays_d, upsample_steps, sig_all, rgbc_all)
            else:
                z_vals, xyzs = render_ops.sample_coarse(rays_o, rays_d, nears, fars, aabb, num_steps, noise)
                sig_c = self.density(xyzs.view(-1, 3))['sigma'].float().contiguous()
                z_all, xyz_all = render_ops.sample_fine_merge(rays_o, rays_d, nears, fars, aabb, z_vals, sig_c, upsample_steps, u_draw())
                if getattr(self.opt, 'eval_fine_density', False):
                    self.density(xyz_all.view(-1, 3))
        if split:
            # both blocks hold num_steps samples per ray, so "one direction per num_steps consecutive samples" covers the list with [d | d]
            dirs2 = self._dirs_twice(rays_d)
            if blockwise:
                sig_l, rgbc_l = self.split_attach(enc, unit, xyz_list, dirs2, num_steps, sig_all, rgbc_all, plan=plan)
            else:
                sig_l, rgbc_l = self.split_forward(enc, unit, xyz_list, dirs2, num_steps, plan=plan)
            # early termination of the backward (north_star; the reference's own is `T < T_thresh` on its march path): only where the gradients
            # go into the half-precision fused field, whose backward rounds them to half anyway
            flush = bool(grad_on and getattr(self.opt, 'early_termination', True) and self._fused_cfg() and self._half())
            out_ray = render_ops.composite_run_indexed(sig_l, rgbc_l, z_all, src, nears, fars, num_steps, soft, thr, dbg, dmask, flush_half_zero=flush,
                                                       variants=7 if fg_bg else 1)
            # per-sample by-products (weights, sorted-order sigma / rgbc copies, detached): a second launch, only if somebody reads them
            aux = _Lazy(lambda: render_ops.composite_run_indexed_aux(sig_l, rgbc_l, z_all, src, nears, fars, num_steps, soft, thr))
            weights_of = lambda v: _Lazy(lambda: aux.get()[0][v])
            sigma_e = _Lazy(lambda: aux.get()[1].view(N, S, 1))
            rgbs_e = _Lazy(lambda: aux.get()[2].view(N, S, 4)[..., :3])
            conf_of = lambda: aux.get()[2].view(N, S, 4)[..., 3:4]
        else:
            sigmas, rgbc = self._field_all(xyz_all.view(-1, 3), rays_d, S)
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
        if fg_bg or not split:                                               # (fg_bg=False: the two composites were not computed — no such keys)
            results['fg'] = pack(1)
            results['bg'] = pack(2)
        return results

    def weights_sum_i(self, sample_dist, sigmas, normals, dirs, weights, z_vals, nears, fars, rgbs, prefix, masks=None,
                      bg_color=None, if_fg=False, is_all=False, num_steps=None):
        """One alpha composite with the reference's signature (renderer.py:407-474) on the compositing kernel of run() (its "all" variant
        takes sigma as given): sigmas [N,S(,1)], rgbs [N,S,3], masks [N,S,1] | None, z_vals [N,S], nears / fars / sample_dist [N,1].
        Differentiable in sigmas / rgbs / masks.  `num_steps` = (far - near) / sample_dist, the kernel's form of the last interval; when it is
        not passed it is recovered from the first ray that hits the box (one host read; rays with near >= far carry no interval)."""
        N, S = z_vals.shape
        nears, fars = nears.reshape(N).contiguous().float(), fars.reshape(N).contiguous().float()
        if num_steps is None:
            sd_ = sample_dist.reshape(-1).float().expand(N) if sample_dist.numel() in (1, N) else sample_dist.reshape(-1).float()[:N]
            ratio = torch.where((fars > nears) & (sd_ > 0), (fars - nears) / sd_.clamp_min(1e-30), torch.full_like(nears, -1.0))
            num_steps = int(torch.round(ratio.max()).item())         # every valid ray has the same (far - near) / sample_dist = the sample count
            if num_steps <= 0:
                raise ValueError("weights_sum_i: no ray intersects the box (near >= far everywhere); pass num_steps")
        conf = masks.reshape(N, S, 1).float() if masks is not None else torch.zeros(N, S, 1, device=z_vals.device)
        rgbc = torch.cat([rgbs.reshape(N, S, 3).float(), conf], dim=-1)
        detach_bg = bool(is_all and getattr(self.opt, 'detach_bg', False))                           # :409-418
        out_ray, out_w = render_ops.composite_run(sigmas.reshape(N, S).float(), rgbc, z_vals.contiguous().float(), nears, fars, num_steps,
                                                  True, float(getattr(self.opt, 'conf_thr', 0.5)), detach_bg,
                                                  bool(getattr(self.opt, 'detach_mask_from_field', False)))
        r = out_ray[0]
        results = {}
        image = r[:, 0:3].reshape(*prefix, 3)
        if if_fg and bg_color is not None:                                                            # :451-453
            results['black_image'] = image
            image = image + (1 - r[:, 4]).unsqueeze(-1).reshape(*prefix, 1) * bg_color
        results['image'] = image
        if getattr(self.opt, 'train_conf', 0) and masks is not None:
            results['render_mask'] = r[:, 5].reshape(*prefix, 1)
        results['depth'] = r[:, 3].reshape(*prefix)
        results['weights_sum'] = r[:, 4]
        results['weights'] = out_w[0]
        results['mask'] = (nears < fars).reshape(*prefix)
        return results

    # ------------------------------------------------------------------------------------------ run_cuda (occupancy march)
    def run_cuda(self, rays_o, rays_d, dt_gamma=0, light_d=None, ambient_ratio=1.0, shading='albedo', bg_color=None,
                 perturb=False, force_all_rays=False, max_steps=1024, T_thresh=1e-4, _noises=None, **kwargs):
        """renderer.py:597-718: occupancy-grid march.  Training: one compacted sample list per batch (march_rays_train ->
        field -> composite_rays_train); inference: the alive-ray loop of :661-688 with device-side compaction."""
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.cuda().contiguous().view(-1, 3)
        rays_d = rays_d.cuda().contiguous().view(-1, 3)
        aabb = self.aabb_train if self.training else self.aabb_infer
        # the reference passes no min_near here: the wrapper's default 0.2 applies (:612-613)
        nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, aabb)
        results = {}
        if self.training:
            weights_sum, depth, image = self._march_train(rays_o, rays_d, nears, fars, dt_gamma, perturb, force_all_rays, max_steps, T_thresh, _noises, results)
        else:
            weights_sum, depth, image = self._march_infer(rays_o, rays_d, nears, fars, dt_gamma, perturb, max_steps, T_thresh, light_d, ambient_ratio, shading)
        bg = getattr(self.opt, 'bg_color', None)                    # read by the reference (:694) but never defined by its argparse
        if bg:
            bg = torch.as_tensor(np.asarray(bg, dtype=np.float32), device=image.device).reshape(1, -1)
            image = image + (1 - weights_sum).unsqueeze(-1) * bg
        results.update(image=image.view(*prefix, 3), depth=depth.view(*prefix), weights_sum=weights_sum.reshape(*prefix),
                       mask=(nears < fars).reshape(*prefix))
        return results

    def _march_train(self, rays_o, rays_d, nears, fars, dt_gamma, perturb, force_all_rays, max_steps, T_thresh, noises, results):
        """:617-635.  The per-call sample counter is one slot of the 16-entry ring `step_counter` (mean_count is their average, update_extra_state)."""
        counter = self.step_counter[self.local_step % 16]
        counter.zero_()
        self.local_step += 1
        xyzs, dirs, deltas, rays = raymarching.march_rays_train(rays_o, rays_d, self.bound, self.density_bitfield, self.cascade, self.grid_size,
                                                                nears, fars, counter, self.mean_count, perturb, 128, force_all_rays, dt_gamma,
                                                                max_steps, noises=noises)
        sigmas, rgbs, _ = self(xyzs, dirs)
        results['rays'] = rays
        results['num_points'] = xyzs.shape[0]
        return raymarching.composite_rays_train(sigmas, rgbs.float(), deltas, rays, T_thresh)

    def _march_infer(self, rays_o, rays_d, nears, fars, dt_gamma, perturb, max_steps, T_thresh, light_d, ambient_ratio, shading):
        """:637-688.  Every round advances each alive ray by n_step occupied samples, accumulates in place and compacts the alive list on the
        device (`rays_alive[rays_alive >= 0]` of :684 as a ballot + scan kernel); the only host read per round is the alive count that sizes
        the next round's launches."""
        N, device = rays_o.shape[0], rays_o.device
        acc = torch.zeros(5 * N, dtype=torch.float32, device=device)                # weights_sum | depth | rgb: one zero-fill, three views
        weights_sum, depth, image = acc[:N], acc[N:2 * N], acc[2 * N:].view(N, 3)
        alive = torch.arange(N, dtype=torch.int32, device=device)
        alive_next = torch.empty_like(alive)
        count = torch.zeros(1, dtype=torch.int32, device=device)
        rays_t = nears.clone()
        n_alive, done = N, 0
        while done < max_steps and n_alive > 0:
            n_step = max(min(N // n_alive, 8), 1)                                   # :671
            xyzs, dirs, deltas = raymarching.march_rays(n_alive, n_step, alive, rays_t, rays_o, rays_d, self.bound, self.density_bitfield,
                                                        self.cascade, self.grid_size, nears, fars, 128, perturb and done == 0, dt_gamma, max_steps)
            sigmas, rgbs, _ = self(xyzs, dirs, light_d, ratio=ambient_ratio, shading=shading)
            raymarching.composite_rays(n_alive, n_step, alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh)
            raymarching.compact_rays_alive(alive, n_alive, alive_next, count)
            alive, alive_next = alive_next, alive
            n_alive = int(count.item())
            done += n_step
        return weights_sum, depth, image

    # ------------------------------------------------------------------------------------------ occupancy grid refresh
    @property
    def mean_density(self):
        """mean of the valid cells after the last refresh (renderer.py:1710); kept on the device by update_extra_state, read on demand"""
        md = self.__dict__.get('_mean_density', 0)
        return float(md[0]) if torch.is_tensor(md) else md

    @mean_density.setter
    def mean_density(self, v):
        self.__dict__['_mean_density'] = v

    @torch.no_grad()
    def update_extra_state(self, decay=0.95, S=128, _rand=None):
        """renderer.py:1658-1715 as kernels (csrc/occupancy.hip): per cascade, jittered cell-centre queries -> density -> EMA-max into
        `density_grid` (Morton order); then mean density, threshold min(mean, density_thresh) and packbits in one go, all on the device.
        `S` (the reference's chunk size) is accepted and ignored: a cascade's 128^3 queries are one batch.  `_rand` = list of [H^3, 3] U[0,1)
        tensors, one per cascade, replaying the `torch.rand_like` draws of :1695 (tests).  The only host read is the 16-slot sample-count
        ring that sizes the next march_rays_train calls (:1714-1716)."""
        if not self.cuda_ray:
            return
        H, dev = self.grid_size, self.density_bitfield.device
        n = H ** 3
        nblk = (n + 255) // 256
        partials = torch.empty(self.cascade * nblk, 2, dtype=torch.float64, device=dev)
        xyzs = torch.empty(n, 3, dtype=torch.float32, device=dev)
        for cas in range(self.cascade):
            bound = min(2 ** cas, self.bound)
            half = bound / H
            rnd = (_rand[cas].to(dev) if _rand is not None else torch.rand(n, 3, device=dev)).contiguous().float()
            check(lib.cnerf_occupancy_points(ptr(rnd), H, float(bound), float(half), ptr(xyzs), stream()), "occupancy_points")
            sigmas = self.density(xyzs)['sigma'].reshape(-1).float().contiguous()
            check(lib.cnerf_occupancy_update(ptr(sigmas), H, float(decay), ptr(self.density_grid[cas]), ptr(partials[cas * nblk:]), stream()), "occupancy_update")
        state = torch.empty(2, dtype=torch.float32, device=dev)
        check(lib.cnerf_occupancy_finalize_pack(ptr(partials), self.cascade * nblk, float(self.density_thresh), ptr(self.density_grid),
                                                self.density_bitfield.numel(), ptr(state), ptr(self.density_bitfield), stream()), "occupancy_finalize_pack")
        self.mean_density = state
        self.iter_density += 1
        total_step = min(16, self.local_step)
        if total_step > 0:
            self.mean_count = self._mean_sample_count(total_step)
        self.local_step = 0

    def _mean_sample_count(self, total_step):
        """average samples per march_rays_train call over the last `total_step` calls (:1714-1716); data-parallel ranks average it too, so
        that every rank sizes its sample budget alike.  That makes update_extra_state a COLLECTIVE under torch.distributed (every rank must
        call it); `opt.sync_mean_count = False` keeps it rank-local (e.g. a rank-0-only evaluation that refreshes the occupancy grid)."""
        tot = self.step_counter[:total_step, 0].sum().float()
        import torch.distributed as dist
        if getattr(self.opt, 'sync_mean_count', True) and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from .. import _coll
            _coll.all_reduce(tot)
            tot = tot / dist.get_world_size()
        return int(tot.item() / total_step)

    def render(self, rays_o, rays_d, staged=False, max_ray_batch=2048, **kwargs):
        """renderer.py:1719-1733."""
        _run = self.run_cuda if self.cuda_ray else self.run
        return _run(rays_o, rays_d, **kwargs)
