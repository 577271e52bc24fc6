"""Python entries of the fused `run()` kernels (customnerf_amd/csrc/render.hip): stratified sampling, importance resampling +
merge, and the three weights_sum_i composites with their backward.  Semantics: nerf/renderer.py:310-363, 384-474, 21-55."""
import torch
from torch.autograd import Function

from .._lib import lib, check, ptr, stream, require_cuda


def sample_coarse(rays_o, rays_d, nears, fars, aabb, T, noise=None, xyz_out=None, unit_out=None, bound=None):
    """-> z_vals [N,T], xyzs [N,T,3]   (renderer.py:310-322; noise [N,T] = the torch.rand draw of :317 or None).
    xyz_out: optional contiguous [N,T,3] float32 destination (e.g. the head of a larger sample list).
    unit_out (+ bound): also write the grid's [0,1] coordinates (xyz + bound) / (2 bound) of the samples there (grid.py:156)."""
    require_cuda(rays_o, rays_d, nears, fars, aabb, noise, xyz_out, unit_out)
    N = rays_o.shape[0]
    z = torch.empty(N, T, dtype=torch.float32, device=rays_o.device)
    xyz = xyz_out if xyz_out is not None else torch.empty(N, T, 3, dtype=torch.float32, device=rays_o.device)
    assert xyz.is_contiguous() and xyz.dtype == torch.float32 and xyz.numel() == N * T * 3
    if unit_out is not None:
        assert unit_out.is_contiguous() and unit_out.dtype == torch.float32 and unit_out.numel() == N * T * 3 and bound is not None
        check(lib.cnerf_sample_coarse_unit(ptr(rays_o), ptr(rays_d), ptr(nears), ptr(fars), ptr(aabb), ptr(noise), N, int(T), ptr(z), ptr(xyz),
                                           ptr(unit_out), float(bound), stream()), "sample_coarse_unit")
        return z, xyz
    check(lib.cnerf_sample_coarse(ptr(rays_o), ptr(rays_d), ptr(nears), ptr(fars), ptr(aabb), ptr(noise), N, int(T), ptr(z), ptr(xyz), stream()),
          "sample_coarse")
    return z, xyz


def sample_coarse_aabb(rays_o, rays_d, aabb, min_near, T, noise, xyz_out, unit_out, bound):
    """near_far_from_aabb (renderer.py:297) + the stratified coarse samples (:310-322) + the grid's [0,1] coordinates in ONE launch
    -> nears [N], fars [N], z_vals [N,T], xyzs [N,T,3]; nears / fars are bit-identical to raymarching.near_far_from_aabb."""
    require_cuda(rays_o, rays_d, aabb, noise, xyz_out, unit_out)
    N = rays_o.shape[0]
    nears = torch.empty(N, dtype=torch.float32, device=rays_o.device)
    fars = torch.empty(N, dtype=torch.float32, device=rays_o.device)
    z = torch.empty(N, T, dtype=torch.float32, device=rays_o.device)
    assert xyz_out.is_contiguous() and xyz_out.dtype == torch.float32 and xyz_out.numel() == N * T * 3
    assert unit_out.is_contiguous() and unit_out.dtype == torch.float32 and unit_out.numel() == N * T * 3
    check(lib.cnerf_sample_coarse_unit_aabb(ptr(rays_o), ptr(rays_d), ptr(aabb), float(min_near), ptr(noise), N, int(T), ptr(nears), ptr(fars), ptr(z),
                                            ptr(xyz_out), ptr(unit_out), float(bound), stream()), "sample_coarse_unit_aabb")
    return nears, fars, z, xyz_out


def sample_fine_merge(rays_o, rays_d, nears, fars, aabb, z_vals, sigmas, t, u=None):
    """-> z_all [N,T+t] (sorted), xyz_all [N,T+t,3]   (renderer.py:334-363; u [N,t] = the draw of sample_pdf:37, None = det)"""
    require_cuda(z_vals, sigmas, u)
    N, T = z_vals.shape
    z_all = torch.empty(N, T + t, dtype=torch.float32, device=z_vals.device)
    xyz_all = torch.empty(N, T + t, 3, dtype=torch.float32, device=z_vals.device)
    check(lib.cnerf_sample_fine_merge(ptr(rays_o), ptr(rays_d), ptr(nears), ptr(fars), ptr(aabb), ptr(z_vals), ptr(sigmas), ptr(u), N, int(T), int(t),
                                      ptr(z_all), ptr(xyz_all), stream()), "sample_fine_merge")
    return z_all, xyz_all


def sample_fine_merge_split(rays_o, rays_d, nears, fars, aabb, z_vals, sigmas, t, u=None, xyz_fine_out=None, unit_fine_out=None, bound=None):
    """Split form: -> z_all [N,T+t] (sorted), xyz_fine [N,t,3] (the new samples in their own block), src_index [N,T+t] int32 = row of
    every sorted position in the sample list [coarse N*T rows | fine N*t rows]."""
    require_cuda(z_vals, sigmas, u, xyz_fine_out)
    N, T = z_vals.shape
    z_all = torch.empty(N, T + t, dtype=torch.float32, device=z_vals.device)
    xyz_fine = xyz_fine_out if xyz_fine_out is not None else torch.empty(N, t, 3, dtype=torch.float32, device=z_vals.device)
    assert xyz_fine.is_contiguous() and xyz_fine.dtype == torch.float32 and xyz_fine.numel() == N * t * 3
    src = torch.empty(N, T + t, dtype=torch.int32, device=z_vals.device)
    if unit_fine_out is not None:                        # also the new samples' [0,1] grid coordinates (see sample_coarse)
        require_cuda(unit_fine_out)
        assert unit_fine_out.is_contiguous() and unit_fine_out.dtype == torch.float32 and unit_fine_out.numel() == N * t * 3 and bound is not None
        check(lib.cnerf_sample_fine_merge_split_unit(ptr(rays_o), ptr(rays_d), ptr(nears), ptr(fars), ptr(aabb), ptr(z_vals), ptr(sigmas), ptr(u), N, int(T),
                                                     int(t), ptr(z_all), ptr(xyz_fine), ptr(src), ptr(unit_fine_out), float(bound), stream()),
              "sample_fine_merge_split_unit")
        return z_all, xyz_fine, src
    check(lib.cnerf_sample_fine_merge_split(ptr(rays_o), ptr(rays_d), ptr(nears), ptr(fars), ptr(aabb), ptr(z_vals), ptr(sigmas), ptr(u), N, int(T), int(t),
                                            ptr(z_all), None, ptr(xyz_fine), ptr(src), stream()), "sample_fine_merge_split")
    return z_all, xyz_fine, src


class _CompositeRunIndexed(Function):
    """_CompositeRun reading its samples through src_index (sample_fine_merge_split).  The per-sample by-products of the launch (weights of
    the three composites, sorted-order sigma / rgbc copies: 67 MB of writes at 128x128x128) are NOT produced here: the backward recomputes
    what it needs from sigma / rgbc, and the result dict asks for them lazily (composite_run_indexed_aux)."""

    @staticmethod
    def forward(ctx, sigmas, rgbc, z_vals, src_index, nears, fars, num_steps, soft_mask, conf_thr, detach_bg, detach_mask, flush_half_zero=False, variants=7):
        sigmas = sigmas.contiguous().float()
        rgbc = rgbc.contiguous().float()
        N, S = z_vals.shape
        ctx.flush = bool(flush_half_zero)
        out_ray = torch.empty(3, N, 6, dtype=torch.float32, device=z_vals.device)
        check(lib.cnerf_composite_run_indexed_variants(ptr(sigmas), ptr(rgbc), ptr(z_vals), ptr(nears), ptr(fars), N, S, int(num_steps), int(soft_mask),
                                                       float(conf_thr), ptr(src_index), ptr(out_ray), None, None, None, int(variants), stream()),
              "composite_run_indexed")
        ctx.save_for_backward(sigmas, rgbc, z_vals, src_index, nears, fars)
        ctx.cfg = (int(num_steps), int(soft_mask), float(conf_thr), int(detach_bg), int(detach_mask))
        return out_ray

    @staticmethod
    def backward(ctx, g_ray):
        if g_ray is None:
            return (None,) * 13
        sigmas, rgbc, z_vals, src_index, nears, fars = ctx.saved_tensors
        num_steps, soft, thr, dbg, dmask = ctx.cfg
        N, S = z_vals.shape
        g_ray = g_ray.contiguous().float()
        g_sigma = torch.empty_like(sigmas)                      # src_index is a permutation of the rows: every element is written
        g_rgbc = torch.empty_like(rgbc)
        if ctx.flush:
            # early termination for the half-precision fused field: rows whose gradients round to zero in the form its backward consumes them are
            # written as exact zeros (bit-identical parameter gradients), and — when every ray owns whole 32-row tiles of the sample list — one
            # byte per tile says whether anything is left in it; the flags travel on the gradient tensor (field.FieldFunction.backward)
            tiles = (num_steps % 32 == 0 and (S - num_steps) % 32 == 0 and 0 < num_steps < S and S <= 256 and g_sigma.numel() == N * S)
            tile_live = torch.empty(N * S // 32, dtype=torch.uint8, device=g_sigma.device) if tiles else None
            check(lib.cnerf_composite_run_backward_indexed_flush(ptr(g_ray), ptr(sigmas), ptr(rgbc), ptr(z_vals), ptr(nears), ptr(fars), N, S, num_steps, soft,
                                                                 thr, dbg, dmask, ptr(src_index), ptr(g_sigma), ptr(g_rgbc), 1, ptr(tile_live), stream()),
                  "composite_run_backward_indexed_flush")
            if tile_live is not None:
                # the flags describe THESE two tensors as written here: identity (address) and version counter of both travel with them, so the
                # field backward can tell a gradient that autograd accumulated into (in place: same object, version bumped) or replaced
                g_sigma._cnerf_tile_live = (tile_live, g_sigma.data_ptr(), g_sigma._version, g_rgbc.data_ptr(), g_rgbc._version)
        else:
            check(lib.cnerf_composite_run_backward_indexed(ptr(g_ray), ptr(sigmas), ptr(rgbc), ptr(z_vals), ptr(nears), ptr(fars), N, S, num_steps, soft, thr,
                                                           dbg, dmask, ptr(src_index), ptr(g_sigma), ptr(g_rgbc), stream()), "composite_run_backward_indexed")
        return g_sigma, g_rgbc, None, None, None, None, None, None, None, None, None, None, None


def composite_run_indexed(sigmas, rgbc, z_vals, src_index, nears, fars, num_steps, soft_mask, conf_thr, detach_bg=False, detach_mask=False,
                          flush_half_zero=False, variants=7):
    """sigmas [P], rgbc [P,4] in sample-list order -> out_ray [3,N,6] (differentiable in sigmas / rgbc).
    flush_half_zero: the producer of sigmas / rgbc is the half-precision fused field (cnerf_composite_run_backward_indexed_flush).
    variants: bit mask of the composites to compute (1 all, 2 edit region, 4 background); the rows of the others are zeros."""
    return _CompositeRunIndexed.apply(sigmas, rgbc, z_vals, src_index, nears, fars, num_steps, soft_mask, conf_thr, detach_bg, detach_mask, flush_half_zero,
                                      variants)


@torch.no_grad()
def composite_run_indexed_aux(sigmas, rgbc, z_vals, src_index, nears, fars, num_steps, soft_mask, conf_thr):
    """The detached per-sample by-products of composite_run_indexed, by a second launch of the same kernel:
    weights [3,N,S], sigma_sorted [N,S], rgbc_sorted [N,S,4]."""
    sigmas = sigmas.detach().contiguous().float()
    rgbc = rgbc.detach().contiguous().float()
    N, S = z_vals.shape
    dev = z_vals.device
    out_ray = torch.empty(3, N, 6, dtype=torch.float32, device=dev)
    out_w = torch.empty(3, N, S, dtype=torch.float32, device=dev)
    sig_s = torch.empty(N, S, dtype=torch.float32, device=dev)
    rgbc_s = torch.empty(N, S, 4, dtype=torch.float32, device=dev)
    check(lib.cnerf_composite_run_indexed(ptr(sigmas), ptr(rgbc), ptr(z_vals), ptr(nears), ptr(fars), N, S, int(num_steps), int(soft_mask), float(conf_thr),
                                          ptr(src_index), ptr(out_ray), ptr(out_w), ptr(sig_s), ptr(rgbc_s), stream()), "composite_run_indexed")
    return out_w, sig_s, rgbc_s


class _CompositeRun(Function):
    """out_ray [3,N,6] (all/fg/bg x image rgb, depth, weights_sum, render_mask), weights [3,N,S]"""

    @staticmethod
    def forward(ctx, sigmas, rgbc, z_vals, nears, fars, num_steps, soft_mask, conf_thr, detach_bg, detach_mask):
        sigmas = sigmas.contiguous().float()
        rgbc = rgbc.contiguous().float()
        N, S = z_vals.shape
        out_ray = torch.empty(3, N, 6, dtype=torch.float32, device=z_vals.device)
        out_w = torch.empty(3, N, S, dtype=torch.float32, device=z_vals.device)
        check(lib.cnerf_composite_run(ptr(sigmas), ptr(rgbc), ptr(z_vals), ptr(nears), ptr(fars), N, S, int(num_steps), int(soft_mask), float(conf_thr),
                                      ptr(out_ray), ptr(out_w), stream()), "composite_run")
        ctx.save_for_backward(sigmas, rgbc, z_vals, nears, fars)
        ctx.cfg = (int(num_steps), int(soft_mask), float(conf_thr), int(detach_bg), int(detach_mask))
        ctx.mark_non_differentiable(out_w)
        ctx.set_materialize_grads(False)
        return out_ray, out_w

    @staticmethod
    def backward(ctx, g_ray, g_w):
        if g_ray is None:
            return (None,) * 10
        sigmas, rgbc, z_vals, nears, fars = ctx.saved_tensors
        num_steps, soft, thr, dbg, dmask = ctx.cfg
        N, S = z_vals.shape
        g_ray = g_ray.contiguous().float()
        g_sigma = torch.empty_like(sigmas)
        g_rgbc = torch.empty_like(rgbc)
        check(lib.cnerf_composite_run_backward(ptr(g_ray), ptr(sigmas), ptr(rgbc), ptr(z_vals), ptr(nears), ptr(fars), N, S, num_steps, soft, thr, dbg,
                                               dmask, ptr(g_sigma), ptr(g_rgbc), stream()), "composite_run_backward")
        return g_sigma, g_rgbc, None, None, None, None, None, None, None, None


def composite_run(sigmas, rgbc, z_vals, nears, fars, num_steps, soft_mask, conf_thr, detach_bg=False, detach_mask=False):
    return _CompositeRun.apply(sigmas, rgbc, z_vals, nears, fars, num_steps, soft_mask, conf_thr, detach_bg, detach_mask)


class _ReconLoss(Function):
    """loss [] = w_rgb * mse(image, rgb_gt) + w_conf * mse(render_mask, mask_gt) on out_ray [3,N,6]; the gradient comes out of the same launch"""

    @staticmethod
    def forward(ctx, out_ray, rgb_gt, mask_gt, w_rgb, w_conf, grad_scale=None):
        """grad_scale: the device scalar the backward pass will be seeded with (DynamicLossScaler.state[0:1]) — the stored gradient is then
        already multiplied by it and backward() hands it on as it is (one element-wise launch per step saved)"""
        require_cuda(out_ray, rgb_gt, mask_gt, grad_scale)
        out_ray = out_ray.contiguous().float()
        N = out_ray.shape[1]
        rgb_gt = rgb_gt.reshape(N, 3).contiguous().float()
        mask_gt = mask_gt.reshape(N).contiguous().float() if mask_gt is not None else None
        loss = torch.empty(65, dtype=torch.float32, device=out_ray.device)       # [0] = loss, [1:] = partial sums
        g = torch.empty_like(out_ray)
        check(lib.cnerf_recon_loss_scaled(ptr(out_ray), ptr(rgb_gt), ptr(mask_gt), N, float(w_rgb), float(w_conf), ptr(grad_scale), ptr(loss), ptr(g),
                                          stream()), "recon_loss")
        ctx.save_for_backward(g)
        ctx.scale = grad_scale
        return loss[0]

    @staticmethod
    def backward(ctx, g_loss):
        g, = ctx.saved_tensors
        sc = ctx.scale
        if sc is None:
            return g * g_loss, None, None, None, None, None
        if g_loss.data_ptr() == sc.data_ptr() and g_loss.dtype == sc.dtype:         # seeded with the very scalar that is already in g
            return g, None, None, None, None, None
        return g * (g_loss / sc.reshape(())), None, None, None, None, None           # some other seed: correct for the folded factor


def recon_loss(out_ray, rgb_gt, mask_gt, w_rgb, w_conf, grad_scale=None):
    """utils_init_nerf.py:220-234 on the fused renderer's raw composite output (results['_out_ray'])"""
    return _ReconLoss.apply(out_ray, rgb_gt, mask_gt, w_rgb, w_conf, grad_scale)
