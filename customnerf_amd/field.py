"""Fused field evaluation (grid features -> sigma, rgb+confidence) on the matrix cores: Python entry over
cnerf_field_forward / cnerf_field_backward.  Mirrors NeRFNetwork.forward/.density (nerf/network_grid.py:159-193)."""
import torch
from torch.autograd import Function

from ._lib import lib, check, ptr, stream, F16, F32, require_cuda, scratch_key, grad_chain_wait, grad_chain_record, scratch_reallocated


def capture_id():
    """capture sequence id of the current stream while it records a hipGraph, else 0"""
    import ctypes
    cid = ctypes.c_uint64(0)
    check(lib.cnerf_stream_capture_id(stream(), ctypes.addressof(cid)), "stream_capture_id")
    return cid.value


def packed_weights(cache, enc_dim, n_hidden_geo, n_rgb_out, p_net, p_den, p_rgb):
    """fp16 MFMA fragment image of the three MLPs (cnerf_field_pack_weights) for the current stream, repacked when a parameter moved.

    Every field launch otherwise re-derives that image from the float32 parameters (14 us per launch, three launches per training step).
    `cache`: a dict owned by the module whose parameters these are.  Freshness = (address, autograd version, fused-optimiser epoch) of the
    three parameters — which a write through `.data` does NOT move: only callers that control every parameter write for as long as they
    hold the image may use it (the trainers switch it on for the duration of a training step: NeRFNetwork.packed_field_weights); everyone
    else leaves it off and the kernels stage from the float32 parameters, which is always right.
    Under stream capture a pack is RECORDED, not executed: the first use inside a capture (and any use after a parameter moved inside it)
    records one, so every replay starts from the parameters as they are then; the eager freshness is dropped, because the replays rewrite
    the image at times this cache does not see."""
    import ctypes
    require_cuda(p_net, p_den, p_rgb)
    key = tuple((p.data_ptr(), p._version, getattr(p, '_cnerf_epoch', 0)) for p in (p_net, p_den, p_rgb))
    slot = (scratch_key(p_net.device), enc_dim, n_hidden_geo, n_rgb_out)
    ent = cache.get(slot)
    if ent is None:
        need = ctypes.c_uint64(0)
        check(lib.cnerf_field_weight_image_bytes(int(enc_dim), int(n_hidden_geo), int(n_rgb_out), ctypes.addressof(need)), "field_weight_image_bytes")
        ent = cache[slot] = {'img': torch.empty(need.value, dtype=torch.uint8, device=p_net.device), 'key': None, 'cap': 0, 'cap_key': None}
    cap = capture_id()
    if cap:
        fresh = ent['cap'] == cap and ent['cap_key'] == key
        ent['cap'], ent['cap_key'], ent['key'] = cap, key, None
    else:
        fresh = ent['key'] == key
        ent['key'] = key
    if not fresh:
        img = ent['img']
        check(lib.cnerf_field_pack_weights(int(enc_dim), int(n_hidden_geo), int(n_rgb_out), ptr(p_net), ptr(p_den), ptr(p_rgb), ptr(img), img.numel(),
                                           stream()), "field_pack_weights")
    return ent['img']


def field_forward_raw(enc, xyz, dirs, dir_group, enc_dim, n_hidden_geo, n_rgb_out, p_net, p_den, p_rgb, with_rgb=True, wimg=None):
    """enc [L,P',2] (half or float) in kernel layout with P' >= P samples per level (the first P are read), xyz [P,3] f32,
    dirs [ceil(P/dir_group),3] f32 -> sigma [P], rgbc [P,4] | None.  wimg: packed_weights() image of the parameters (half precision), or None."""
    require_cuda(enc, xyz, p_net, p_den, wimg)
    P = xyz.shape[0]
    assert enc.is_contiguous() and enc.shape[1] >= P
    dt = F16 if enc.dtype == torch.float16 else F32
    sigma = torch.empty(P, dtype=torch.float32, device=xyz.device)
    rgbc = torch.empty(P, 4, dtype=torch.float32, device=xyz.device) if with_rgb else None
    check(lib.cnerf_field_forward_img(ptr(enc), ptr(xyz), ptr(dirs) if with_rgb else None, int(dir_group), P, int(enc_dim), int(n_hidden_geo),
                                      int(n_rgb_out), ptr(p_net), ptr(p_den), ptr(p_rgb) if with_rgb else None, ptr(sigma), ptr(rgbc), dt,
                                      int(enc.shape[1]), ptr(wimg), stream()), "field_forward")
    return sigma, rgbc


def field_forward_rows(enc, row0, xyz_rows, dirs, dir_group, enc_dim, n_hidden_geo, n_rgb_out, p_net, p_den, p_rgb, sigma_out, rgbc_out, wimg=None):
    """Full evaluation (sigma + rgb/confidence) of rows row0 .. row0 + len(xyz_rows) of the kernel-layout feature buffer enc [L, P, 2], written
    into sigma_out [n] / rgbc_out [n, 4] (views into buffers that cover the whole sample list).  renderer._run_fused evaluates the coarse block
    as soon as its features exist — its sigma feeds the importance sampling, so the separate density-only pass of the reference
    (renderer.py:326) costs nothing extra — and the fine block after its gather."""
    require_cuda(enc, xyz_rows, dirs, p_net, p_den, p_rgb, sigma_out, rgbc_out, wimg)
    n = xyz_rows.shape[0]
    assert enc.is_contiguous() and row0 + n <= enc.shape[1] and xyz_rows.is_contiguous() and dirs.is_contiguous()
    assert sigma_out.is_contiguous() and rgbc_out.is_contiguous() and sigma_out.numel() == n and rgbc_out.numel() == 4 * n
    dt = F16 if enc.dtype == torch.float16 else F32
    src = enc.data_ptr() + row0 * enc.shape[2] * enc.element_size()
    check(lib.cnerf_field_forward_img(src, ptr(xyz_rows), ptr(dirs), int(dir_group), n, int(enc_dim), int(n_hidden_geo), int(n_rgb_out),
                                      ptr(p_net), ptr(p_den), ptr(p_rgb), ptr(sigma_out), ptr(rgbc_out), dt, int(enc.shape[1]), ptr(wimg), stream()),
          "field_forward")


_WS = {}


def _workspace(P, enc_dim, n_hidden_geo, n_rgb_out, dt, device):
    import ctypes
    need = ctypes.c_uint64(0)
    check(lib.cnerf_field_backward_workspace_bytes(P, enc_dim, n_hidden_geo, n_rgb_out, dt, ctypes.addressof(need)), "field_backward_workspace_bytes")
    key = scratch_key(device)
    buf = _WS.get(key)
    if buf is None or buf.numel() < need.value:
        buf = torch.empty(int(need.value * 1.1) + 256, dtype=torch.uint8, device=device)
        _WS[key] = buf
        scratch_reallocated()
    return buf


class FieldFunction(Function):
    """sigma [P], rgbc [P,4] = field(enc [L,P,2], xyz, dirs; params).  Differentiable in enc and the three parameter vectors
    (positions / directions get no gradient on CustomNeRF's path).  Nothing is saved but the inputs: the backward recomputes."""

    @staticmethod
    def forward(ctx, enc, xyz, dirs, dir_group, enc_dim, n_hidden_geo, n_rgb_out, p_net, p_den, p_rgb, grad_in_place=False, wimg=None):
        ctx.params = (p_net, p_den, p_rgb) if grad_in_place else None          # the Parameter objects themselves (their .grad is the target)
        ctx.wimg = wimg                                                        # (the backward recomputes from the same parameters: same image)
        xyz = xyz.contiguous().float()
        dirs = dirs.contiguous().float()
        sigma, rgbc = field_forward_raw(enc, xyz, dirs, dir_group, enc_dim, n_hidden_geo, n_rgb_out, p_net, p_den, p_rgb, True, wimg)
        ctx.save_for_backward(enc, xyz, dirs, p_net, p_den, p_rgb)
        ctx.cfg = (dir_group, enc_dim, n_hidden_geo, n_rgb_out)
        return sigma, rgbc

    @staticmethod
    def backward(ctx, g_sigma, g_rgbc):
        enc, xyz, dirs, p_net, p_den, p_rgb = ctx.saved_tensors
        dir_group, enc_dim, n_hidden_geo, n_rgb_out = ctx.cfg
        P = xyz.shape[0]
        dt = F16 if enc.dtype == torch.float16 else F32
        # early termination: the compositing backward (render_ops._CompositeRunIndexed with flush_half_zero) hands over, on the gradient tensor itself,
        # one byte per 32-row tile of the sample list — 0 = every output gradient of the tile is exactly zero (autograd passes the very tensor
        # object on; a gradient that was accumulated or copied on the way simply arrives without the attribute and nothing is skipped)
        # Sound only while both gradients are EXACTLY what the compositing backward wrote: a second consumer of sigma / rgbc makes autograd
        # accumulate — in place into the first arrival (same object, attribute and all: the version counter moves) or into a new tensor — and
        # the dead-tile flags would then drop real gradient.  The hand-over therefore carries address + version of both tensors.
        tile_live = None
        flags = getattr(g_sigma, '_cnerf_tile_live', None)
        if flags is not None:
            tl, ps, vs, pc, vc = flags
            if (g_sigma.data_ptr() == ps and g_sigma._version == vs and g_rgbc.data_ptr() == pc and g_rgbc._version == vc and g_sigma.is_contiguous()
                    and g_rgbc.is_contiguous() and g_sigma.dtype == torch.float32 and g_rgbc.dtype == torch.float32 and tl.numel() * 32 == P and tl.is_cuda):
                tile_live = tl
        g_sigma = g_sigma.contiguous().float()
        g_rgbc = g_rgbc.contiguous().float()
        g_enc = torch.empty_like(enc)
        # the reduction of the per-workgroup partials ADDS into its destination: with persistent .grad buffers (trainer.flat_grad_buffer)
        # it adds straight into them — no zero-fill and no AccumulateGrad passes for the three parameter vectors
        in_place = ctx.params is not None and all(p.grad is not None and p.grad.dtype == torch.float32 and p.grad.is_contiguous() and
                                                  p.grad.device == p.device for p in ctx.params)
        if in_place:
            g_net, g_den, g_rgb = (p.grad for p in ctx.params)
        else:
            g_all = torch.zeros(p_net.numel() + p_den.numel() + p_rgb.numel(), dtype=torch.float32, device=p_net.device)  # one fill, three views
            g_net, g_den, g_rgb = (t.view_as(p) for t, p in zip(g_all.split([p_net.numel(), p_den.numel(), p_rgb.numel()]), (p_net, p_den, p_rgb)))
        ws = _workspace(P, enc_dim, n_hidden_geo, n_rgb_out, dt, xyz.device)
        if in_place:
            grad_chain_wait(xyz.device)                                # the partial reduction adds into the shared .grad buffers
        check(lib.cnerf_field_backward_img(ptr(enc), ptr(xyz), ptr(dirs), int(dir_group), P, int(enc_dim), int(n_hidden_geo), int(n_rgb_out),
                                           ptr(p_net), ptr(p_den), ptr(p_rgb), ptr(g_sigma), ptr(g_rgbc), ptr(g_enc), ptr(g_net), ptr(g_den), ptr(g_rgb),
                                           ptr(ws), ws.numel(), dt, ptr(tile_live), ptr(getattr(ctx, 'wimg', None)), stream()), "field_backward")
        if in_place:
            grad_chain_record(xyz.device)
            return g_enc, None, None, None, None, None, None, None, None, None, None, None
        return g_enc, None, None, None, None, None, None, g_net, g_den, g_rgb, None, None


class FieldAttach(Function):
    """FieldFunction for outputs that were already computed block by block with field_forward_rows: forward hands (sigma, rgbc) on, backward is
    FieldFunction's (it recomputes the forward from the saved inputs, so nothing else has to be kept)."""

    @staticmethod
    def forward(ctx, enc, xyz, dirs, dir_group, enc_dim, n_hidden_geo, n_rgb_out, p_net, p_den, p_rgb, grad_in_place, sigma, rgbc, wimg=None):
        ctx.params = (p_net, p_den, p_rgb) if grad_in_place else None
        ctx.wimg = wimg
        ctx.save_for_backward(enc, xyz.contiguous().float(), dirs.contiguous().float(), p_net, p_den, p_rgb)
        ctx.cfg = (dir_group, enc_dim, n_hidden_geo, n_rgb_out)
        return sigma.detach(), rgbc.detach()

    @staticmethod
    def backward(ctx, g_sigma, g_rgbc):
        return FieldFunction.backward(ctx, g_sigma, g_rgbc)[:11] + (None, None, None)


def field_attach(enc, xyz, dirs, dir_group, enc_dim, n_hidden_geo, n_rgb_out, p_net, p_den, p_rgb, sigma, rgbc, grad_in_place=False, wimg=None):
    return FieldAttach.apply(enc, xyz, dirs, dir_group, enc_dim, n_hidden_geo, n_rgb_out, p_net, p_den, p_rgb, grad_in_place, sigma, rgbc, wimg)


def field(enc, xyz, dirs, dir_group, enc_dim, n_hidden_geo, n_rgb_out, p_net, p_den, p_rgb, grad_in_place=False, wimg=None):
    return FieldFunction.apply(enc, xyz, dirs, dir_group, enc_dim, n_hidden_geo, n_rgb_out, p_net, p_den, p_rgb, grad_in_place, wimg)
