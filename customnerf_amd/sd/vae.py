"""AutoencoderKL encoder (SD-1.5 configuration), forward and input-gradient, as a graph of libcustomnerf_hip.so primitives.

Replaces `self.vae.encode(imgs).latent_dist` + `.sample() * 0.18215` of StableDiffusion.encode_imgs (nerf/sd.py:97-105),
including the bilinear 512x512 resize in front of it (utils_init_nerf.py:303).  The SDS gradient enters the NeRF only
through this network's input gradient (sd.py:150-152), so every op is an autograd.Function whose backward is again one of
the library's kernels: the input-gradient of a convolution is the same implicit GEMM with flipped/transposed weights (and
the transposed-stride loader for the three downsampling convs), GroupNorm+SiLU has a fused backward, the single-head
mid-block attention is differentiated through explicit P = softmax(QK^T) GEMMs.  Weights are frozen (no weight gradients).
Module tree / parameter names are diffusers' (arch.vae_encoder_params)."""
import torch
from torch.autograd import Function

from . import ops, pack
from .arch import VAE_ATTN_ALIASES
from .edit_fn import SampleLatents


class _Conv(Function):
    """conv (+ residual).  gn_sums: a pre-zeroed [B, G, 2] buffer — the GEMM epilogue accumulates the statistics of the output for the
    GroupNorm that consumes it next (second, non-differentiable output = 1 when it did, 0 when the schedule was split-K)."""

    @staticmethod
    def forward(ctx, x, w, wd, bias, k, stride, pad, out_hw, residual, gn_sums, groups):
        ctx.set_materialize_grads(False)                     # the flag output has no gradient: no zeros are made for it in the backward
        ctx.wd, ctx.k, ctx.stride, ctx.pad, ctx.in_hw = wd, k, stride, pad, (x.shape[1], x.shape[2])
        ctx.has_res = residual is not None
        if gn_sums is None:
            return ops.conv2d(x, w, bias, k, stride=stride, pad=pad, out_hw=out_hw, residual=residual), None
        oh, ow = out_hw if out_hw is not None else ((x.shape[1] + 2 * pad - k) // stride + 1, (x.shape[2] + 2 * pad - k) // stride + 1)
        y, ok = ops.conv2d(x, w, bias, k, stride=stride, pad=pad, out_hw=out_hw, residual=residual, gn=(gn_sums, groups, oh * ow))
        flag = torch.tensor(1 if ok else 0)
        ctx.mark_non_differentiable(flag)
        return y, flag

    @staticmethod
    def backward(ctx, dy, _flag=None):
        dy = dy.contiguous()
        dx = ops.conv2d(dy, ctx.wd, None, ctx.k, stride=1, pad=ctx.k - 1 - ctx.pad, tstride=ctx.stride, out_hw=ctx.in_hw)
        return dx, None, None, None, None, None, None, None, (dy if ctx.has_res else None), None, None


class _GroupNorm(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, silu, sums_in=None, sums_ready=False):
        if sums_in is not None:
            y, sums = ops.groupnorm(x, gamma, beta, groups, eps, silu, sums=sums_in, sums_ready=sums_ready)
        else:
            y, sums = ops.groupnorm(x, gamma, beta, groups, eps, silu)
        ctx.save_for_backward(x, gamma, beta, sums)
        ctx.cfg = (groups, eps, silu)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, sums = ctx.saved_tensors
        groups, eps, silu = ctx.cfg
        return ops.groupnorm_backward(x, dy, gamma, beta, groups, eps, silu, sums), None, None, None, None, None, None, None


class _Linear(Function):
    @staticmethod
    def forward(ctx, x, w, wT, bias, residual):
        ctx.wT = wT
        ctx.has_res = residual is not None
        return ops.linear(x, w, bias=bias, residual=residual)

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        return ops.linear(dy, ctx.wT), None, None, None, (dy if ctx.has_res else None)


class _Attention1Head(Function):
    """softmax(q k^T / sqrt(C)) v for q, k, v [B, T, C] (one head: the VAE mid block)."""

    @staticmethod
    def forward(ctx, q, k, v):
        alpha = 1.0 / float(q.shape[-1]) ** 0.5
        P = ops.attention_scores(q, k, 1, alpha)                     # [B, 1, T, ld]
        ctx.save_for_backward(q, k, v, P)
        ctx.alpha = alpha
        return ops.attention_apply(P, v, 1)

    @staticmethod
    def backward(ctx, dO):
        q, k, v, P = ctx.saved_tensors
        B, T, C = q.shape
        ld = P.shape[-1]
        dO = dO.contiguous()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        half = dict(dtype=torch.float16, device=q.device)
        for b in range(B):
            Pb = P[b, 0]
            PT = torch.empty(ld, ld, **half)                                            # P^T [Tk, Tq]
            ops.transpose_batched(Pb, T, T, ld, ld, 1, 0, 0, PT)
            dOT = torch.empty(C, ld, **half)
            ops.transpose_batched(dO[b], T, C, C, ld, 1, 0, 0, dOT)
            ops.gemm_nt(PT, dOT, T, C, ld, ld, ld, C, dv[b])                           # dV = P^T dO
            dP = torch.empty(T, ld, **half)
            ops.gemm_nt(dO[b], v[b], T, T, C, C, C, ld, dP)                            # dP = dO V^T
            ops.softmax_backward_(Pb, dP, T, T, ld)                                     # dS (pad columns zero)
            kT = torch.empty(C, ld, **half)
            ops.transpose_batched(k[b], T, C, C, ld, 1, 0, 0, kT)
            ops.gemm_nt(dP, kT, T, C, ld, ld, ld, C, dq[b], alpha=ctx.alpha)           # dQ = alpha dS K
            dST = torch.empty(ld, ld, **half)
            ops.transpose_batched(dP, T, T, ld, ld, 1, 0, 0, dST)
            qT = torch.empty(C, ld, **half)
            ops.transpose_batched(q[b], T, C, C, ld, 1, 0, 0, qT)
            ops.gemm_nt(dST, qT, T, C, ld, ld, ld, C, dk[b], alpha=ctx.alpha)          # dK = alpha dS^T Q
        return dq, dk, dv


class _ImageInput(Function):
    @staticmethod
    def forward(ctx, img, Ho, Wo):
        ctx.shape = img.shape
        return ops.image_to_vae_input(img, Ho, Wo)

    @staticmethod
    def backward(ctx, d_out):
        B, _, Hi, Wi = ctx.shape
        return ops.image_to_vae_input_backward(d_out, B, Hi, Wi), None, None


class _ConvW:
    def __init__(self, sd, p, dev):
        w = sd[p + ".weight"].to(dev)
        self.k = w.shape[-1]
        self.w, self.wd, self.b = pack.pack_conv(w), pack.pack_conv_dgrad(w), pack.f32(sd[p + ".bias"].to(dev))

    def __call__(self, x, stride=1, pad=None, out_hw=None, residual=None, gn=None):
        """gn = (pool, groups): also return the GroupNorm statistics of the output -> (y, (sums, ready)); else -> y"""
        pad = (self.k // 2) if pad is None else pad
        if gn is None:
            return _Conv.apply(x, self.w, self.wd, self.b, self.k, stride, pad, out_hw, residual, None, 0)[0]
        sums = gn[0].take()
        y, flag = _Conv.apply(x, self.w, self.wd, self.b, self.k, stride, pad, out_hw, residual, sums, gn[1])
        return y, (sums, bool(flag))


class _LinW:
    def __init__(self, sd, p, dev):
        w = sd[p + ".weight"].to(dev)
        w = w.reshape(w.shape[0], -1)                                    # 1x1 convolutions are linears over NHWC channels
        self.w, self.wT, self.b = pack.pack_linear(w), pack.pack_linear_T(w), pack.f32(sd[p + ".bias"].to(dev))

    def __call__(self, x, residual=None):
        return _Linear.apply(x, self.w, self.wT, self.b, residual)


class _ResnetFn(Function):
    """A whole residual block (norm1 + SiLU -> conv1 -> norm2 + SiLU -> conv2 + shortcut) with a hand-written backward for frozen weights.
    Against the per-op Functions autograd chains (round 4): the skip gradient joins the branch gradient INSIDE norm1's backward apply kernel
    (one rounding to half, no add launch), and both backward norms take their scratch statistics pre-zeroed from the pass's pool (no
    zero-fill launches).  Outputs: y, the statistics of y for the next norm (or None), a 0 / 1 flag tensor: statistics ready."""

    @staticmethod
    def forward(ctx, x, blk, groups, eps, pool, x_sums, x_ready, out_gn):
        ctx.set_materialize_grads(False)                     # (statistics and flag outputs: no zero tensors for them in the backward)
        B, H, W, _ = x.shape
        if x_sums is not None:
            h, s1 = ops.groupnorm(x, *blk.n1, groups, eps, True, sums=x_sums, sums_ready=x_ready)
        else:
            h, s1 = ops.groupnorm(x, *blk.n1, groups, eps, True, pool)
        c1w, c2w = blk.c1, blk.c2
        s2 = pool.take()
        c1, ok1 = ops.conv2d(h, c1w.w, c1w.b, c1w.k, stride=1, pad=c1w.k // 2, gn=(s2, groups, H * W))
        h, _ = ops.groupnorm(c1, *blk.n2, groups, eps, True, sums=s2, sums_ready=ok1)
        res = ops.linear(x, blk.sc.w, bias=blk.sc.b) if blk.sc is not None else x
        so, ok = None, False
        if out_gn:
            so = pool.take()
            y, ok = ops.conv2d(h, c2w.w, c2w.b, c2w.k, stride=1, pad=c2w.k // 2, residual=res, gn=(so, groups, H * W))
        else:
            y = ops.conv2d(h, c2w.w, c2w.b, c2w.k, stride=1, pad=c2w.k // 2, residual=res)
        ctx.save_for_backward(x, s1, c1, s2, pool.take(), pool.take())       # + two pre-zeroed scratch slices for the backward norms
        ctx.blk, ctx.cfg = blk, (groups, eps, H, W)
        flag = torch.tensor(1 if ok else 0)
        ctx.mark_non_differentiable(flag)
        if so is not None:
            ctx.mark_non_differentiable(so)
        return y, so, flag

    @staticmethod
    def backward(ctx, dy, _so=None, _flag=None):
        x, s1, c1, s2, scr2, scr1 = ctx.saved_tensors
        if getattr(ctx, "scratch_used", False):       # a second backward through the same graph (retain_graph): the pooled slices hold the first
            scr2 = scr1 = None                        # pass's sums — let the calls allocate and zero their own
        ctx.scratch_used = True
        blk = ctx.blk
        groups, eps, H, W = ctx.cfg
        dy = dy.contiguous()
        c1w, c2w = blk.c1, blk.c2
        d = ops.conv2d(dy, c2w.wd, None, c2w.k, stride=1, pad=c2w.k - 1 - c2w.k // 2, out_hw=(H, W))
        d = ops.groupnorm_backward(c1, d, *blk.n2, groups, eps, True, s2, scratch=scr2)
        d = ops.conv2d(d, c1w.wd, None, c1w.k, stride=1, pad=c1w.k - 1 - c1w.k // 2, out_hw=(H, W))
        skip = ops.linear(dy, blk.sc.wT) if blk.sc is not None else dy
        dx = ops.groupnorm_backward(x, d, *blk.n1, groups, eps, True, s1, residual=skip, scratch=scr1)
        return dx, None, None, None, None, None, None, None


class _Resnet:
    def __init__(self, sd, p, dev):
        self.n1 = (pack.f32(sd[p + "norm1.weight"].to(dev)), pack.f32(sd[p + "norm1.bias"].to(dev)))
        self.n2 = (pack.f32(sd[p + "norm2.weight"].to(dev)), pack.f32(sd[p + "norm2.bias"].to(dev)))
        self.c1, self.c2 = _ConvW(sd, p + "conv1", dev), _ConvW(sd, p + "conv2", dev)
        self.sc = _LinW(sd, p + "conv_shortcut", dev) if (p + "conv_shortcut.weight") in sd else None

    def __call__(self, x, groups, eps, pool, x_sums=None, out_gn=False):
        """x_sums = (sums, ready) of x from its producer (or None); out_gn: also return the statistics of the output -> (y, (sums, ready) | None)"""
        xs, xr = x_sums if x_sums is not None else (None, False)
        y, so, flag = _ResnetFn.apply(x, self, groups, eps, pool, xs, xr, out_gn)
        return y, ((so, bool(flag)) if out_gn else None)


class VAEEncoder:
    def __init__(self, cfg, state_dict, device="cuda"):
        self.cfg = cfg
        dev = torch.device(device)
        sd = dict(state_dict)
        for old, new in VAE_ATTN_ALIASES.items():                       # older diffusers checkpoints
            for suffix in (".weight", ".bias"):
                k_old = "encoder.mid_block.attentions.0." + old + suffix
                if k_old in sd:
                    sd["encoder.mid_block.attentions.0." + new + suffix] = sd.pop(k_old)
        boc = cfg["block_out_channels"]
        self.conv_in = _ConvW(sd, "encoder.conv_in", dev)
        self.down = []
        for i in range(len(boc)):
            res = [_Resnet(sd, f"encoder.down_blocks.{i}.resnets.{j}.", dev) for j in range(cfg["layers_per_block"])]
            ds = _ConvW(sd, f"encoder.down_blocks.{i}.downsamplers.0.conv", dev) if i < len(boc) - 1 else None
            self.down.append((res, ds))
        self.mid0, self.mid1 = _Resnet(sd, "encoder.mid_block.resnets.0.", dev), _Resnet(sd, "encoder.mid_block.resnets.1.", dev)
        a = "encoder.mid_block.attentions.0."
        self.an = (pack.f32(sd[a + "group_norm.weight"].to(dev)), pack.f32(sd[a + "group_norm.bias"].to(dev)))
        self.aq, self.ak, self.av, self.ao = (_LinW(sd, a + n, dev) for n in ("to_q", "to_k", "to_v", "to_out.0"))
        self.no = (pack.f32(sd["encoder.conv_norm_out.weight"].to(dev)), pack.f32(sd["encoder.conv_norm_out.bias"].to(dev)))
        self.conv_out = _ConvW(sd, "encoder.conv_out", dev)
        self.quant = _LinW(sd, "quant_conv", dev)

    def moments(self, x):
        """x [B, H, W, 8] half NHWC in [-1, 1] (3 channels + zero padding) -> [B, H/8, W/8, 2*latent] half (mean | logvar)"""
        G, eps = self.cfg["groups"], self.cfg["eps"]
        n_res = sum(len(res) for res, _ in self.down) + 2
        pool = ops.SumsPool(5 * n_res + 8, x.shape[0], G, x.device)          # one zero-fill: forward statistics (filled by the producing GEMMs) + backward scratch
        h, hs = self.conv_in(x, gn=(pool, G))
        for bi, (res, ds) in enumerate(self.down):
            for j, r in enumerate(res):
                feeds_norm = not (ds is not None and j == len(res) - 1)       # the last resnet of a block feeds the downsample conv
                h, hs = r(h, G, eps, pool, x_sums=hs, out_gn=feeds_norm)
            if ds is not None:
                h, hs = ds(h, stride=2, pad=0, out_hw=(h.shape[1] // 2, h.shape[2] // 2), gn=(pool, G))      # F.pad(0,1,0,1) + stride-2 conv
        h, hs = self.mid0(h, G, eps, pool, x_sums=hs, out_gn=True)
        B, H, W, C = h.shape
        n = _GroupNorm.apply(h, *self.an, G, eps, False, *hs).view(B, H * W, C)
        o = _Attention1Head.apply(self.aq(n), self.ak(n), self.av(n))
        h = self.ao(o, residual=h.view(B, H * W, C)).view(B, H, W, C)
        h, hs = self.mid1(h, G, eps, pool, out_gn=True)
        h = self.conv_out(_GroupNorm.apply(h, *self.no, G, eps, True, *hs))
        return self.quant(h)

    def encode_imgs(self, imgs, sample_noise, resize=(512, 512)):
        """StableDiffusion.encode_imgs (sd.py:97-105) with the caller's resize (utils_init_nerf.py:303) folded in:
        imgs [B, 3, H, W] float32 in [0, 1] -> latents [B, latent, h, w] float32 = (mean + std * sample_noise) * 0.18215.
        Differentiable w.r.t. imgs."""
        x = _ImageInput.apply(imgs, resize[0], resize[1])
        return SampleLatents.apply(self.moments(x).contiguous(), sample_noise, self.cfg["scaling_factor"])     # clamp / exp / sample / scale: one launch
