"""Python side of include/customnerf_sd.h: thin wrappers that allocate outputs (torch owns memory) and fill the GEMM
descriptor.  Activations are float16, NHWC / [tokens, channels]; weights arrive packed K-major ([N, K]) in float16,
biases and norm affine parameters in float32.  No torch arithmetic happens here and there is no fallback: every
function ends in a libcustomnerf_hip.so call on torch's current stream (HIP-graph capturable: no host sync)."""
import ctypes

import torch

from .._lib import lib, check, ptr, stream, require_cuda, SdGemmDesc

ACT_NONE, ACT_SILU, ACT_GELU, ACT_QUICK_GELU, ACT_GEGLU = 0, 1, 2, 3, 4
# round 6: GroupNorm statistics / LayerNorm riding on the split-K tail kernel and on the channel concat.  False restores the round-5 launches
# (stand-alone k_gn_stats / k_layernorm) — the A side of scratch/edit_ab.py's same-box comparison; nothing else reads it.
TAIL_FUSION = True
_WS = {}
_PROFILE = None


def set_profile(records):
    """records: list (or None).  While set, every GEMM launch is bracketed by events on the launch stream and appended as
    (start_event, end_event, flops) — bench.py's live MFMA roofline measurement (eager launches only, not graph replays)."""
    global _PROFILE
    _PROFILE = records


_WS_RETIRED = []          # outgrown workspaces stay allocated: HIP graphs captured earlier (UNet.graphed) hold their addresses


def _workspace(nbytes, device):
    """Split-K scratch shared by all GEMM launches on `device`.  When a later GEMM needs more than the current buffer, the old one is
    RETIRED, not freed: a captured graph replays its GEMM nodes with the pointer it was captured with, and a freed block would be
    handed to some other tensor by the caching allocator while those replays keep writing partial sums into it."""
    buf = _WS.get(device)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _WS_RETIRED.append(buf)
        buf = torch.empty(max(int(nbytes), 64 << 20), dtype=torch.uint8, device=device)
        _WS[device] = buf
    return buf


def _launch(d, device):
    """-> True when the library ran the problem split-K (round 6: its tail kernel fills desc.gn_sums like the one-launch epilogue does)"""
    need = ctypes.c_uint64(0)
    check(lib.cnerf_sd_gemm_workspace_bytes(ctypes.byref(d), ctypes.byref(need)), "sd_gemm_workspace_bytes")
    ws = _workspace(need.value, device) if need.value else None
    if not TAIL_FUSION and need.value:
        d.gn_sums = None
    if _PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(lib.cnerf_sd_gemm(ctypes.byref(d), ptr(ws), ws.numel() if ws is not None else 0, stream()), "sd_gemm")
    if _PROFILE is not None:
        e1.record()
        _PROFILE.append((e0, e1, 2.0 * d.M * d.N * d.K * d.batch_outer * d.batch_inner, (int(d.M), int(d.N), int(d.K), int(d.mode), int(d.batch_outer * d.batch_inner), int(d.H_in), int(d.Cin), int(d.tstride), int(d.ups))))
    return need.value > 0


def _attach_gn(d, gn, M, N):
    """gn = (sums [B, G, 2] int64 pre-zeroed, groups, rows_per_image) or None -> whether the request is admissible"""
    if gn is None:
        return False
    sums, groups, rows = gn
    if rows < 64 or M % rows or N % groups or N // groups < 4:
        return False
    d.gn_sums, d.gn_groups, d.gn_rows = ptr(sums), groups, rows
    return True


def _desc(A, B, C, M, N, K, lda, ldb, ldc, bias=None, bias_rows=None, rows_per_bias_row=0, residual=None, ldr=0, act=0, alpha=1.0, C32=None):
    d = SdGemmDesc()
    d.A, d.B, d.C, d.C32 = ptr(A), ptr(B), ptr(C), ptr(C32)
    d.bias, d.bias_rows, d.residual = ptr(bias), ptr(bias_rows), ptr(residual)
    d.M, d.N, d.K, d.lda, d.ldb, d.ldc, d.ldr = M, N, K, lda, ldb, ldc, ldr
    d.rows_per_bias_row, d.alpha, d.act = rows_per_bias_row, alpha, act
    d.batch_outer = d.batch_inner = 1
    d.ups = d.tstride = 1
    return d


def linear(x, w, bias=None, residual=None, act=ACT_NONE, alpha=1.0, out=None, out32=False, gn=None, ln=None):
    """x [..., K] half, w [N, K] half -> [..., N] half (float32 when out32).
    gn = (sums, groups, rows_per_image): also accumulate the GroupNorm statistics of the output into `sums`; returns (y, ok) then.
    ln = (gamma, beta[, eps]): also return LayerNorm(y) over the last dimension -> (y, n).  When the library runs the problem split-K its tail
    kernel writes n while it finishes y (one wave per row); otherwise n comes from a layernorm() launch — the same values either way."""
    require_cuda(x, w)
    K = x.shape[-1]
    N = w.shape[0]
    x2 = x.reshape(-1, K)
    M = x2.shape[0]
    assert w.shape[1] == K and x2.stride(1) == 1 and x2.dtype == torch.float16 and w.dtype == torch.float16
    No = N // 2 if act == ACT_GEGLU else N            # GEGLU pairs (value, gate) -> N / 2 outputs (weights interleaved: pack.interleave_geglu)
    if out is None:
        out = torch.empty(M, No, dtype=torch.float32 if out32 else torch.float16, device=x.device)
    r2 = residual.reshape(M, N) if residual is not None else None
    d = _desc(x2, w, None if out32 else out, M, N, K, x2.stride(0), w.stride(0), out.stride(0), bias=bias, residual=r2,
              ldr=r2.stride(0) if r2 is not None else 0, act=act, alpha=alpha, C32=out if out32 else None)
    want = _attach_gn(d, gn, M, N)
    n_out = None
    if ln is not None and gn is None and not out32 and act != ACT_GEGLU and TAIL_FUSION:
        gamma, beta = ln[0], ln[1]
        n_out = torch.empty(M, N, dtype=torch.float16, device=x.device)
        d.ln_out, d.ln_gamma, d.ln_beta, d.ln_eps = ptr(n_out), ptr(gamma), ptr(beta), float(ln[2]) if len(ln) > 2 else 1e-5
        yes = ctypes.c_int(0)
        check(lib.cnerf_sd_gemm_serves_ln(ctypes.byref(d), ctypes.byref(yes)), "sd_gemm_serves_ln")
        if not yes.value:
            d.ln_out = None
            n_out = None
    split = _launch(d, x.device)
    want = want and (TAIL_FUSION or not split)
    y = out.reshape(*x.shape[:-1], No)
    if ln is not None:
        n = n_out.reshape(y.shape) if n_out is not None else layernorm(y.contiguous(), ln[0], ln[1], float(ln[2]) if len(ln) > 2 else 1e-5)
        return y, n
    return (y, want) if gn is not None else y


def conv2d(x, w, bias, ksize, stride=1, pad=1, ups=1, tstride=1, out_hw=None, bias_rows=None, residual=None, act=ACT_NONE, gn=None):
    """x [B, H, W, Cin] half (NHWC), w [Cout, k*k*Cin] half packed (kh, kw, ci) -> [B, Ho, Wo, Cout] half.
    stride/pad/ups/tstride as in customnerf_sd.h; bias_rows [B, Cout] float32 = per-image bias (time embedding)."""
    require_cuda(x, w)
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    assert x.is_contiguous() and x.dtype == torch.float16 and w.shape[1] == ksize * ksize * Cin
    if out_hw is None:
        Hu, Wu = H * ups, W * ups
        out_hw = ((Hu + 2 * pad - ksize) // stride + 1, (Wu + 2 * pad - ksize) // stride + 1)
    Ho, Wo = out_hw
    y = torch.empty(B, Ho, Wo, Cout, dtype=torch.float16, device=x.device)
    M = B * Ho * Wo
    d = _desc(x, w, y, M, Cout, ksize * ksize * Cin, 0, w.stride(0), Cout, bias=bias, bias_rows=bias_rows, rows_per_bias_row=Ho * Wo if bias_rows is not None else 0,
              residual=residual, ldr=Cout, act=act)
    d.mode, d.Cin, d.H_in, d.W_in, d.H_out, d.W_out = 1, Cin, H, W, Ho, Wo
    d.KH = d.KW = ksize
    if bias_rows is not None:
        assert bias_rows.dtype == torch.float32 and bias_rows.stride(1) == 1
        d.ld_bias_rows = bias_rows.stride(0)
    d.stride, d.pad_t, d.pad_l, d.ups, d.tstride = stride, pad, pad, ups, tstride
    want = _attach_gn(d, gn, M, Cout)
    split = _launch(d, x.device)
    return (y, want and (TAIL_FUSION or not split)) if gn is not None else y


GN_FRAC_BITS = 20        # include/customnerf_sd.h CNERF_SD_GN_FRAC_BITS: GroupNorm statistics are int64 fixed point (exact, order-independent sums)


def gn_sums_to_float(sums):
    """the fixed-point statistics [B, G, 2] as float64 (sum, sum of squares); NaN where the sum is poisoned / out of range (csrc/sd_gn_fix.h:
    bits 63..60 not all equal)"""
    top = sums >> 60
    return torch.where((top == 0) | (top == -1), sums.double() * (1.0 / (1 << GN_FRAC_BITS)), torch.full((), float('nan'), dtype=torch.float64, device=sums.device))


class SumsPool:
    """One pre-zeroed int64 buffer for the GroupNorm statistics of a whole network pass: a single fill launch instead of one
    zero-fill per norm.  take() hands out consecutive [B, G, 2] slices."""

    def __init__(self, n_norms, B, groups, device):
        self.buf = torch.zeros(n_norms, B, groups, 2, dtype=torch.int64, device=device)
        self.i = 0

    def take(self):
        s = self.buf[self.i]
        self.i += 1
        return s


def groupnorm(x, gamma, beta, groups, eps, silu, pool=None, sums=None, sums_ready=True):
    """x [B, ..., C] half -> (y, sums [B, G, 2] int64 fixed point: sum and sum of squares per group, gn_sums_to_float()).
    sums given and sums_ready: the statistics were already accumulated by the GEMM that produced x (its gn= request) — only the apply
    pass runs; sums given, not ready: a pre-zeroed buffer to fill (the producer ran split-K and left it untouched)."""
    require_cuda(x, gamma)
    B, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C)
    assert x.is_contiguous() and x.dtype == torch.float16
    y = torch.empty_like(x)
    mode = (2 if sums_ready else 0) if sums is not None else (0 if pool is not None else 1)
    if sums is None:
        sums = pool.take() if pool is not None else torch.empty(B, groups, 2, dtype=torch.int64, device=x.device)
    check(lib.cnerf_sd_groupnorm_forward(ptr(x), ptr(gamma), ptr(beta), B, HW, C, groups, eps, int(silu), ptr(sums), mode, ptr(y), stream()),
          "sd_groupnorm_forward")
    return y, sums


def groupnorm_backward(x, dy, gamma, beta, groups, eps, silu, sums, residual=None, scratch=None):
    """dx of groupnorm() for frozen gamma / beta.  residual (x's shape, half): a second gradient arriving at x (a skip connection), added inside
    the apply kernel; scratch: a pre-zeroed [B, G, 2] int64 slice (SumsPool.take()) instead of a fresh buffer + its zero-fill launch."""
    B, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C)
    dy = dy.contiguous()
    dx = torch.empty_like(x)
    if residual is not None:
        residual = residual.contiguous()
        assert residual.shape == x.shape and residual.dtype == torch.float16
    zeroed = scratch is not None
    if scratch is None:
        scratch = torch.empty(B, groups, 2, dtype=torch.int64, device=x.device)
    check(lib.cnerf_sd_groupnorm_backward_ex(ptr(x), ptr(dy), ptr(gamma), ptr(beta), B, HW, C, groups, eps, int(silu), ptr(sums), ptr(scratch), int(zeroed),
                                             ptr(residual), ptr(dx), stream()), "sd_groupnorm_backward")
    return dx


def layernorm(x, gamma, beta, eps=1e-5):
    C = x.shape[-1]
    assert x.is_contiguous() and x.dtype == torch.float16
    y = torch.empty_like(x)
    check(lib.cnerf_sd_layernorm_forward(ptr(x), ptr(gamma), ptr(beta), x.numel() // C, C, eps, ptr(y), stream()), "sd_layernorm_forward")
    return y


def geglu(x):
    C = x.shape[-1] // 2
    assert x.is_contiguous() and x.dtype == torch.float16
    y = torch.empty(*x.shape[:-1], C, dtype=torch.float16, device=x.device)
    check(lib.cnerf_sd_geglu(ptr(x), x.numel() // (2 * C), C, ptr(y), stream()), "sd_geglu")
    return y


def add(a, b):
    assert a.shape == b.shape and a.is_contiguous() and b.is_contiguous() and a.dtype == torch.float16
    y = torch.empty_like(a)
    check(lib.cnerf_sd_add(ptr(a), ptr(b), a.numel(), ptr(y), stream()), "sd_add")
    return y


def silu(x):
    assert x.is_contiguous() and x.dtype == torch.float16
    y = torch.empty_like(x)
    check(lib.cnerf_sd_silu(ptr(x), x.numel(), ptr(y), stream()), "sd_silu")
    return y


def concat_channels(a, b, gn=None):
    """channel concat of NHWC half tensors.  gn = (sums [B, G, 2] int64 pre-zeroed, groups, rows_per_image): also accumulate the GroupNorm
    statistics of the result (returns (y, ok) then; ok False = request not admissible, nothing written to sums)."""
    assert a.shape[:-1] == b.shape[:-1] and a.is_contiguous() and b.is_contiguous()
    C1, C2 = a.shape[-1], b.shape[-1]
    y = torch.empty(*a.shape[:-1], C1 + C2, dtype=torch.float16, device=a.device)
    rows = a.numel() // C1
    if gn is not None:
        sums, groups, rpi = gn
        ok = TAIL_FUSION and (C1 + C2) % groups == 0 and (C1 + C2) // groups >= 8 and rows % rpi == 0 and (rows // rpi) * groups <= 512
        check(lib.cnerf_sd_concat_gn(ptr(a), ptr(b), rows, C1, C2, ptr(y), ptr(sums) if ok else None, groups if ok else 0, rpi if ok else 0, stream()), "sd_concat_gn")
        return y, ok
    check(lib.cnerf_sd_concat(ptr(a), ptr(b), rows, C1, C2, ptr(y), stream()), "sd_concat")
    return y


def pad8(n):
    return (n + 7) // 8 * 8


def transpose_batched(src, rows, cols, ld_src, ld_dst, batch, stride_src, stride_dst, dst):
    check(lib.cnerf_sd_transpose(ptr(src), ptr(dst), rows, cols, ld_src, ld_dst, batch, stride_src, stride_dst, stream()), "sd_transpose")
    return dst


def attention_scores(q, k, heads, alpha):
    """q [B, Tq, C], k [B, Tk, C] half -> P [B, heads, Tq, pad8(Tk)] = softmax(alpha q_h k_h^T) (pad columns zero)."""
    B, Tq, C = q.shape
    Tk = k.shape[1]
    d_ = C // heads
    ldS = pad8(Tk)
    assert q.stride(2) == 1 and k.stride(2) == 1          # strided views of a fused qkv projection are fine
    S = torch.empty(B, heads, Tq, ldS, dtype=torch.float16, device=q.device)
    d = _desc(q, k, S, Tq, Tk, d_, q.stride(1), k.stride(1), ldS, alpha=alpha)
    d.batch_outer, d.batch_inner = B, heads
    d.sa_o, d.sa_i, d.sb_o, d.sb_i, d.sc_o, d.sc_i = q.stride(0), d_, k.stride(0), d_, heads * Tq * ldS, Tq * ldS
    _launch(d, q.device)
    check(lib.cnerf_sd_softmax_forward(ptr(S), B * heads * Tq, Tk, ldS, stream()), "sd_softmax_forward")
    return S


def attention_apply(P, v, heads):
    """P [B, heads, Tq, ldS], v [B, Tk, C] half -> O [B, Tq, C] = concat_h P_h v_h."""
    B, _, Tq, ldS = P.shape
    Tk, C = v.shape[1], v.shape[2]
    d_ = C // heads
    assert v.stride(2) == 1
    vT = torch.empty(B, C, ldS, dtype=torch.float16, device=v.device)
    transpose_batched(v, Tk, C, v.stride(1), ldS, B, v.stride(0), C * ldS, vT)
    O = torch.empty(B, Tq, C, dtype=torch.float16, device=v.device)
    d = _desc(P, vT, O, Tq, d_, ldS, ldS, ldS, C)
    d.batch_outer, d.batch_inner = B, heads
    d.sa_o, d.sa_i, d.sb_o, d.sb_i, d.sc_o, d.sc_i = heads * Tq * ldS, Tq * ldS, C * ldS, d_ * ldS, Tq * C, d_
    _launch(d, v.device)
    return O


def attention(q, k, v, heads, causal=False):
    """softmax(q k^T / sqrt(d)) v per head; q [B, Tq, C], k / v [B, Tk, C] half (strided views of a fused projection are fine).
    Head dims <= 160: the fused kernel (scores never materialised); larger heads: explicit scores (attention_scores / _apply)."""
    B, Tq, C = q.shape
    Tk = k.shape[1]
    d_ = C // heads
    if d_ > 160 or d_ % 8:
        assert not causal
        return attention_apply(attention_scores(q, k, heads, 1.0 / float(d_) ** 0.5), v, heads)
    assert q.stride(2) == 1 and k.stride(2) == 1 and v.stride(2) == 1
    if v.stride(1) % 8 == 0 and v.stride(0) % 8 == 0 and v.data_ptr() % 16 == 0:
        # V as the projection leaves it (e.g. the last third of a fused qkv GEMM): the kernel transposes its key tiles in LDS
        out = torch.empty(B, Tq, C, dtype=torch.float16, device=q.device)
        check(lib.cnerf_sd_attention_v(ptr(q), ptr(k), ptr(v), ptr(out), B, heads, Tq, Tk, d_, q.stride(1), q.stride(0), k.stride(1), k.stride(0),
                                       v.stride(1), v.stride(0), C, Tq * C, int(causal), stream()), "sd_attention_v")
        return out
    return attention_vt(q, k, transpose_v(v), heads, causal)


def transpose_v(v):
    """v [B, Tk, C] -> V^T [B, C, round32(Tk)] (pad columns zero): the operand layout of the fused attention kernel"""
    B, Tk, C = v.shape
    ldv = (Tk + 31) // 32 * 32
    vT = torch.empty(B, C, ldv, dtype=torch.float16, device=v.device)
    transpose_batched(v, Tk, C, v.stride(1), ldv, B, v.stride(0), C * ldv, vT)          # zero-fills the pad columns
    return vT


def attention_vt(q, k, vT, heads, causal=False):
    """fused attention with V already transposed (transpose_v): q [B, Tq, C], k [B, Tk, C] (strided views fine), vT [B, C, ldv]"""
    B, Tq, C = q.shape
    Tk = k.shape[1]
    d_ = C // heads
    ldv = vT.shape[2]
    out = torch.empty(B, Tq, C, dtype=torch.float16, device=q.device)
    check(lib.cnerf_sd_attention(ptr(q), ptr(k), ptr(vT), ptr(out), B, heads, Tq, Tk, d_, q.stride(1), q.stride(0), k.stride(1), k.stride(0), ldv, C * ldv,
                                 C, Tq * C, int(causal), stream()), "sd_attention")
    return out


def gemm_nt(A, Bm, M, N, K, lda, ldb, ldc, out, alpha=1.0):
    """out[M, N] = alpha * A[M, K] Bm[N, K]^T on raw (possibly strided) half buffers."""
    d = _desc(A, Bm, out, M, N, K, lda, ldb, ldc, alpha=alpha)
    _launch(d, out.device)
    return out


def softmax_backward_(P, dP, rows, cols, ld):
    check(lib.cnerf_sd_softmax_backward(ptr(P), ptr(dP), rows, cols, ld, stream()), "sd_softmax_backward")
    return dP


def image_to_vae_input(img, Ho, Wo):
    """img [B, 3, Hi, Wi] float32 in [0,1] -> [B, Ho, Wo, 8] half = 2 * bilinear(img) - 1, channels padded to 8."""
    require_cuda(img)
    B, _, Hi, Wi = img.shape
    img = img.contiguous().float()
    out = torch.empty(B, Ho, Wo, 8, dtype=torch.float16, device=img.device)
    check(lib.cnerf_sd_image_to_vae_input(ptr(img), B, Hi, Wi, Ho, Wo, ptr(out), stream()), "sd_image_to_vae_input")
    return out


def image_to_vae_input_backward(d_out, B, Hi, Wi):
    Ho, Wo = d_out.shape[1], d_out.shape[2]
    d_out = d_out.contiguous()
    d_img = torch.empty(B, 3, Hi, Wi, dtype=torch.float32, device=d_out.device)
    check(lib.cnerf_sd_image_to_vae_input_backward(ptr(d_out), B, Hi, Wi, Ho, Wo, ptr(d_img), stream()), "sd_image_to_vae_input_backward")
    return d_img


def clip_preprocess(img, size=224, mean=(0.48145466, 0.4578275, 0.40821073), std=(0.26862954, 0.26130258, 0.27577711)):
    """torchvision Resize(size, BICUBIC, antialias=None) + CenterCrop(size) + Normalize(mean, std) (nerf/clip.py:13-17):
    img [B, 3, H, W] float32 -> [B, 3, size, size] float32."""
    import ctypes
    require_cuda(img)
    B, _, Hi, Wi = img.shape
    img = img.contiguous().float()
    out = torch.empty(B, 3, size, size, dtype=torch.float32, device=img.device)
    m, sdev = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
    check(lib.cnerf_sd_clip_preprocess(ptr(img), B, Hi, Wi, size, ctypes.cast(m, ctypes.c_void_p), ctypes.cast(sdev, ctypes.c_void_p), ptr(out), stream()),
          "sd_clip_preprocess")
    return out


def patchify(x, patch):
    """x [B, 3, S, S] float32 -> [B, (S/patch)^2, patch*patch*3] half, rows in (kh, kw, c) order (pack.pack_conv's K order)."""
    require_cuda(x)
    B, _, S, _ = x.shape
    x = x.contiguous().float()
    out = torch.empty(B, (S // patch) ** 2, patch * patch * 3, dtype=torch.float16, device=x.device)
    check(lib.cnerf_sd_patchify(ptr(x), B, S, patch, ptr(out), stream()), "sd_patchify")
    return out


def timestep_embedding(t, dim):
    """t [B] float32 (device) -> [B, dim] half (cos | sin)."""
    out = torch.empty(t.shape[0], dim, dtype=torch.float16, device=t.device)
    check(lib.cnerf_sd_timestep_embedding(ptr(t), t.shape[0], dim, ptr(out), stream()), "sd_timestep_embedding")
    return out


def add_noise(latents, noise, alpha_bar, out=None):
    """latents, noise [1, 4, h, w] float32 -> UNet input [2, h, w, 8] half (the CFG pair); `out`: a contiguous [2, h, w, 8] half destination
    (one pair of a multi-view batch)."""
    _, _, h, w = latents.shape
    assert latents.shape[0] == 1 and latents.is_contiguous() and noise.is_contiguous()
    if out is None:
        out = torch.empty(2, h, w, 8, dtype=torch.float16, device=latents.device)
    assert out.is_contiguous() and out.dtype == torch.float16 and out.numel() == 2 * h * w * 8
    check(lib.cnerf_sd_add_noise(ptr(latents), ptr(noise), float(alpha_bar), h * w, ptr(out), stream()), "sd_add_noise")
    return out


def sds_grad(eps, noise, alpha_bar, guidance, lambda_sd, out=None):
    """eps [2, h, w, ld] half (uncond, text), noise [1, 4, h, w] float32 -> grad [1, 4, h, w] float32 (`out`: contiguous destination)."""
    _, h, w, ld = eps.shape
    assert eps.is_contiguous() and noise.is_contiguous() and noise.shape[0] == 1
    grad = torch.empty_like(noise) if out is None else out
    assert grad.is_contiguous() and grad.dtype == torch.float32 and grad.numel() == noise.numel()
    check(lib.cnerf_sd_sds_grad(ptr(eps), ld, ptr(noise), float(alpha_bar), float(guidance), float(lambda_sd), h * w, ptr(grad), stream()), "sd_sds_grad")
    return grad


# ---- the small tensors of one editing step, one launch each (csrc/edit_ops.hip)
def _f32c(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.contiguous().float()


def ray_images(out_ray, B, H, W):
    """out_ray [3, B*H*W, 6] float32 (renderer results['_out_ray']) -> the all / fg / bg images [B, 3, H, W] float32 (utils_init_nerf.py:361-366)"""
    require_cuda(out_ray)
    out_ray = _f32c(out_ray)
    assert out_ray.shape == (3, B * H * W, 6)
    imgs = [torch.empty(B, 3, H, W, dtype=torch.float32, device=out_ray.device) for _ in range(3)]
    check(lib.cnerf_edit_ray_images(ptr(out_ray), B, H * W, ptr(imgs[0]), ptr(imgs[1]), ptr(imgs[2]), stream()), "edit_ray_images")
    return imgs


def ray_images_backward(d_imgs, B, H, W, device):
    """three image gradients [B, 3, H, W] (None = zeros) -> d(out_ray) [3, B*H*W, 6]"""
    d = [None if g is None else _f32c(g) for g in d_imgs]
    out = torch.empty(3, B * H * W, 6, dtype=torch.float32, device=device)
    check(lib.cnerf_edit_ray_images_backward(ptr(d[0]), ptr(d[1]), ptr(d[2]), B, H * W, ptr(out), stream()), "edit_ray_images_backward")
    return out


def l1_loss_scaled(a, b, scale):
    """-> (loss [1] = scale * mean|a - b|, dsign = d loss / d a, same shape as a)"""
    require_cuda(a)
    a, b = _f32c(a), _f32c(b)
    assert a.shape == b.shape
    loss = torch.empty(1, dtype=torch.float32, device=a.device)
    dsign = torch.empty_like(a)
    check(lib.cnerf_edit_l1_loss(ptr(a), ptr(b), a.numel(), float(scale), ptr(loss), ptr(dsign), stream()), "edit_l1_loss")
    return loss, dsign


def sds_loss(latents, grad):
    """-> (loss [1] = 0.5 sum d^2, 2 d) with d = latents - (latents - grad); d loss / d latents = (2 d) * 0.5   (sd.py:150-152)"""
    require_cuda(latents)
    latents, grad = _f32c(latents), _f32c(grad)
    assert latents.shape == grad.shape
    loss = torch.empty(1, dtype=torch.float32, device=latents.device)
    diff = torch.empty_like(latents)
    check(lib.cnerf_edit_sds_loss(ptr(latents), ptr(grad), latents.numel(), ptr(loss), ptr(diff), stream()), "edit_sds_loss")
    return loss, diff


def scale_by_scalar(src, scalar, mult=1.0):
    """src * (scalar * mult) with `scalar` a one-element device tensor (no host read)"""
    src = _f32c(src)
    scalar = _f32c(scalar.reshape(1)).to(src.device)
    out = torch.empty_like(src)
    check(lib.cnerf_edit_scale_by_scalar(ptr(src), ptr(scalar), float(mult), src.numel(), ptr(out), stream()), "edit_scale_by_scalar")
    return out


def sample_latents(moments, noise, scaling_factor):
    """moments [B, h, w, 8] half (mean | logvar), noise [B, 4, h, w] float32 -> latents [B, 4, h, w] float32 (sd.py:102-104)"""
    require_cuda(moments)
    B, h, w, c = moments.shape
    assert c == 8 and moments.dtype == torch.float16 and moments.is_contiguous()
    noise = _f32c(noise)
    assert noise.shape == (B, 4, h, w)
    lat = torch.empty(B, 4, h, w, dtype=torch.float32, device=moments.device)
    check(lib.cnerf_sd_sample_latents(ptr(moments), ptr(noise), B, h * w, float(scaling_factor), ptr(lat), stream()), "sd_sample_latents")
    return lat


def sample_latents_backward(moments, noise, d_lat, scaling_factor):
    B, h, w, _ = moments.shape
    d_lat, noise = _f32c(d_lat), _f32c(noise)
    d_m = torch.empty_like(moments)
    check(lib.cnerf_sd_sample_latents_backward(ptr(moments), ptr(noise), ptr(d_lat), B, h * w, float(scaling_factor), ptr(d_m), stream()),
          "sd_sample_latents_backward")
    return d_m


def set_floats(dst, values):
    """a handful of host floats into a float32 device tensor without a staging tensor (one launch, graph-capturable)"""
    vals = [float(v) for v in values]
    assert dst.dtype == torch.float32 and dst.is_contiguous() and dst.numel() == len(vals) <= 16
    arr = (ctypes.c_float * len(vals))(*vals)
    check(lib.cnerf_set_floats(ptr(dst), ctypes.cast(arr, ctypes.c_void_p), len(vals), stream()), "set_floats")
    return dst
