"""Weight packing for the SDS kernels: diffusers / torch layouts (OIHW convolutions, [out, in] linears, float32) ->
K-major float16 matrices [N, K] consumed by cnerf_sd_gemm.  Done once at load time (host-side plumbing)."""
import torch


def pad8(n):
    return (n + 7) // 8 * 8


def pack_conv(w):
    """[Cout, Cin, kh, kw] -> [Cout, kh*kw*pad8(Cin)] half, K ordered (kh, kw, ci) to match NHWC activations."""
    co, ci, kh, kw = w.shape
    w = w.permute(0, 2, 3, 1)
    if ci % 8:
        w = torch.nn.functional.pad(w, (0, pad8(ci) - ci))
    return w.reshape(co, -1).to(torch.float16).contiguous()


def pack_conv_dgrad(w):
    """Weights of the input-gradient convolution: dX = conv(dY, flip(W)^T).  [Cout, Cin, kh, kw] -> [pad8(Cin), kh*kw*pad8(Cout)]."""
    co, ci, kh, kw = w.shape
    wt = w.flip(2, 3).permute(1, 2, 3, 0)                       # [ci, kh', kw', co]
    if co % 8:
        wt = torch.nn.functional.pad(wt, (0, pad8(co) - co))
    wt = wt.reshape(ci, -1)
    if ci % 8:
        wt = torch.nn.functional.pad(wt, (0, 0, 0, pad8(ci) - ci))
    return wt.to(torch.float16).contiguous()


def pack_linear(w):
    return w.to(torch.float16).contiguous()


def pack_linear_T(w):
    """[out, in] -> [in, out]: the input-gradient of y = x W^T is dx = dy W = dy (W^T)^T."""
    return w.t().to(torch.float16).contiguous()


def f32(t):
    return t.to(torch.float32).contiguous()


def pad_vec8(v):
    n = v.shape[0]
    return torch.nn.functional.pad(v, (0, pad8(n) - n)) if n % 8 else v


def interleave_geglu(w, b):
    """diffusers GEGLU: proj [2*inner, C] = [value rows ; gate rows].  Interleave to (value_0, gate_0, value_1, gate_1, ...) so that the
    GEMM epilogue sees each pair in adjacent columns (ACT_GEGLU) and the 2*inner-wide intermediate is never written."""
    inner = w.shape[0] // 2
    wi = torch.stack([w[:inner], w[inner:]], 1).reshape(2 * inner, -1)
    bi = torch.stack([b[:inner], b[inner:]], 1).reshape(2 * inner)
    return wi, bi
