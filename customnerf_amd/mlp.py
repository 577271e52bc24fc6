"""MLP evaluation entry used by `tcnn.Network` (see tcnn.py for the contract): Python side of cnerf_mlp_forward /
cnerf_mlp_backward (customnerf_amd/csrc/mlp.hip, MFMA kernels).  There is no torch / CPU fallback: the library must
be built and the tensors must live on the GPU.

Reference call sites: nerf/network_grid.py:18-54 (RGB_network), :98-139.
"""
import ctypes

import torch
from torch.autograd import Function

from ._lib import lib, check, ptr, stream, F16, F32, require_cuda, scratch_reallocated

_ACT = {"None": 0, "Sigmoid": 1}
_WS = {}


def _workspace(P, cfg, dt, device):
    need = ctypes.c_uint64(0)
    n_in, n_out, n_neurons, n_hidden, _ = cfg
    check(lib.cnerf_mlp_backward_workspace_bytes(P, n_in, n_out, n_neurons, n_hidden, dt, ctypes.addressof(need)), "mlp_backward_workspace_bytes")
    buf = _WS.get(device)
    if buf is None or buf.numel() < need.value:
        buf = torch.empty(int(need.value * 1.1) + 256, dtype=torch.uint8, device=device)
        _WS[device] = buf
        scratch_reallocated()
    return buf


def _rows(x):
    """2-D view with unit inner stride (row stride may exceed the width: column slices are passed without a copy)."""
    if x.stride(-1) != 1 or (x.shape[0] > 1 and x.stride(0) < x.shape[1]):
        x = x.contiguous()
    return x


class MLPFunction(Function):
    """y [P, n_out] = mlp(x [P, n_in]; params).  Nothing is saved but the inputs: the backward recomputes the forward."""

    @staticmethod
    def forward(ctx, x, params, cfg, dtype):
        n_in, n_out, n_neurons, n_hidden, act = cfg
        require_cuda(x, params)
        ctx.x_dtype = x.dtype
        x = _rows(x.to(dtype))
        P = x.shape[0]
        dt = F16 if dtype == torch.float16 else F32
        y = torch.empty(P, n_out, dtype=dtype, device=x.device)
        check(lib.cnerf_mlp_forward(ptr(x), x.stride(0) if P > 1 else max(x.shape[1], 1), ptr(params), P, n_in, n_out, n_neurons, n_hidden, act,
                                    ptr(y), n_out, dt, stream()), "mlp_forward")
        ctx.save_for_backward(x, params)
        ctx.cfg, ctx.dt = cfg, dt
        return y

    @staticmethod
    def backward(ctx, gy):
        x, params = ctx.saved_tensors
        n_in, n_out, n_neurons, n_hidden, act = ctx.cfg
        P = x.shape[0]
        gy = _rows(gy.to(x.dtype))
        gx = torch.empty(P, n_in, dtype=x.dtype, device=x.device) if ctx.needs_input_grad[0] else None
        gp = torch.zeros_like(params)
        if P > 0:
            ws = _workspace(P, ctx.cfg, ctx.dt, x.device)
            check(lib.cnerf_mlp_backward(ptr(x), x.stride(0) if P > 1 else max(x.shape[1], 1), ptr(params), ptr(gy),
                                         gy.stride(0) if P > 1 else n_out, P, n_in, n_out, n_neurons, n_hidden, act,
                                         ptr(gx), n_in, ptr(gp), ptr(ws), ws.numel(), ctx.dt, stream()), "mlp_backward")
        return (gx.to(ctx.x_dtype) if gx is not None else None), gp, None, None


def mlp_forward(x, params, n_in, n_out, n_neurons, n_hidden_layers, output_activation, dtype):
    if x.shape[-1] != n_in:
        raise ValueError(f"expected {n_in} input features, got {x.shape[-1]}")
    prefix = x.shape[:-1]
    cfg = (int(n_in), int(n_out), int(n_neurons), int(n_hidden_layers), _ACT[output_activation])
    y = MLPFunction.apply(x.reshape(-1, n_in), params, cfg, dtype)
    return y.reshape(*prefix, n_out)
