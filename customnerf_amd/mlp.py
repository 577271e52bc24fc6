"""MLP evaluation entry used by `tcnn.Network` (see tcnn.py for the contract).

ROUND-1 STAGING NOTE: the fused MFMA kernels (cnerf_mlp_forward/backward, cnerf_field_forward/backward) are being
brought up; until they land this module evaluates the same contract with torch matmuls on the GPU so the rest of
the path (grid encoder, marching, compositing kernels) can be validated end to end.  It is NOT a fallback for a
missing library: importing it requires libcustomnerf_hip.so.
"""
import torch
import torch.nn.functional as F

from . import _lib  # noqa: F401


def mlp_forward(x, params, n_in, n_out, n_neurons, n_hidden_layers, output_activation, dtype):
    from .tcnn import layer_dims
    dims = layer_dims(n_in, n_out, n_neurons, n_hidden_layers)
    prefix = x.shape[:-1]
    h = x.reshape(-1, x.shape[-1]).to(dtype)
    if h.shape[-1] < dims[0][1]:
        h = F.pad(h, (0, dims[0][1] - h.shape[-1]))
    off = 0
    for li, (o, i) in enumerate(dims):
        W = params[off:off + o * i].view(o, i).to(dtype)
        off += o * i
        h = F.linear(h, W)
        if li < len(dims) - 1:
            h = torch.relu(h)
        elif output_activation == 'Sigmoid':
            h = torch.sigmoid(h)
    return h[:, :n_out].reshape(*prefix, n_out)
