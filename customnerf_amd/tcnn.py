"""`tinycudann.Network`-shaped fully fused MLP (reference call sites: nerf/network_grid.py:18-54, 98-139).

Same constructor and call surface as `tcnn.Network(n_input_dims, n_output_dims, network_config)`: an nn.Module whose only
parameter is one flat float32 vector `params` (checkpoint key `*.params`), bias-free, input/output widths padded to
multiples of 16, ReLU hidden activation, optional sigmoid output.  Layout of `params`: the row-major [out, in] matrices of
the layers, first to last.

tinycudann is third-party, un-vendored and unpinned in the reference (README.md:49-50): its arithmetic is PARITY
UNPINNED; the semantic implemented here is stated in oracle/torch_oracle.py (`mlp_forward`) and DESIGN.md.

compute dtype: float16 (tcnn's: fp16 weights/activations, fp32 accumulate, fp16 output) or float32 (exact mode used
by the parity tests).  `set_default_dtype()` switches the module-wide default.
"""
import math

import torch
import torch.nn as nn

from . import _lib  # noqa: F401  (the HIP library must be present: no fallback)

_DEFAULT_DTYPE = torch.float16


def set_default_dtype(dtype):
    global _DEFAULT_DTYPE
    assert dtype in (torch.float16, torch.float32)
    _DEFAULT_DTYPE = dtype


def pad16(n):
    return (n + 15) // 16 * 16


def layer_dims(n_in, n_out, n_neurons, n_hidden_layers):
    dims, cur = [], pad16(n_in)
    for _ in range(n_hidden_layers):
        dims.append((n_neurons, cur))
        cur = n_neurons
    dims.append((pad16(n_out), cur))
    return dims


class Network(nn.Module):
    def __init__(self, n_input_dims, n_output_dims, network_config, seed=1337, dtype=None):
        super().__init__()
        cfg = dict(network_config)
        if cfg.get("otype", "FullyFusedMLP") not in ("FullyFusedMLP", "CutlassMLP"):
            raise ValueError(f"unsupported network otype {cfg.get('otype')}")
        if cfg.get("activation", "ReLU") != "ReLU":
            raise ValueError("only ReLU hidden activation is supported (all reference call sites use it)")
        self.output_activation = cfg.get("output_activation", "None")
        if self.output_activation not in ("None", "Sigmoid"):
            raise ValueError(f"unsupported output activation {self.output_activation}")
        self.n_input_dims = int(n_input_dims)
        self.n_output_dims = int(n_output_dims)
        self.n_neurons = int(cfg.get("n_neurons", 64))
        self.n_hidden_layers = int(cfg.get("n_hidden_layers", 1))
        if self.n_neurons != 64:
            raise ValueError("n_neurons must be 64 (the only width the reference uses)")
        self.dims = layer_dims(self.n_input_dims, self.n_output_dims, self.n_neurons, self.n_hidden_layers)
        self.dtype = dtype
        self.seed = seed
        n = sum(o * i for o, i in self.dims)
        self.params = nn.Parameter(torch.empty(n, dtype=torch.float32))
        self.reset_parameters()

    def reset_parameters(self):
        g = torch.Generator().manual_seed(self.seed)
        chunks = []
        for o, i in self.dims:                       # Xavier uniform per matrix (tcnn's default initialisation)
            s = math.sqrt(6.0 / (i + o))
            chunks.append((torch.rand(o * i, generator=g) * 2 - 1) * s)
        with torch.no_grad():
            self.params.copy_(torch.cat(chunks))

    @property
    def compute_dtype(self):
        return self.dtype if self.dtype is not None else _DEFAULT_DTYPE

    def matrices(self):
        out, off = [], 0
        for o, i in self.dims:
            out.append(self.params[off:off + o * i].view(o, i))
            off += o * i
        return out

    def forward(self, x):
        from .mlp import mlp_forward
        return mlp_forward(x, self.params, self.n_input_dims, self.n_output_dims, self.n_neurons, self.n_hidden_layers,
                           self.output_activation, self.compute_dtype)

    def extra_repr(self):
        return (f"n_input_dims={self.n_input_dims}, n_output_dims={self.n_output_dims}, n_neurons={self.n_neurons}, "
                f"n_hidden_layers={self.n_hidden_layers}, output_activation={self.output_activation}, n_params={self.params.numel()}")


class NetworkWithInputEncoding(Network):
    """Not used by CustomNeRF; present so `tcnn.NetworkWithInputEncoding` fails with a clear message."""

    def __init__(self, *a, **k):
        raise NotImplementedError("NetworkWithInputEncoding is not on CustomNeRF's path (SURVEY.md §8)")
