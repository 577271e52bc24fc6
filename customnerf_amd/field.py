"""Fused field evaluation (grid features -> sigma, rgb+confidence) on the matrix cores: Python entry over
cnerf_field_forward / cnerf_field_backward.  Mirrors NeRFNetwork.forward/.density (nerf/network_grid.py:159-193)."""
import torch
from torch.autograd import Function

from ._lib import lib, check, ptr, stream, F16, F32, require_cuda


def field_forward_raw(enc, xyz, dirs, dir_group, enc_dim, n_hidden_geo, n_rgb_out, p_net, p_den, p_rgb, with_rgb=True):
    """enc [L,P,2] (half or float) in kernel layout, xyz [P,3] f32, dirs [ceil(P/dir_group),3] f32 -> sigma [P], rgbc [P,4] | None."""
    require_cuda(enc, xyz, p_net, p_den)
    P = xyz.shape[0]
    dt = F16 if enc.dtype == torch.float16 else F32
    sigma = torch.empty(P, dtype=torch.float32, device=xyz.device)
    rgbc = torch.empty(P, 4, dtype=torch.float32, device=xyz.device) if with_rgb else None
    check(lib.cnerf_field_forward(ptr(enc), ptr(xyz), ptr(dirs) if with_rgb else None, int(dir_group), P, int(enc_dim), int(n_hidden_geo),
                                  int(n_rgb_out), ptr(p_net), ptr(p_den), ptr(p_rgb) if with_rgb else None, ptr(sigma), ptr(rgbc), dt, stream()),
          "field_forward")
    return sigma, rgbc
