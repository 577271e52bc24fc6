"""Host-side trainer logic that needs no GPU: --batch_rays subsampling (utils_init_nerf.py:210-215)."""
import argparse

import numpy as np
import torch

from customnerf_amd.trainer import ReconTrainer


def test_batch_rays_draw_is_the_references_numpy_draw():
    N = 16 * 16
    g = torch.Generator().manual_seed(0)
    rays_o, rays_d = torch.randn(1, N, 3, generator=g), torch.randn(1, N, 3, generator=g)
    rgbs, mask = torch.rand(N, 3, generator=g), torch.rand(N, 1, generator=g)
    shim = argparse.Namespace(opt=argparse.Namespace(batch_rays=100))
    np.random.seed(4242)
    want = np.random.choice(N, size=[100], replace=False)                 # the reference's call, same global generator state
    np.random.seed(4242)
    o, d, c, m = ReconTrainer.select_rays(shim, rays_o, rays_d, rgbs, mask)
    assert o.shape == (1, 100, 3) and c.shape == (1, 100, 3) and m.shape == (1, 100, 1)
    assert torch.equal(o[0], rays_o[0, want]) and torch.equal(d[0], rays_d[0, want]) and torch.equal(c[0], rgbs[want]) and torch.equal(m[0], mask[want])
    assert len(set(want.tolist())) == 100                                  # without replacement
    shim.opt.batch_rays = 0                                                # default: the whole view, untouched
    assert ReconTrainer.select_rays(shim, rays_o, rays_d, rgbs, mask)[0] is rays_o
