"""GPU parity of the generic fully fused MLP (cnerf_mlp_forward/backward behind customnerf_amd.tcnn.Network) against the
oracle's statement of the tinycudann.Network contract (oracle/torch_oracle.py mlp_forward)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import torch_oracle as to      # noqa: E402

# (n_in, n_out, hidden layers, output activation): the reference's call sites (network_grid.py:18-54, 98-139) + ragged widths
CASES = [
    (32, 64, 2, "None"),       # `network`
    (32, 16, 2, "None"),
    (64, 1, 1, "None"),
    (91, 4, 1, "Sigmoid"),
    (91, 3, 1, "Sigmoid"),
    (3, 3, 1, "Sigmoid"),
    (27, 16, 1, "None"),
    (128, 7, 2, "Sigmoid"),
    (100, 5, 2, "None"),
    (16, 40, 1, "Sigmoid"),
]


def _net(n_in, n_out, nh, act, dtype):
    from customnerf_amd import tcnn
    return tcnn.Network(n_in, n_out, {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": act, "n_neurons": 64,
                                      "n_hidden_layers": nh}, dtype=dtype).cuda()


@pytest.mark.parametrize("n_in,n_out,nh,act", CASES)
@pytest.mark.parametrize("half", [False, True], ids=["f32", "f16"])
def test_mlp_forward(n_in, n_out, nh, act, half):
    P = 4133
    dtype = torch.float16 if half else torch.float32
    net = _net(n_in, n_out, nh, act, dtype)
    g = torch.Generator().manual_seed(n_in * 131 + n_out)
    x = torch.rand(P, n_in, generator=g) * 2 - 1
    with torch.no_grad():
        y = net(x.cuda())
        y_ref = to.mlp_forward(x, net.params.detach().cpu(), n_in, n_out, 64, nh, act, half=half)
    assert y.shape == (P, n_out) and y.dtype == dtype
    if half:
        np.testing.assert_allclose(y.float().cpu().numpy(), y_ref.numpy(), rtol=0, atol=4e-3)
    else:
        np.testing.assert_allclose(y.cpu().numpy(), y_ref.numpy(), rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("n_in,n_out,nh,act", [(91, 4, 1, "Sigmoid"), (32, 64, 2, "None"), (100, 5, 2, "Sigmoid"), (3, 3, 1, "None"), (16, 40, 1, "Sigmoid")])
@pytest.mark.parametrize("half", [False, True], ids=["f32", "f16"])
def test_mlp_backward(n_in, n_out, nh, act, half):
    P = 2077
    dtype = torch.float16 if half else torch.float32
    net = _net(n_in, n_out, nh, act, dtype)
    g = torch.Generator().manual_seed(n_in * 17 + n_out)
    x = torch.rand(P, n_in, generator=g) * 2 - 1
    gy = torch.randn(P, n_out, generator=g)
    p_ref = net.params.detach().cpu().clone().requires_grad_(True)
    x_ref = x.clone().requires_grad_(True)
    to.mlp_forward(x_ref, p_ref, n_in, n_out, 64, nh, act, half=half).backward(gy)
    xg = x.cuda().requires_grad_(True)
    net(xg).backward(gy.cuda().to(dtype))
    rt = 3e-2 if half else 1e-3
    for name, a, b in (("params", net.params.grad, p_ref.grad), ("x", xg.grad, x_ref.grad)):
        a, b = a.float().cpu().numpy(), b.numpy()
        scale = float(np.abs(b).max())
        assert scale > 0, name
        err = np.abs(a - b).max() / scale
        assert err < rt, f"{name}: max|diff|/max|ref| = {err:.3e}"
    # padded rows / columns of the parameter matrices get exactly zero gradient
    in_pad = (n_in + 15) // 16 * 16
    g0 = net.params.grad[:64 * in_pad].view(64, in_pad)
    assert torch.all(g0[:, n_in:] == 0)
    out_pad = (n_out + 15) // 16 * 16
    go = net.params.grad[-out_pad * 64:].view(out_pad, 64)
    assert torch.all(go[n_out:] == 0)


def test_mlp_strided_input_empty_and_errors():
    from customnerf_amd import tcnn
    net = _net(27, 3, 1, "Sigmoid", torch.float32)
    big = torch.rand(515, 40).cuda()
    with torch.no_grad():
        y_view = net(big[:, 5:32])                     # column slice: row stride 40, no copy
        y_copy = net(big[:, 5:32].contiguous())
        assert torch.equal(y_view, y_copy)
        assert net(big[:0, 5:32]).shape == (0, 3)
        y3 = net(big[:512, 5:32].reshape(8, 64, 27))
        assert y3.shape == (8, 64, 3) and torch.equal(y3.reshape(-1, 3), y_copy[:512])
    with pytest.raises(ValueError):
        net(big[:, :26])
    with pytest.raises(ValueError):
        tcnn.Network(32, 4, {"n_neurons": 128, "n_hidden_layers": 1})
    wide = tcnn.Network(32, 80, {"n_neurons": 64, "n_hidden_layers": 2}).cuda()      # > 64 outputs: rejected by the library
    with pytest.raises(ValueError):
        wide(torch.rand(8, 32).cuda())
    with pytest.raises(RuntimeError):
        net(torch.rand(8, 27))                         # CPU tensor: no CPU path


def test_three_generic_networks_equal_the_fused_field():
    """The reference's un-fused chain (network_grid.py:159-193: grid features -> network -> density_network / rgb_network, three
    tcnn.Network calls) evaluated with the generic kernels equals the single-launch fused field on the same parameters."""
    from customnerf_amd import tcnn
    from customnerf_amd.field import field_forward_raw
    P, L = 3001, 16
    g = torch.Generator().manual_seed(11)
    enc = (torch.rand(L, P, 2, generator=g) * 2 - 1).cuda()                     # encoder kernel layout
    xyz = ((torch.rand(P, 3, generator=g) * 2 - 1) * 1.5).cuda()
    d = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1).cuda()
    cfg = lambda nh, act="None": {"n_neurons": 64, "n_hidden_layers": nh, "output_activation": act}
    net = tcnn.Network(32, 64, cfg(2), seed=1, dtype=torch.float32).cuda()
    den = tcnn.Network(64, 1, cfg(1), seed=2, dtype=torch.float32).cuda()
    rgb = tcnn.Network(91, 4, cfg(1, "Sigmoid"), seed=3, dtype=torch.float32).cuda()
    with torch.no_grad():
        s_f, c_f = field_forward_raw(enc, xyz, d, 1, 32, 2, 4, net.params, den.params, rgb.params)
        fea = net(enc.permute(1, 0, 2).reshape(P, 32))
        raw = den(fea)[:, 0]
        blob = 5.0 * torch.exp(-(xyz ** 2).sum(-1) / (2 * 0.2 ** 2))
        s_g = torch.exp(raw + blob)
        freqs = [d] + [f(d * 2.0 ** k) for k in range(4) for f in (torch.sin, torch.cos)]
        c_g = rgb(torch.cat(freqs + [fea], dim=-1))
    np.testing.assert_allclose(s_g.cpu().numpy(), s_f.cpu().numpy(), rtol=3e-5, atol=1e-6)
    np.testing.assert_allclose(c_g.cpu().numpy(), c_f.cpu().numpy(), rtol=0, atol=3e-6)
