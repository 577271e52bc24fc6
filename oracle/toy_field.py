"""ORACLE — TEST INFRASTRUCTURE ONLY.

A closed-form field (no parameters, no native code) used to drive the reference's renderer when the golden
vectors are generated (tests/golden/make_golden.py) and to drive the oracle / product renderers in the tests
that replay those vectors.  It is ours, not the reference's.
"""
import torch

_A = torch.tensor([[0.9, -0.4, 0.3, 0.5, 0.1, -0.7],
                   [-0.6, 0.8, 0.2, -0.3, 0.9, 0.4],
                   [0.1, 0.5, -0.9, 0.6, -0.2, 0.3],
                   [2.5, -1.5, 2.0, 0.0, 0.0, 0.0]])
_B = torch.tensor([0.1, -0.2, 0.3, 0.4])
_C = torch.tensor([0.2, 0.0, -0.1])


def toy_sigma(x):
    return 30.0 * torch.exp(-((x - _C.to(x)) ** 2).sum(-1) / 0.3) + 0.5 * (torch.sin(3.0 * x[..., 0]) + 1.0)


def toy_rgbc(x, d):
    return torch.sigmoid(torch.cat([x, d], dim=-1) @ _A.to(x).t() + _B.to(x))


class ToyField:
    """Same call surface the renderer needs: density(x)->{'sigma'}, __call__(x,d)->(sigma, rgbc[P,4], None)."""

    def density(self, x):
        return {'sigma': toy_sigma(x)}

    def __call__(self, x, d):
        return toy_sigma(x), toy_rgbc(x, d), None


def toy_eps(x, t, ctx):
    """A closed-form stand-in for the UNet's epsilon prediction (ours, like the rest of this file): x [B, 4, h, w], t a scalar or [B] tensor of
    timesteps, ctx [B, 77, D] text embeddings -> [B, 4, h, w].  Depends on all three inputs, differs between the two halves of a CFG batch."""
    t = torch.as_tensor(t, dtype=torch.float32).reshape(-1, 1, 1, 1).to(x.device) * 1e-3
    c = ctx.float()[:, :, :4].mean(1).reshape(-1, 4, 1, 1).to(x.device)
    ch = torch.arange(4, dtype=torch.float32, device=x.device).reshape(1, 4, 1, 1)
    return torch.tanh(0.7 * x.float() + 0.3 * t - 0.2 * ch) + 0.25 * c * torch.cos(x.float() * (1.0 + 0.5 * ch)) + 0.05 * x.float().roll(1, dims=-1)
