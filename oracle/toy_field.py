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
