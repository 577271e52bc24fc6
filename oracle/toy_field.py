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


def toy_sigma(x, theta=None):
    s = 30.0 * torch.exp(-((x - _C.to(x)) ** 2).sum(-1) / 0.3) + 0.5 * (torch.sin(3.0 * x[..., 0]) + 1.0)
    if theta is not None:                                  # theta [6]: a handful of "weights" so that a loss has a parameter gradient to compare
        s = s * (1.0 + theta[0].to(x)) + theta[1].to(x) ** 2 * torch.exp(-(x ** 2).sum(-1))
    return s


def toy_rgbc(x, d, theta=None):
    z = torch.cat([x, d], dim=-1) @ _A.to(x).t() + _B.to(x)
    if theta is not None:
        z = z + theta[2:6].to(x) * (1.0 + x[..., :1])
    return torch.sigmoid(z)


class ToyField:
    """Same call surface the renderer needs: density(x)->{'sigma'}, __call__(x,d)->(sigma, rgbc[P,4], None).  theta: optional [6] tensor
    (requires_grad for a parameter gradient); None = the parameter-free field the first golden vectors were made with."""

    def __init__(self, theta=None):
        self.theta = theta

    def density(self, x):
        return {'sigma': toy_sigma(x, self.theta)}

    def __call__(self, x, d):
        return toy_sigma(x, self.theta), toy_rgbc(x, d, self.theta), None


def toy_eps(x, t, ctx):
    """A closed-form stand-in for the UNet's epsilon prediction (ours, like the rest of this file): x [B, 4, h, w], t a scalar or [B] tensor of
    timesteps, ctx [B, 77, D] text embeddings -> [B, 4, h, w].  Depends on all three inputs, differs between the two halves of a CFG batch."""
    t = torch.as_tensor(t, dtype=torch.float32).reshape(-1, 1, 1, 1).to(x.device) * 1e-3
    c = ctx.float()[:, :, :4].mean(1).reshape(-1, 4, 1, 1).to(x.device)
    ch = torch.arange(4, dtype=torch.float32, device=x.device).reshape(1, 4, 1, 1)
    return torch.tanh(0.7 * x.float() + 0.3 * t - 0.2 * ch) + 0.25 * c * torch.cos(x.float() * (1.0 + 0.5 * ch)) + 0.05 * x.float().roll(1, dims=-1)


_V = torch.tensor([[0.8, -0.5, 0.3], [-0.2, 0.9, 0.4], [0.5, 0.5, -0.7], [0.3, -0.3, 0.6]])


def toy_vae_latents(x):
    """A closed-form stand-in for `vae.encode(x).latent_dist.sample()` (ours): x [B, 3, H, W] in [-1, 1] -> [B, 4, H / 8, W / 8], differentiable."""
    p = torch.nn.functional.avg_pool2d(x.float(), 8)
    return 1.5 * torch.tanh(torch.einsum('oc,bchw->bohw', _V.to(p), p)) + 0.2 * p.mean(1, keepdim=True) * torch.roll(p, 1, dims=-2).mean(1, keepdim=True)


def toy_encode_imgs(imgs):
    """StableDiffusion.encode_imgs (nerf/sd.py:97-105) on the toy VAE: imgs in [0, 1]"""
    return toy_vae_latents(2 * imgs - 1) * 0.18215
