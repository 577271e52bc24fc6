"""Optimiser step of the reference's recipe (main.py:182: Adam betas (0.9, 0.99), eps 1e-15, grid lr x10 via
NeRFNetwork.get_params; main.py:189: lr * 0.1^(iter/iters)) as one fused HIP launch per parameter tensor:
un-scale + moments + update + fp16 shadow refresh + gradient zeroing."""
import torch

from ._lib import lib, check, ptr, stream, require_cuda


def adam_step(p, g, m, v, lr, betas=(0.9, 0.99), eps=1e-15, step=1, grad_scale_inv=1.0, zero_grad=True, p_half=None):
    require_cuda(p, g, m, v, p_half)
    check(lib.cnerf_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), ptr(p_half), p.numel(), float(lr), float(betas[0]), float(betas[1]),
                              float(eps), int(step), float(grad_scale_inv), int(zero_grad), stream()), "adam_step")


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam-compatible surface (param groups with per-group lr) over cnerf_adam_step.
    `half_shadows`: {param: fp16 tensor} refreshed in the same pass (GridEncoder.set_half_table)."""

    def __init__(self, params, lr=5e-4, betas=(0.9, 0.99), eps=1e-15, zero_grad_in_step=True):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.zero_grad_in_step = zero_grad_in_step
        self.half_shadows = {}
        self.grad_scale_inv = 1.0

    @torch.no_grad()
    def step(self, closure=None):
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st['step'] = 0
                    st['exp_avg'] = torch.zeros_like(p)
                    st['exp_avg_sq'] = torch.zeros_like(p)
                st['step'] += 1
                adam_step(p.data, p.grad, st['exp_avg'], st['exp_avg_sq'], group['lr'], group['betas'], group['eps'], st['step'],
                          self.grad_scale_inv, self.zero_grad_in_step, self.half_shadows.get(p))
                # the kernel writes through the raw pointer, which does not bump torch's version counter:
                # advance our own epoch so caches keyed on the parameter (GridEncoder.half_table) notice.
                p._cnerf_epoch = getattr(p, '_cnerf_epoch', 0) + 1

    def zero_grad(self, set_to_none=False):
        if self.zero_grad_in_step and not set_to_none:
            return                      # gradients were zeroed by the fused step
        super().zero_grad(set_to_none=set_to_none)
