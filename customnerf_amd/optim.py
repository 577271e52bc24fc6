"""Optimiser step of the reference's recipe (main.py:182: Adam betas (0.9, 0.99), eps 1e-15, grid lr x10 via
NeRFNetwork.get_params; main.py:189: lr * 0.1^(iter/iters)) as one fused HIP launch per parameter tensor:
un-scale + moments + update + fp16 shadow refresh + gradient zeroing."""
import torch

import ctypes

from ._lib import lib, check, ptr, stream, require_cuda, AdamJobs, ADAM_MAX_JOBS, GridAdam

SMALL_PARAM_MAX = 1 << 16          # tensors up to this size go through the single-workgroup multi-tensor launch (the three MLPs: 22.5 k floats)


def adam_step(p, g, m, v, lr, betas=(0.9, 0.99), eps=1e-15, step=1, grad_scale_inv=1.0, zero_grad=True, p_half=None):
    require_cuda(p, g, m, v, p_half)
    check(lib.cnerf_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), ptr(p_half), p.numel(), float(lr), float(betas[0]), float(betas[1]),
                              float(eps), int(step), float(grad_scale_inv), int(zero_grad), stream()), "adam_step")


class DynamicLossScaler:
    """torch.cuda.amp.GradScaler's policy (the reference trainer's fp16 recipe: init 65536, x2 every 2000 clean steps, x0.5 and a
    skipped optimiser step on inf/NaN) with the whole state on the device: `state` = float32[4] {scale, growth_tracker, found_inf,
    good_steps}.  Nothing in a step reads it back, so the step stays free of host syncs (GradScaler.step() does an .item())."""

    def __init__(self, device, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        self.state = torch.tensor([init_scale, 0.0, 0.0, 0.0], dtype=torch.float32, device=device)
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval
        self._update_consumed = False       # FusedAdam.step() already applied this step's update() in its last launch

    def scale(self, loss):
        return loss * self.state[0]

    def backward(self, loss):
        """scale(loss).backward() without the multiply node: the scale seeds the backward pass as d(scaled loss)/d(loss)
        (three launches fewer per step: the multiply, the ones_like seed, the multiply's backward)"""
        loss.backward(gradient=self.state[0].to(loss.dtype).reshape(loss.shape))

    def check(self, flat_grads):
        """found_inf |= any non-finite value (call on the all-reduced flat gradient buffer: every rank then takes the same decision)"""
        require_cuda(flat_grads)
        check(lib.cnerf_scaler_check(ptr(flat_grads), flat_grads.numel(), ptr(self.state), stream()), "scaler_check")

    def watch(self, on=True):
        """cnerf_scaler_watch: while watched, the field backward and the grid scatter raise found_inf themselves when they produce a non-finite
        gradient — a trainer whose gradients all come from those two (trainer.inf_check_is_folded) then skips check().  One watched scaler
        per process: the trainers call this at the start of every step."""
        check(lib.cnerf_scaler_watch(ptr(self.state) if on else None), "scaler_watch")

    def update(self):
        """GradScaler.update().  A no-op for the step whose FusedAdam.step() already applied it in the tail of its multi-tensor launch, so the
        standard idiom `opt.step(); scaler.update()` (the reference's Trainer, the drop-in flows) counts every step exactly once."""
        if self._update_consumed:
            self._update_consumed = False
            return
        check(lib.cnerf_scaler_update(ptr(self.state), float(self.growth_factor), float(self.backoff_factor), int(self.growth_interval), stream()), "scaler_update")

    def get_scale(self):
        return float(self.state[0])                                    # host read: diagnostics only

    def good_steps(self):
        return int(self.state[3])

    def state_dict(self):
        """torch.cuda.amp.GradScaler.state_dict() keys (what the reference checkpoints under 'scaler', utils_init_nerf.py:797-798),
        plus the count of non-skipped optimiser steps that drives Adam's bias correction here."""
        st = self.state.detach().cpu().tolist()
        return {"scale": st[0], "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor, "growth_interval": self.growth_interval,
                "_growth_tracker": int(st[1]), "_good_steps": int(st[3])}

    def load_state_dict(self, sd):
        good = sd.get("_good_steps", int(self.state[3]))
        self.state.copy_(torch.tensor([float(sd["scale"]), float(sd.get("_growth_tracker", 0)), 0.0, float(good)]))
        self.growth_factor, self.backoff_factor, self.growth_interval = sd["growth_factor"], sd["backoff_factor"], sd["growth_interval"]


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam-compatible surface (param groups with per-group lr) over cnerf_adam_step.
    `half_shadows`: {param: fp16 tensor} refreshed in the same pass (GridEncoder.set_half_table)."""

    def __init__(self, params, lr=5e-4, betas=(0.9, 0.99), eps=1e-15, zero_grad_in_step=True):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.zero_grad_in_step = zero_grad_in_step
        self.half_shadows = {}
        self.grad_scale_inv = 1.0
        self.scaler = None              # DynamicLossScaler: gradient scale, skip decision and step count come from its device state
        self.skip_params = set()        # id(p) of parameters some other owner updates (dp.ShardedExchange: the sharded grid table)
        self.updates_scaler = False     # set by step(): the last launch of this step also ran GradScaler.update() (the caller must not run it again)

    # ---- the grid table's step inside the backward scatter (cnerf_grid_backward_adam) -------------------------------------------------------
    def arm_in_backward(self, p):
        """Arm THIS step's update of parameter `p` (the grid table) for the backward pass that is about to run: the scatter's flush applies it where
        the gradient becomes final, and step() then leaves `p` alone (bit-identical parameters; the table's own Adam launch — 62 us at the HBM roofline
        for the benchmark table — disappears).  Needs the on-device loss scaler, a persistent contiguous float32 `p.grad` (the trainers'
        flat_grad_buffer) and the current learning rate already in the parameter group.  The caller guarantees that the coming backward pass is the
        only contribution to `p.grad` in this optimiser step (ReconTrainer: one render, one backward) and that nothing but the fused field's own
        kernels can raise found_inf (trainer.inf_check_is_folded).  -> True if armed.  step() disarms whatever happened."""
        self.disarm_in_backward()
        if self.scaler is None or p.grad is None or not (p.is_cuda and p.is_contiguous() and p.grad.is_contiguous() and p.dtype == torch.float32
                                                           and p.grad.dtype == torch.float32 and p.numel() % 4 == 0):
            return False
        group = next((g for g in self.param_groups if any(q is p for q in g['params'])), None)
        if group is None or id(p) in self.skip_params:
            return False
        st = self.state[p]
        if not st:
            st['step'] = 0
            st['exp_avg'] = torch.zeros_like(p)
            st['exp_avg_sq'] = torch.zeros_like(p)
        sh = self.half_shadows.get(p)
        require_cuda(p.data, p.grad, st['exp_avg'], st['exp_avg_sq'], sh, self.scaler.state)
        cfg = GridAdam()
        cfg.p, cfg.g, cfg.m, cfg.v = p.data.data_ptr(), p.grad.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr()
        cfg.p_half = sh.data_ptr() if sh is not None else None
        cfg.n = p.numel()
        cfg.lr, cfg.beta1, cfg.beta2, cfg.eps = float(group['lr']), float(group['betas'][0]), float(group['betas'][1]), float(group['eps'])
        cfg.scaler_state = self.scaler.state.data_ptr()
        cfg.extra_inv = float(self.grad_scale_inv)
        cfg.zero_grad = int(self.zero_grad_in_step)
        check(lib.cnerf_grid_backward_adam(ctypes.addressof(cfg)), "grid_backward_adam")
        self._armed = p
        return True

    def disarm_in_backward(self):
        """-> the parameter whose armed step a backward pass applied (step() must skip it), or None; nothing stays armed"""
        p, self._armed = getattr(self, '_armed', None), None
        if p is None:
            return None
        done = ctypes.c_int(0)
        check(lib.cnerf_grid_backward_adam_consumed(ctypes.addressof(done)), "grid_backward_adam_consumed")
        check(lib.cnerf_grid_backward_adam(None), "grid_backward_adam")
        return p if done.value else None

    @torch.no_grad()
    def step(self, closure=None):
        self.updates_scaler = False
        if self.scaler is not None:
            self.scaler._update_consumed = False      # (a caller that skipped update() after the previous step, as the trainers used to)
        stepped = self.disarm_in_backward()             # the table, when the backward scatter already applied its update (arm_in_backward)
        if stepped is not None:
            self.state[stepped]['step'] += 1
            stepped._cnerf_epoch = getattr(stepped, '_cnerf_epoch', 0) + 1
        small = []                      # (p, state, group) of the small tensors: one multi-tensor launch at the end (with a scaler)
        betas0, eps0 = self.param_groups[0]['betas'], self.param_groups[0]['eps']
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is None or id(p) in self.skip_params or p is stepped:
                    continue
                st = self.state[p]
                if not st:
                    st['step'] = 0
                    st['exp_avg'] = torch.zeros_like(p)
                    st['exp_avg_sq'] = torch.zeros_like(p)
                st['step'] += 1
                if (self.scaler is not None and p.numel() <= SMALL_PARAM_MAX and len(small) < ADAM_MAX_JOBS and group['betas'] == betas0
                        and group['eps'] == eps0 and p.grad.is_contiguous() and p.is_contiguous()):
                    small.append((p, st, group))
                    continue
                if self.scaler is not None:
                    require_cuda(p.data, p.grad, st['exp_avg'], st['exp_avg_sq'], self.half_shadows.get(p))
                    check(lib.cnerf_adam_step_scaled(ptr(p.data), ptr(p.grad), ptr(st['exp_avg']), ptr(st['exp_avg_sq']), ptr(self.half_shadows.get(p)),
                                                     p.numel(), float(group['lr']), float(group['betas'][0]), float(group['betas'][1]), float(group['eps']),
                                                     ptr(self.scaler.state), float(self.grad_scale_inv), int(self.zero_grad_in_step), stream()), "adam_step_scaled")
                else:
                    adam_step(p.data, p.grad, st['exp_avg'], st['exp_avg_sq'], group['lr'], group['betas'], group['eps'], st['step'],
                              self.grad_scale_inv, self.zero_grad_in_step, self.half_shadows.get(p))
                # the kernel writes through the raw pointer, which does not bump torch's version counter:
                # advance our own epoch so caches keyed on the parameter (GridEncoder.half_table) notice.
                p._cnerf_epoch = getattr(p, '_cnerf_epoch', 0) + 1
        if small:
            # the small tensors in ONE single-workgroup launch, which — being the last Adam launch of the step — also applies the scaler's
            # update (found_inf ? back off : count the step, grow every `growth_interval`): two or three launches and the update launch saved
            jobs = AdamJobs()
            for j, (p, st, group) in enumerate(small):
                sh = self.half_shadows.get(p)
                require_cuda(p.data, p.grad, st['exp_avg'], st['exp_avg_sq'], sh)
                jobs.p[j], jobs.g[j], jobs.m[j], jobs.v[j] = p.data.data_ptr(), p.grad.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr()
                jobs.p_half[j] = sh.data_ptr() if sh is not None else None
                jobs.n[j], jobs.lr[j] = p.numel(), float(group['lr'])
                p._cnerf_epoch = getattr(p, '_cnerf_epoch', 0) + 1
            jobs.n_jobs = len(small)
            sc = self.scaler
            check(lib.cnerf_adam_step_scaled_multi(ctypes.addressof(jobs), float(betas0[0]), float(betas0[1]), float(eps0), ptr(sc.state),
                                                   float(self.grad_scale_inv), int(self.zero_grad_in_step), 1, float(sc.growth_factor),
                                                   float(sc.backoff_factor), int(sc.growth_interval), stream()), "adam_step_scaled_multi")
            self.updates_scaler = True
            sc._update_consumed = True                # ... so the caller's scaler.update() for this step does nothing

    def zero_grad(self, set_to_none=False):
        if self.zero_grad_in_step and not set_to_none:
            return                      # gradients were zeroed by the fused step
        super().zero_grad(set_to_none=set_to_none)
