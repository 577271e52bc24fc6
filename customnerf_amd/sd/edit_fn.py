"""Autograd Functions over csrc/edit_ops.hip: what lies between the kernels of one editing step (the ray buffer -> NCHW images, the background
L1 term, the VAE's posterior sample, the SDS loss) as one launch per direction each, instead of chains of framework element-wise ops
(round 5: 67 -> <= 10 framework launches per step, profiles/r05_edit_step_kernels_*.txt).  Semantics are the torch expressions they replace,
cited per class; tests/test_gpu_sd_editing.py holds both side by side."""
from torch.autograd import Function

from . import ops


class RayImages(Function):
    """out_ray [3, B*H*W, 6] -> (pred_rgb, pred_rgb_fg, pred_rgb_bg), each [B, 3, H, W] float32:
    `outputs[...]['image'].reshape(B, H, W, 3).permute(0, 3, 1, 2).contiguous()` of utils_init_nerf.py:361-366, three times."""

    @staticmethod
    def forward(ctx, out_ray, B, H, W):
        ctx.set_materialize_grads(False)
        ctx.dims = (B, H, W, out_ray.device)
        return tuple(ops.ray_images(out_ray, B, H, W))

    @staticmethod
    def backward(ctx, d_all, d_fg, d_bg):
        B, H, W, dev = ctx.dims
        return ops.ray_images_backward((d_all, d_fg, d_bg), B, H, W, dev), None, None, None


class ScaledL1(Function):
    """scale * F.l1_loss(a, b) (utils_init_nerf.py:389-391: `opt.keep_bg * F.l1_loss(pt_rgb_bg, pred_rgb_bg)`), differentiable in both."""

    @staticmethod
    def forward(ctx, a, b, scale):
        loss, dsign = ops.l1_loss_scaled(a, b, scale)
        ctx.save_for_backward(dsign)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        dsign, = ctx.saved_tensors
        da = ops.scale_by_scalar(dsign, g, 1.0) if ctx.needs_input_grad[0] else None
        db = ops.scale_by_scalar(dsign, g, -1.0) if ctx.needs_input_grad[1] else None
        return da, db, None


class SDSLoss(Function):
    """sd.py:150-152: `target = (latents - grad).detach(); loss = 0.5 * F.mse_loss(latents, target, reduction='sum')` — a scalar whose
    gradient with respect to the latents is `grad` (as the subtraction rounds it)."""

    @staticmethod
    def forward(ctx, latents, grad):
        loss, diff2 = ops.sds_loss(latents, grad)
        ctx.save_for_backward(diff2)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        diff2, = ctx.saved_tensors                             # 2 (latents - target): torch's order of operations, overflow included
        return ops.scale_by_scalar(diff2, g, 0.5), None


class SampleLatents(Function):
    """sd.py:102-104 on the encoder's moments [B, h, w, 8] half: `posterior.sample() * 0.18215` with the caller's noise -> [B, 4, h, w] float32"""

    @staticmethod
    def forward(ctx, moments, noise, scaling_factor):
        ctx.save_for_backward(moments, noise)
        ctx.sf = float(scaling_factor)
        return ops.sample_latents(moments, noise, scaling_factor)

    @staticmethod
    def backward(ctx, d_lat):
        moments, noise = ctx.saved_tensors
        return ops.sample_latents_backward(moments, noise, d_lat, ctx.sf), None, None
