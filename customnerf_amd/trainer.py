"""Minimal counterpart of the reference trainer's reconstruction step (nerf/utils_init_nerf.py:194-241
`train_step_pretrain`, :599-629 the loop body) with the reference's optimiser recipe (main.py:182-189) and the
data-parallel layer the reference lacks: ray/view sharding across ranks, one RCCL all-reduce of
the flattened gradients per step.
"""
import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import _coll
from .optim import DynamicLossScaler, FusedAdam


def flat_grad_buffer(parameters):
    """Give every trainable parameter a persistent, zeroed `.grad` that is a view into ONE flat fp32 buffer (parameter order), so that
    the per-step all-reduce runs in place on that buffer: no gather / scatter copies around the collective.  autograd accumulates into an
    existing `.grad` in place and the fused Adam step zeroes it in place, so the views stay bound."""
    ps = [p for p in parameters if p.requires_grad]
    if not ps:
        return None
    flat = torch.zeros(sum(_slot(p.numel()) for p in ps), dtype=torch.float32, device=ps[0].device)
    off = 0
    for p in ps:
        p.grad = flat[off:off + p.numel()].view_as(p)
        off += _slot(p.numel())
    return flat


def enable_grad_in_place(model):
    """With persistent .grad views (flat_grad_buffer) the grid scatter may add straight into embeddings.grad (GridEncoder.grad_in_place):
    autograd then has nothing to accumulate for the table."""
    enc = getattr(model, 'pos_en', None)
    if enc is not None and hasattr(enc, 'attach_backward'):
        enc.grad_in_place = True
    model.grad_in_place = True                          # the fused field's three parameter vectors (NeRFNetwork.split_forward)


def _slot(n):
    return (n + 3) // 4 * 4            # every view starts 16-byte aligned (the fused Adam kernel loads float4); pad elements stay zero


def _grads_alias_flat(grads, flat):
    off = flat.data_ptr()
    for g in grads:
        if g.dtype != flat.dtype or not g.is_contiguous() or g.data_ptr() != off:
            return False
        off += _slot(g.numel()) * flat.element_size()
    return off == flat.data_ptr() + flat.numel() * flat.element_size()


def allreduce_grads_flat(parameters, flat, world_size):
    """One flattened all-reduce (sum) of all gradients (RCCL over xGMI on GPUs, gloo in the CPU tests); returns the flat buffer
    for reuse.  In place when the gradients are views of `flat` (flat_grad_buffer), through a staging copy otherwise.  The 1/world
    factor is NOT applied here (folded into the Adam un-scale).  Shared by ReconTrainer and EditTrainer."""
    if world_size <= 1:
        return flat
    grads = [p.grad for p in parameters if p.grad is not None]
    if flat is not None and _grads_alias_flat(grads, flat):
        _coll.all_reduce(flat, op=dist.ReduceOp.SUM)
        return flat
    n = sum(g.numel() for g in grads)
    if flat is None or flat.numel() != n:
        flat = torch.empty(n, dtype=torch.float32, device=grads[0].device)       # staging buffer (the caller keeps it for the next step)
    torch._foreach_copy_(list(flat.split([g.numel() for g in grads])), [g.reshape(-1) for g in grads])
    _coll.all_reduce(flat, op=dist.ReduceOp.SUM)
    torch._foreach_copy_([g.reshape(-1) for g in grads], list(flat.split([g.numel() for g in grads])))
    return flat


def register_half_shadow(optimizer, model, fp16):
    """Let the fused Adam refresh the grid table's fp16 shadow in its own pass (no separate cast launch per step)."""
    enc = getattr(model, 'pos_en', None)
    if fp16 and enc is not None and hasattr(enc, 'half_table') and enc.embeddings.is_cuda and enc.level_dim % 2 == 0:
        optimizer.half_shadows[enc.embeddings] = enc.half_table()


def refresh_half_shadow(optimizer, model):
    enc = getattr(model, 'pos_en', None)
    if enc is not None and enc.embeddings in optimizer.half_shadows:
        enc.set_half_table(optimizer.half_shadows[enc.embeddings])   # the step just rewrote it: mark it current


class packed_weights_window:
    """`with packed_weights_window(model):` — for the duration of a training step the fused field kernels read the packed fp16 image of the MLP
    parameters (NeRFNetwork._weight_image / field.packed_weights: one pack launch per parameter version instead of 14 us of staging in each
    of the step's three field launches).  Only inside the step: there every parameter write goes through the trainer's optimiser, which moves
    the version counters the image is keyed on; outside (evaluation under torch_ema's `.data` swaps, user code) the kernels stage from the
    float32 parameters.  `opt.packed_field_weights = False` switches it off."""

    def __init__(self, model, opt=None):
        self.d = getattr(model, '__dict__', None) if getattr(opt, 'packed_field_weights', True) else None

    def __enter__(self):
        if self.d is not None:
            self.prev = self.d.get('packed_field_weights', False)
            self.d['packed_field_weights'] = True

    def __exit__(self, *exc):
        if self.d is not None:
            self.d['packed_field_weights'] = self.prev
        return False


def inf_check_is_folded(trainer):
    """Round 6: GradScaler's inf check is a pass over every gradient (9 us for the benchmark table, 30 us for the reference
    field's).  With the scaler WATCHED (DynamicLossScaler.watch) the two kernels that produce the gradients raise found_inf themselves — the field
    backward's partial reduction for the three MLPs, the scatter's emit for the table (a non-finite incoming gradient is the only way its exact
    fixed-point sums can go non-finite) — so the pass is skipped when EVERY trainable parameter is one of those four tensors of a fused
    half-precision NeRFNetwork and no gradient exchange runs in between (the exchange checks its own reduced shards)."""
    m = trainer.model
    if trainer.world_size != 1 or getattr(trainer, '_dp', None) is not None or trainer.scaler is None or not getattr(trainer, 'fused_adam', True):
        return False
    if not (hasattr(m, '_fused_cfg') and hasattr(m, '_half') and hasattr(m, 'pos_en') and getattr(m, 'grad_in_place', False)):
        return False
    try:
        if not (m._fused_cfg() and m._half()):
            return False
        known = {id(m.pos_en.embeddings), id(m.network.params), id(m.density_network.params), id(m.rgb_network.params)}
    except Exception:
        return False
    return all(id(p) in known for p in m.parameters() if p.requires_grad) and bool(getattr(trainer.opt, 'fold_inf_check', True))


def check_grads_finite(scaler, parameters, flat):
    """GradScaler's inf check on the (all-reduced) gradients: one pass over the flat buffer when the gradients alias it."""
    grads = [p.grad for p in parameters if p.grad is not None]
    if flat is not None and _grads_alias_flat(grads, flat):
        scaler.check(flat)
    else:
        for g in grads:
            scaler.check(g.contiguous())


def setup_sharded_dp(trainer, model, fp16, rank=None, group=None):
    """dp_mode 'sharded' (customnerf_amd.dp.ShardedExchange): the big parameters (the grid table) leave the trainer's FusedAdam — their owner
    shards are updated by the exchange (and, only with `dp.async_ops`, the MLP all-reduce is hooked in front of the grid scatter)."""
    from .dp import ShardedExchange, BIG_PARAM_MIN
    opt_ = trainer.optimizer
    lr_by_id = {id(p): i for i, g in enumerate(opt_.param_groups) for p in g['params']}
    params = [p for p in model.parameters() if p.requires_grad]
    enc = getattr(model, 'pos_en', None)
    half = bool(fp16 and enc is not None and hasattr(enc, 'half_table') and enc.level_dim % 2 == 0)
    rank = dist.get_rank(group) if rank is None else rank
    dp = ShardedExchange(params, trainer._flat, lambda p: trainer.base_lrs[lr_by_id[id(p)]], trainer.world_size, rank, betas=(0.9, 0.99), eps=1e-15,
                         scaler=trainer.scaler, half_shadow=half, group=group)
    opt_.skip_params = {id(st['p']) for st in dp.state}
    if half and enc is not None:
        sh = dp.shadow_table(enc.embeddings)
        if sh is not None:
            opt_.half_shadows[enc.embeddings] = sh                 # refresh_half_shadow hands it to the encoder after every step
            enc.set_half_table(sh)
    if dp.async_ops and enc is not None and hasattr(enc, 'attach_backward'):
        from .gridencoder import grid as ge
        ge.set_pre_scatter_hook(dp.start_small)                    # MLP gradients go out while the grid scatter still runs
    return dp


def apply_optimizer_step(trainer):
    """gradient exchange + GradScaler check + Adam + scaler update + fp16 shadow hand-over, for ReconTrainer and EditTrainer (fused Adam path).
    world_size 1: no exchange; dp_mode 'allreduce': one in-place fp32 all-reduce of the flat buffer; 'sharded': dp.ShardedExchange."""
    f = trainer.lr_factor()
    for g, base in zip(trainer.optimizer.param_groups, trainer.base_lrs):
        g['lr'] = base * f
    dp = getattr(trainer, '_dp', None)
    trainer.optimizer.grad_scale_inv = 1.0 / (trainer.loss_scale * trainer.world_size)
    if dp is not None:
        ev = getattr(trainer, '_exchange_events', None)              # bench.py: event pairs around the exchange (None = off)
        if ev is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        dp.exchange()
        dp.check()
        dp.step(f, trainer.loss_scale)
        if ev is not None:
            e1.record()
            ev.append((e0, e1))
        trainer.optimizer.step()                                     # the small (replicated) parameters; the big ones are in skip_params
        if trainer.scaler is not None:
            trainer.scaler.update()                                  # (a no-op when FusedAdam's multi-tensor launch of the small tensors already did it)
    else:
        trainer.allreduce_grads()
        if trainer.scaler is not None:
            if not getattr(trainer, '_inf_folded', False):
                check_grads_finite(trainer.scaler, list(trainer.model.parameters()), trainer._flat)  # on the all-reduced gradients: every rank takes the same skip decision
            trainer.optimizer.step()
            trainer.scaler.update()                                  # (a no-op when FusedAdam's multi-tensor launch of the small tensors already did it)
        else:
            trainer.optimizer.step()
    refresh_half_shadow(trainer.optimizer, trainer.model)


class CheckpointMixin:
    """Checkpoints of a (possibly sharded data-parallel) trainer.  In dp_mode 'sharded' with the fp16 shadow the optimiser updates the owners'
    float32 master SHARDS and the shadow table only; the float32 nn.Parameter is refreshed here (ShardedExchange.consolidate — a collective, so
    EVERY rank calls these methods; rank 0 writes the file).  checkpoint.checkpoint_state refuses a model whose parameters are marked stale."""

    def state_dict(self):
        if getattr(self, '_dp', None) is not None:
            self._dp.consolidate()
        return self.model.state_dict()

    def save_checkpoint(self, path, epoch=0, stats=None, full=False, rank=None):
        from . import checkpoint
        dp = getattr(self, '_dp', None)
        if dp is not None:
            if full:
                dp.export_optimizer_state(self.optimizer)         # consolidates + gathers the sharded Adam moments into the unsharded layout
            else:
                dp.consolidate()
        if rank is None:
            rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
        if rank != 0:
            return None
        return checkpoint.save_checkpoint(path, self.model, epoch=epoch, global_step=self.global_step, stats=stats, optimizer=self.optimizer, full=full,
                                          scaler=self.scaler)


class EvalMixin:
    """Trainer_Nerf.eval_step / test_step (nerf/utils_init_nerf.py:396-434, 436-485): one full view rendered without perturbation and the image
    panels the reference's evaluate_one_epoch / test loops write out.  The reference passes `**vars(self.opt)` to render(); the sampling keys
    run() / run_cuda() read are forwarded here.  (Its render() ignores `staged` — renderer.py:1719-1733 — as this package's does.)"""

    def _render_eval(self, model, rays_o, rays_d, perturb=False):
        opt = self.opt
        kw = dict(num_steps=opt.num_steps, upsample_steps=opt.upsample_steps, dt_gamma=getattr(opt, 'dt_gamma', 0), max_steps=opt.max_steps)
        with torch.no_grad(), torch.autocast('cuda', dtype=torch.float16, enabled=self.fp16):
            return model.render(rays_o, rays_d, staged=True, perturb=perturb, force_all_rays=True, **kw)

    def eval_step(self, data):
        """-> (panel [B, H, n W, 3] = gt | prediction | depth (| gt mask | predicted mask | foreground | background under opt.train_conf), None, None,
        loss = tensor([0])), as utils_init_nerf.py:396-434"""
        rgbs, mask, rays_o, rays_d, H, W, _ = data
        dev = next(self.model.parameters()).device
        rgbs, mask, rays_o, rays_d = rgbs.to(dev), mask.to(dev), rays_o.to(dev), rays_d.to(dev)
        B = rays_o.shape[0]
        outputs = self._render_eval(self.model, rays_o, rays_d)
        ims = [rgbs.reshape(B, H, W, 3).float(), outputs['image'].reshape(B, H, W, 3).float(), outputs['depth'].reshape(B, H, W, 1).float().repeat(1, 1, 1, 3)]
        if getattr(self.opt, 'train_conf', 0):
            ims += [mask.reshape(B, H, W, 1).float().repeat(1, 1, 1, 3),
                    outputs['render_mask'].reshape(B, H, W, -1).float().mean(-1, keepdim=True).repeat(1, 1, 1, 3),
                    outputs['fg']['image'].reshape(B, H, W, 3).float(), outputs['bg']['image'].reshape(B, H, W, 3).float()]
        return torch.cat(ims, dim=2), None, None, torch.tensor([0])

    def test_step(self, data, bg_color=None, perturb=False, if_gui=False):
        """-> (pred_rgb [B, H, W, 3] — the panel prediction | mask | foreground | background under opt.train_conf and opt.render_all —,
        pred_depth [B, H, W], dir), as utils_init_nerf.py:436-485.  (The reference also renders model_pretrained there and drops the result.)"""
        direction = torch.tensor(0)
        if if_gui:
            rays_o, rays_d, H, W, direction = data['rays_o'], data['rays_d'], data['H'], data['W'], data['dir']
        else:
            _, _, rays_o, rays_d, H, W, _ = data
        dev = next(self.model.parameters()).device
        rays_o, rays_d = rays_o.to(dev), rays_d.to(dev)
        B = rays_o.shape[0]
        outputs = self._render_eval(self.model, rays_o, rays_d, perturb=perturb)
        pred_rgb = outputs['image'].reshape(B, H, W, 3).float()
        pred_depth = outputs['depth'].reshape(B, H, W).float()
        if getattr(self.opt, 'train_conf', 0) and getattr(self.opt, 'render_all', False):
            pred_rgb = torch.cat([pred_rgb, outputs['render_mask'].reshape(B, H, W, 1).float().repeat(1, 1, 1, 3),
                                  outputs['fg']['image'].reshape(B, H, W, 3).float(), outputs['bg']['image'].reshape(B, H, W, 3).float()], dim=2)
        return pred_rgb, pred_depth, direction


class ReconTrainer(CheckpointMixin, EvalMixin):
    def __init__(self, model, opt, lr=None, fp16=False, world_size=1, fused_adam=True, loss_scale='dynamic', dp_mode='allreduce'):
        """dp_mode (world_size > 1): 'allreduce' = one in-place fp32 all-reduce of the flat gradient buffer; 'sharded' = customnerf_amd.dp.ShardedExchange
        (fp16 all-to-all payload summed in fp32 on arrival, sharded Adam, all-gather of the fp16 shadow; MLP groups all-reduced in fp32).
        loss_scale: 'dynamic' = the reference's GradScaler policy (utils_init_nerf.py `self.scaler = GradScaler(enabled=self.fp16)`:
        init 65536, x2 / 2000 clean steps, x0.5 + skipped step on inf) kept on the device (optim.DynamicLossScaler, fused Adam only);
        a float = static scale."""
        self.model, self.opt = model, opt
        self.fp16 = fp16
        self.world_size = world_size
        self.dp_mode = dp_mode if world_size > 1 else 'none'
        if loss_scale == 'dynamic' and not (fp16 and fused_adam):
            loss_scale = 128.0
        self.scaler = DynamicLossScaler(next(model.parameters()).device) if loss_scale == 'dynamic' else None
        self.loss_scale = 1.0 if (not fp16 or self.scaler is not None) else float(loss_scale)
        lr = opt.lr if lr is None else lr
        groups = model.get_params(lr)                        # grid lr x10 (network_grid.py:196-206)
        self.base_lrs = [g['lr'] for g in groups]
        if fused_adam:
            self.optimizer = FusedAdam(groups, betas=(0.9, 0.99), eps=1e-15)
            self.optimizer.scaler = self.scaler
            register_half_shadow(self.optimizer, model, fp16)
        else:
            self.optimizer = torch.optim.Adam(groups, betas=(0.9, 0.99), eps=1e-15)
        self.fused_adam = fused_adam
        self.global_step = 0
        self._flat = flat_grad_buffer(self.model.parameters())   # persistent, pre-zeroed .grad views of one flat buffer (zeroed by the fused step)
        if fused_adam:
            enable_grad_in_place(model)                          # (torch.optim.Adam + zero_grad keeps the views too, but stay on the plain path there)
        self._dp = setup_sharded_dp(self, model, fp16) if (self.dp_mode == 'sharded' and fused_adam) else None

    def lr_factor(self):
        return 0.1 ** min(self.global_step / self.opt.iters, 1)      # main.py:189

    def dp_describe(self):
        """one-line description of the gradient exchange (bench.py's config.parallelism)"""
        if self.dp_mode == 'sharded' and getattr(self, '_dp', None) is not None:
            return self._dp.describe()
        return "fp32 grad all-reduce, in place on the flat buffer"

    def loss(self, outputs, rgbs, mask):
        """utils_init_nerf.py:220-234"""
        if '_out_ray' in outputs and outputs['_out_ray'].is_cuda and getattr(self.opt, 'fused_loss', True):
            from .nerf.render_ops import recon_loss                   # loss + gradient in one launch on the fused renderer's composite output
            with_mask = bool(getattr(self.opt, 'train_conf', 0)) and 'render_mask' in outputs
            sc = getattr(self, 'scaler', None)                        # the backward seed (DynamicLossScaler.backward) folded into the loss gradient
            return recon_loss(outputs['_out_ray'], rgbs, mask if with_mask else None, getattr(self.opt, 'train_rgb', 1.0),
                              getattr(self.opt, 'train_conf', 0) if with_mask else 0.0, sc.state[0:1] if sc is not None else None)
        pred_rgb = outputs['image']
        loss = getattr(self.opt, 'train_rgb', 1.0) * F.mse_loss(pred_rgb.reshape(-1, 3).float(), rgbs.reshape(-1, 3))
        if getattr(self.opt, 'train_conf', 0) and 'render_mask' in outputs:
            loss = loss + self.opt.train_conf * F.mse_loss(outputs['render_mask'].reshape(-1).float(), mask.reshape(-1))
        return loss

    def allreduce_grads(self):
        """One flattened RCCL all-reduce (sum) per step; the 1/world factor is folded into the Adam un-scale."""
        self._flat = allreduce_grads_flat(list(self.model.parameters()), self._flat, self.world_size)

    def select_rays(self, rays_o, rays_d, rgbs, mask, select_inds=None):
        """`--batch_rays` (utils_init_nerf.py:210-215): a random subset of `opt.batch_rays` rays of the view, drawn without replacement by
        numpy on the host as the reference does (`np.random.choice`; `select_inds` replays a recorded draw), gathered on the device."""
        n = int(getattr(self.opt, 'batch_rays', 0) or 0)
        if not n and select_inds is None:
            return rays_o, rays_d, rgbs, mask
        N = rays_o.reshape(-1, 3).shape[0]
        if select_inds is None:
            import numpy as np
            select_inds = np.random.choice(N, size=[n], replace=False)
        idx = torch.as_tensor(select_inds, dtype=torch.long).to(rays_o.device)
        pick = lambda t, c: t.reshape(1, N, c).index_select(1, idx) if t is not None else None
        return pick(rays_o, 3), pick(rays_d, 3), pick(rgbs, 3), pick(mask, 1)

    def _watch_scaler(self):
        """at the start of every step: is the inf check folded into the gradient producers (inf_check_is_folded), and if so point them at this scaler"""
        self._inf_folded = inf_check_is_folded(self)
        if self.scaler is not None:
            self.scaler.watch(self._inf_folded)

    def train_step(self, rays_o, rays_d, rgbs, mask, select_inds=None, **render_kw):
        self._watch_scaler()
        try:
            with packed_weights_window(self.model, self.opt):
                return self._train_step(rays_o, rays_d, rgbs, mask, select_inds, **render_kw)
        finally:
            if self.scaler is not None:
                self.scaler.watch(False)                 # (the library holds a raw pointer into the scaler's state only for the duration of a step)

    def _arm_table_step(self):
        """The grid table's Adam update inside the backward scatter (optim.FusedAdam.arm_in_backward): only where this step's ONE backward pass is the
        table gradient's only source and found_inf is final when the scatter runs — fused Adam with the on-device scaler, the inf check folded into
        the field's gradient producers (inf_check_is_folded: every trainable parameter is the fused field's, persistent in-place gradients), one
        GPU, eager (a captured backward would freeze the learning rate).  `opt.fuse_table_adam = False` switches it off."""
        if not (self.fused_adam and getattr(self, '_inf_folded', False) and self.world_size == 1 and getattr(self, '_dp', None) is None
                and getattr(self.opt, 'fuse_table_adam', True) and not torch.cuda.is_current_stream_capturing()):
            return False
        f = self.lr_factor()                                          # (apply_optimizer_step sets the same values again after the backward pass)
        for g, base in zip(self.optimizer.param_groups, self.base_lrs):
            g['lr'] = base * f
        self.optimizer.grad_scale_inv = 1.0 / (self.loss_scale * self.world_size)
        return self.optimizer.arm_in_backward(self.model.pos_en.embeddings)

    def _train_step(self, rays_o, rays_d, rgbs, mask, select_inds=None, **render_kw):
        self.model.train()
        rays_o, rays_d, rgbs, mask = self.select_rays(rays_o, rays_d, rgbs, mask, select_inds)
        render_kw.setdefault('fg_bg', getattr(self, 'needs_fg_bg', False))       # the reconstruction loss reads the first composite only
        with torch.autocast('cuda', dtype=torch.float16, enabled=self.fp16):
            outputs = self.model.render(rays_o, rays_d, staged=False, perturb=True, force_all_rays=True, **render_kw)
            loss = self.loss(outputs, rgbs, mask)
        if self.scaler is not None:
            self._arm_table_step()
            try:
                self.scaler.backward(loss)
            except BaseException:
                self.optimizer.disarm_in_backward()
                raise
        else:
            (loss * self.loss_scale).backward()
        if self.fused_adam:
            apply_optimizer_step(self)
        else:
            self.allreduce_grads()
            f = self.lr_factor()
            for g, base in zip(self.optimizer.param_groups, self.base_lrs):
                g['lr'] = base * f
            inv = 1.0 / (self.loss_scale * self.world_size)
            if inv != 1.0:
                torch._foreach_mul_([p.grad for p in self.model.parameters() if p.grad is not None], inv)
            self.optimizer.step()
            self.optimizer.zero_grad(set_to_none=False)
        self.global_step += 1
        return loss.detach(), outputs

    def train_step_graphed(self, rays_o, rays_d, rgbs, mask, **render_kw):
        """Buffer ownership: everything a captured graph reads is either one of the view's four resident tensors (pinned by the cache entry), a
        parameter / its .grad / optimiser state (never reallocated), an allocation made during capture (graph pool), or a cached scratch buffer
        — and the cache is flushed whenever one of those is reallocated (_lib.scratch_generation).

        train_step with render + loss + backward replayed as ONE hipGraph per view (opt-in).  A view is recognised by the addresses of its four
        resident tensors (a dataset keeps its rays on the GPU and hands the same tensors out every epoch); the first visit runs two eager steps
        (allocations, workspaces) and captures the third, later visits replay it.  The optimiser step stays eager: its learning rate is a host-side
        scalar that changes every step.  The GPU time is the eager step's (the step is GPU-bound: DESIGN.md section 6); what the graph buys is the
        host — 0.18 instead of 0.79 ms of enqueue per step (scratch/graph_probe.py) — i.e. immunity against a slow or shared host.
        Fused Adam with the on-device loss scaler, no --batch_rays subsampling; data-parallel trainers too (round 6): the gradient exchange and
        the optimiser step run eagerly after the replay — what matters under `--scaling strong`, where a rank's 2048-ray share of a view is 0.5 ms
        of kernels behind 1 ms of eager enqueue (profiles/r06_batch_sweep.json).  -> (loss, None): the loss is a static tensor that the
        next replay of the same view overwrites; the per-ray outputs stay inside the graph's memory pool."""
        if not (self.fused_adam and self.scaler is not None) or int(getattr(self.opt, 'batch_rays', -1) or -1) > 0:
            raise ValueError("train_step_graphed: fused Adam with the dynamic loss scaler, no batch_rays")
        from ._lib import scratch_generation
        tensors = (rays_o, rays_d, rgbs, mask)
        key = tuple((t.data_ptr(), tuple(t.shape), t.dtype) for t in tensors) + (tuple(sorted(render_kw.items())),)
        cache = self.__dict__.setdefault('_graphs', {})
        if cache and any(e[3] != scratch_generation() for e in cache.values()):
            # a grow-on-demand workspace (scatter plan, partial-gradient rows) was reallocated since a cached graph was captured — e.g. by an
            # eval_step on a larger view: it holds the freed buffer's address.  Drop them all (and their pool); every view recaptures.
            cache.clear()
            self.__dict__.pop('_graph_pool', None)
        ent = cache.get(key)
        if ent is None:
            gen0 = scratch_generation()
            for _ in range(2):
                self.train_step(rays_o, rays_d, rgbs, mask, **render_kw)
            torch.cuda.synchronize()
            self.model.train()
            graph = torch.cuda.CUDAGraph()
            self._watch_scaler()
            try:
                with torch.cuda.graph(graph, pool=self.__dict__.get('_graph_pool')), packed_weights_window(self.model, self.opt):
                    with torch.autocast('cuda', dtype=torch.float16, enabled=self.fp16):
                        outputs = self.model.render(rays_o, rays_d, staged=False, perturb=True, force_all_rays=True,
                                                    **dict(render_kw, fg_bg=render_kw.get('fg_bg', getattr(self, 'needs_fg_bg', False))))
                        loss = self.loss(outputs, rgbs, mask)
                    self.scaler.backward(loss)
                    loss = loss.detach()
            finally:
                self.scaler.watch(False)
            gen1 = scratch_generation()
            if gen1 != gen0 and cache:
                # The workspaces are keyed by (device, stream): the ones a graph references belong to the CAPTURE stream and are allocated
                # (graph pool) or re-grown during a capture.  This view grew one — in its warm-up or in its capture — so every graph captured
                # before holds the address of a block that went back to the shared pool and may alias this view's allocations: drop them
                # (they recapture on their next visit); the new graph, captured against the buffers as they are now, stays.
                cache.clear()
            self._graph_pool = graph.pool()
            ent = cache[key] = (graph, loss, tensors, gen1)          # (the tensors keep the captured addresses alive; gen1: the scratch generation this graph is valid for)
            for k_ in list(cache):                                   # (entries that survived are valid for the current generation too)
                cache[k_] = cache[k_][:3] + (gen1,)
        self.model.train()
        self._inf_folded = inf_check_is_folded(self)                 # (the captured kernels carry the pointer they were captured with: this scaler's)
        ent[0].replay()
        apply_optimizer_step(self)
        self.global_step += 1
        return ent[1], None


def train_one_epoch(trainer, views, shard=None, log=None):
    """Loop body of Trainer_Nerf.train_one_epoch (nerf/utils_init_nerf.py:577-671) over an indexable of view tuples
    (rgbs, mask, rays_o, rays_d, H, W, img_path) — e.g. customnerf_amd.nerf.provider.NerfstudioScene — for ReconTrainer or EditTrainer:
    occupancy refresh every `opt.update_extra_interval` steps on the march path (:601-606), one optimiser step per view, mean loss.
    shard = (rank, world): view-parallel data parallelism, rank r takes views r, r + world, ... (the reference uses a DistributedSampler)."""
    model, opt = trainer.model, trainer.opt
    rank, world = shard if shard is not None else (0, 1)
    model.train()
    total, n = 0.0, 0
    for i in range(rank, len(views), world):
        if getattr(model, 'cuda_ray', False) and trainer.global_step % getattr(opt, 'update_extra_interval', 16) == 0:
            with torch.autocast('cuda', dtype=torch.float16, enabled=trainer.fp16):
                model.update_extra_state()
        data = views[i]
        if hasattr(trainer, 'train_step_editing'):                     # EditTrainer
            loss, loss_dict = trainer.train_step(data)
        else:
            rgbs, mask, rays_o, rays_d, H, W, _ = data
            loss, _ = trainer.train_step(rays_o, rays_d, rgbs, mask, num_steps=opt.num_steps, upsample_steps=opt.upsample_steps,
                                         dt_gamma=getattr(opt, 'dt_gamma', 0), max_steps=opt.max_steps)
        total = total + loss                                           # stays on the device: no per-step host sync
        n += 1
    avg = float(total) / max(n, 1)
    if log is not None:
        log(f"==> Finished epoch: average_loss {avg:.6f} over {n} views, lr x{trainer.lr_factor():.4f}")
    return avg
