"""Checkpoint files in the reference trainer's format (nerf/utils_init_nerf.py:779-901), so that a `df_epXXXX.pth` written by the
reference can be used as `--editing_from` here and vice versa.

File = torch.save of {'epoch', 'global_step', 'stats', ['mean_count', 'mean_density' when cuda_ray], 'model': state_dict,
['optimizer', 'lr_scheduler', 'scaler', 'ema' when full]}.  The model's state-dict keys are the reference's:
aabb_train / aabb_infer / density_grid / density_bitfield / step_counter buffers, pos_en.embeddings / pos_en.offsets,
network.params / density_network.params / rgb_network.params — `*.params` is the flat float32 vector of tinycudann.Network
(row-major [out, in] matrices, widths padded to 16: the layout assumption is stated in DESIGN.md §2, tinycudann being unpinned)."""
import glob
import os

import torch


def checkpoint_state(model, epoch=0, global_step=0, stats=None, optimizer=None, full=False, scaler=None, lr_scheduler=None, ema=None):
    stale = [k for k, p in model.named_parameters() if getattr(p, '_cnerf_stale', False)]
    if stale:
        raise RuntimeError(f"{stale}: the float32 parameter lags the sharded optimiser's master shards (customnerf_amd.dp, fp16-shadow mode); "
                           "save through trainer.save_checkpoint() / trainer.state_dict() (collective: every rank calls it), which consolidate first")
    state = {'epoch': int(epoch), 'global_step': int(global_step),
             'stats': stats if stats is not None else {"loss": [], "valid_loss": [], "results": [], "checkpoints": [], "best_result": None}}
    if getattr(model, 'cuda_ray', False):
        state['mean_count'] = model.mean_count
        state['mean_density'] = model.mean_density
    if full and optimizer is not None:
        state['optimizer'] = optimizer.state_dict()
    if full and lr_scheduler is not None:
        state['lr_scheduler'] = lr_scheduler.state_dict()               # :795 (any object with state_dict(): LambdaLR, ...)
    if full and scaler is not None:
        state['scaler'] = scaler.state_dict()                           # GradScaler keys (:797-798)
    if full and ema is not None:
        state['ema'] = ema.state_dict()                                 # torch_ema's ExponentialMovingAverage (:799-800)
    state['model'] = model.state_dict()
    return state


def save_checkpoint(path, model, epoch=0, global_step=0, stats=None, optimizer=None, full=False, scaler=None, lr_scheduler=None, ema=None):
    """utils_init_nerf.py:779-815 (`best` bookkeeping and checkpoint rotation are the caller's).  A `full` checkpoint carries the same
    optional entries as the reference's: 'optimizer', 'lr_scheduler', 'scaler', 'ema' (each only when the object is passed)."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(checkpoint_state(model, epoch, global_step, stats, optimizer, full, scaler, lr_scheduler, ema), path)
    return path


def latest_checkpoint(ckpt_dir):
    files = sorted(glob.glob(os.path.join(ckpt_dir, '*.pth')))
    return files[-1] if files else None


def load_checkpoint(model, checkpoint, model_only=False, optimizer=None, map_location=None, log=print, scaler=None, lr_scheduler=None, ema=None):
    """utils_init_nerf.py:838-901.  `checkpoint`: path or an already loaded dict.  Returns a dict with what was restored:
    {'epoch', 'global_step', 'stats', 'missing_keys', 'unexpected_keys'} (epoch / step / stats None when absent or model_only)."""
    ck = torch.load(checkpoint, map_location=map_location, weights_only=False) if isinstance(checkpoint, (str, os.PathLike)) else checkpoint
    out = {'epoch': None, 'global_step': None, 'stats': None, 'missing_keys': [], 'unexpected_keys': []}
    if 'model' not in ck:                                               # a bare state dict (:850-853)
        model.load_state_dict(ck)
        return out
    sd = ck['model']
    own = model.state_dict()
    for k, v in list(sd.items()):                                       # the half shadow etc. are rebuilt, shapes must agree for what is loaded
        if k in own and tuple(own[k].shape) != tuple(v.shape):
            raise ValueError(f"checkpoint tensor {k} has shape {tuple(v.shape)}, the model expects {tuple(own[k].shape)}")
    missing, unexpected = model.load_state_dict(sd, strict=False)
    out['missing_keys'], out['unexpected_keys'] = list(missing), list(unexpected)
    if missing:
        log(f"[WARN] missing keys: {missing}")
    if unexpected:
        log(f"[WARN] unexpected keys: {unexpected}")
    if hasattr(model, 'pos_en') and hasattr(model.pos_en, 'invalidate_half_table'):
        model.pos_en.invalidate_half_table()                            # the fp16 shadow of the grid table is re-cast on next use
    if ema is not None and 'ema' in ck:                                 # :862-867 (restored even when model_only, as in the reference)
        try:
            ema.load_state_dict(ck['ema'])
        except Exception as e:
            log(f"[WARN] failed to load EMA: {e!r}")
    if getattr(model, 'cuda_ray', False):
        if 'mean_count' in ck:
            model.mean_count = ck['mean_count']
        if 'mean_density' in ck:
            model.mean_density = ck['mean_density']
    if model_only:
        return out
    out['stats'], out['epoch'], out['global_step'] = ck.get('stats'), ck.get('epoch'), ck.get('global_step')
    if optimizer is not None and 'optimizer' in ck:
        try:
            optimizer.load_state_dict(ck['optimizer'])
        except Exception as e:                                          # the reference swallows this too (:885-890)
            log(f"[WARN] Failed to load optimizer: {e!r}")
    if lr_scheduler is not None and 'lr_scheduler' in ck:               # :889-894
        try:
            lr_scheduler.load_state_dict(ck['lr_scheduler'])
        except Exception as e:
            log(f"[WARN] Failed to load scheduler: {e!r}")
    if scaler is not None and 'scaler' in ck:                           # :896-901
        try:
            scaler.load_state_dict(ck['scaler'])
        except Exception as e:
            log(f"[WARN] Failed to load scaler: {e!r}")
    return out
