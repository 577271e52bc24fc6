"""CPU: checkpoint files in the reference trainer's format (utils_init_nerf.py:779-901) round-trip through customnerf_amd.checkpoint."""
import torch
import pytest


def _model(**kw):
    from customnerf_amd.nerf.network_grid import NeRFNetwork
    from customnerf_amd.scene import make_opt
    return NeRFNetwork(make_opt(num_levels=4, **kw))


def test_checkpoint_roundtrip_and_reference_structure(tmp_path):
    from customnerf_amd import checkpoint as ck
    m = _model(cuda_ray=True)
    with torch.no_grad():
        for p in m.parameters():
            p.uniform_(-1, 1)
        m.density_grid.uniform_(0, 20)
    m.mean_count, m.mean_density = 123, 4.5
    opt = torch.optim.Adam(m.get_params(1e-3), betas=(0.9, 0.99), eps=1e-15)
    path = ck.save_checkpoint(str(tmp_path / "checkpoints" / "df_ep0007.pth"), m, epoch=7, global_step=700, optimizer=opt, full=True)
    raw = torch.load(path, weights_only=False)
    assert set(raw) == {'epoch', 'global_step', 'stats', 'mean_count', 'mean_density', 'optimizer', 'model'}        # utils_init_nerf.py:784-799
    for k in ("aabb_train", "aabb_infer", "pos_en.embeddings", "pos_en.offsets", "network.params", "density_network.params", "rgb_network.params",
              "density_grid", "density_bitfield", "step_counter"):
        assert k in raw['model'], k
    m2 = _model(cuda_ray=True)
    opt2 = torch.optim.Adam(m2.get_params(1e-3), betas=(0.9, 0.99), eps=1e-15)
    info = ck.load_checkpoint(m2, path, optimizer=opt2)
    assert info['epoch'] == 7 and info['global_step'] == 700 and info['missing_keys'] == [] and info['unexpected_keys'] == []
    assert m2.mean_count == 123 and m2.mean_density == 4.5
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    assert ck.latest_checkpoint(str(tmp_path / "checkpoints")) == path


def test_checkpoint_reference_style_inputs(tmp_path):
    from customnerf_amd import checkpoint as ck
    m = _model()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    sd["pos_en.embeddings"] = torch.full_like(sd["pos_en.embeddings"], 0.25)
    # bare state dict (:850-853)
    info = ck.load_checkpoint(_model(), sd)
    assert info['epoch'] is None
    # model_only + unexpected / missing keys are reported, not fatal (strict=False, :855-860)
    ref = {'epoch': 3, 'global_step': 30, 'stats': {}, 'model': dict(sd, **{"bg_net.params": torch.zeros(4)})}
    del ref['model']['rgb_network.params']
    m3 = _model()
    info = ck.load_checkpoint(m3, ref, model_only=True, log=lambda *_: None)
    assert info['unexpected_keys'] == ["bg_net.params"] and info['missing_keys'] == ["rgb_network.params"] and info['epoch'] is None
    assert float(m3.pos_en.embeddings.detach().mean()) == 0.25
    # a table of another geometry is refused with a clear message
    bad = {'model': dict(sd, **{"pos_en.embeddings": torch.zeros(10, 2)})}
    with pytest.raises(ValueError):
        ck.load_checkpoint(_model(), bad)


def test_full_checkpoint_carries_lr_scheduler_and_ema(tmp_path):
    """the optional entries of the reference's `full` checkpoint (utils_init_nerf.py:794-800) and their restore (:862-894)"""
    from customnerf_amd import checkpoint as ck

    class Ema:                                                      # torch_ema's interface: state_dict / load_state_dict
        def __init__(self, v):
            self.v = v

        def state_dict(self):
            return {'decay': 0.95, 'shadow_params': [self.v.clone()]}

        def load_state_dict(self, sd):
            self.v = sd['shadow_params'][0].clone()

    m = _model()
    opt = torch.optim.Adam(m.get_params(1e-3), betas=(0.9, 0.99), eps=1e-15)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda it: 0.1 ** min(it / 100, 1))          # main.py:189
    for _ in range(5):
        opt.step(); sched.step()
    ema = Ema(torch.arange(4.0))
    path = ck.save_checkpoint(str(tmp_path / "df_ep0001.pth"), m, epoch=1, global_step=5, optimizer=opt, full=True, lr_scheduler=sched, ema=ema)
    raw = torch.load(path, weights_only=False)
    assert {'optimizer', 'lr_scheduler', 'ema'} <= set(raw) and 'scaler' not in raw
    m2 = _model()
    opt2 = torch.optim.Adam(m2.get_params(1e-3), betas=(0.9, 0.99), eps=1e-15)
    sched2 = torch.optim.lr_scheduler.LambdaLR(opt2, lambda it: 0.1 ** min(it / 100, 1))
    ema2 = Ema(torch.zeros(4))
    ck.load_checkpoint(m2, path, optimizer=opt2, lr_scheduler=sched2, ema=ema2)
    assert sched2.last_epoch == sched.last_epoch == 5 and torch.equal(ema2.v, torch.arange(4.0))


def test_half_table_invalidation_hooks():
    """`.data` writes do not move the version counter the fp16 shadow is keyed on: reset_parameters() and load_checkpoint() invalidate it"""
    from customnerf_amd import checkpoint as ck
    m = _model()
    enc = m.pos_en
    enc._half_version = ("stale",)
    enc.reset_parameters()
    assert enc._half_version is None
    enc._half_version = ("stale",)
    ck.load_checkpoint(m, {'model': m.state_dict()}, model_only=True, log=lambda *_: None)
    assert enc._half_version is None
