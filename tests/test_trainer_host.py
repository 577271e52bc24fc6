"""Host-side trainer logic that needs no GPU: --batch_rays subsampling (utils_init_nerf.py:210-215)."""
import argparse

import numpy as np
import torch

from customnerf_amd.trainer import ReconTrainer


def test_batch_rays_draw_is_the_references_numpy_draw():
    N = 16 * 16
    g = torch.Generator().manual_seed(0)
    rays_o, rays_d = torch.randn(1, N, 3, generator=g), torch.randn(1, N, 3, generator=g)
    rgbs, mask = torch.rand(N, 3, generator=g), torch.rand(N, 1, generator=g)
    shim = argparse.Namespace(opt=argparse.Namespace(batch_rays=100))
    np.random.seed(4242)
    want = np.random.choice(N, size=[100], replace=False)                 # the reference's call, same global generator state
    np.random.seed(4242)
    o, d, c, m = ReconTrainer.select_rays(shim, rays_o, rays_d, rgbs, mask)
    assert o.shape == (1, 100, 3) and c.shape == (1, 100, 3) and m.shape == (1, 100, 1)
    assert torch.equal(o[0], rays_o[0, want]) and torch.equal(d[0], rays_d[0, want]) and torch.equal(c[0], rgbs[want]) and torch.equal(m[0], mask[want])
    assert len(set(want.tolist())) == 100                                  # without replacement
    shim.opt.batch_rays = 0                                                # default: the whole view, untouched
    assert ReconTrainer.select_rays(shim, rays_o, rays_d, rgbs, mask)[0] is rays_o


class _FakeEvent:
    def __init__(self, t, done=True):
        self.t, self.done = t, done

    def query(self):
        return self.done

    def elapsed_time(self, other):
        return other.t - self.t


def test_traversal_tuner_schedule_and_decision(monkeypatch):
    """gridencoder.grid.TraversalTuner (round 6): both traversals on calls `first` and `first + 1` (order swapped), then every `period` calls; the
    decision is taken from the four event pairs once ALL of them have completed (no blocking), with a 2 % margin; nothing is recorded or queried
    while a hipGraph is being captured."""
    from customnerf_amd.gridencoder import grid as G
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)
    t = G.TraversalTuner(first=2, period=8)
    plans = [t.plan() for _ in range(12)]
    assert plans[0] == plans[1] == [G.LEVEL_MAJOR]
    assert plans[2] == [G.LEVEL_MAJOR, G.SAMPLE_MAJOR] and plans[3] == [G.SAMPLE_MAJOR, G.LEVEL_MAJOR]
    assert plans[4:10] == [[G.LEVEL_MAJOR]] * 6 and plans[10] == [G.LEVEL_MAJOR, G.SAMPLE_MAJOR] and plans[11] == [G.SAMPLE_MAJOR, G.LEVEL_MAJOR]
    # a finished trial: sample-major 10 % faster -> chosen; the decision needs all four pairs complete
    t = G.TraversalTuner(first=0, period=100)
    t.plan(); t.plan()
    pend = _FakeEvent(3.0, done=False)
    t.record(G.LEVEL_MAJOR, _FakeEvent(0.0), _FakeEvent(1.0)); t.record(G.SAMPLE_MAJOR, _FakeEvent(1.0), _FakeEvent(1.9))
    t.record(G.SAMPLE_MAJOR, _FakeEvent(2.0), _FakeEvent(2.9)); t.record(G.LEVEL_MAJOR, _FakeEvent(3.0), pend)
    assert t.plan() == [G.LEVEL_MAJOR] and not t.history                   # one pair still running: nothing decided, nothing blocked
    pend.t, pend.done = 4.0, True
    assert t.plan() == [G.SAMPLE_MAJOR] and t.choice == G.SAMPLE_MAJOR and len(t.history) == 1
    assert abs(t.history[0][1] - 1.0) < 1e-9 and abs(t.history[0][2] - 0.9) < 1e-9
    # inside the margin: the choice stays
    t.pending = [(G.LEVEL_MAJOR, _FakeEvent(0.0), _FakeEvent(1.0)), (G.SAMPLE_MAJOR, _FakeEvent(0.0), _FakeEvent(0.995)),
                 (G.LEVEL_MAJOR, _FakeEvent(0.0), _FakeEvent(1.0)), (G.SAMPLE_MAJOR, _FakeEvent(0.0), _FakeEvent(0.995))]
    t.choice = G.LEVEL_MAJOR
    t.plan()
    assert t.choice == G.LEVEL_MAJOR
    # under capture: the current choice, no event traffic
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: True)
    t2 = G.TraversalTuner(first=0, period=4)
    t2.pending = [(G.LEVEL_MAJOR, None, None)] * 4                          # would raise if harvested
    assert [t2.plan() for _ in range(6)] == [[G.LEVEL_MAJOR]] * 6


def test_inf_check_fold_conditions():
    """trainer.inf_check_is_folded: only a single-GPU fused half-precision field whose trainable parameters are exactly the four tensors the two
    watched producers write may skip the explicit GradScaler check"""
    from customnerf_amd.trainer import inf_check_is_folded

    class P:
        def __init__(self, t):
            self.embeddings = t

    class Net:
        def __init__(self, t):
            self.params = t

    class Model:
        grad_in_place = True

        def __init__(self, extra=None, fused=True, half=True):
            mk = lambda: torch.nn.Parameter(torch.zeros(4))
            self.pos_en, self.network, self.density_network, self.rgb_network = P(mk()), Net(mk()), Net(mk()), Net(mk())
            self.extra, self._f, self._h = extra, fused, half

        def _fused_cfg(self):
            return (32, 2, 4) if self._f else None

        def _half(self):
            return self._h

        def parameters(self):
            ps = [self.pos_en.embeddings, self.network.params, self.density_network.params, self.rgb_network.params]
            return ps + ([self.extra] if self.extra is not None else [])

    def trainer(model, **kw):
        d = dict(model=model, world_size=1, _dp=None, scaler=object(), fused_adam=True, opt=argparse.Namespace())
        d.update(kw)
        return argparse.Namespace(**d)

    assert inf_check_is_folded(trainer(Model()))
    assert not inf_check_is_folded(trainer(Model(), world_size=2))
    assert not inf_check_is_folded(trainer(Model(), _dp=object()))
    assert not inf_check_is_folded(trainer(Model(), scaler=None))
    assert not inf_check_is_folded(trainer(Model(fused=False)))
    assert not inf_check_is_folded(trainer(Model(half=False)))
    assert not inf_check_is_folded(trainer(Model(extra=torch.nn.Parameter(torch.zeros(2)))))      # a parameter nobody watches
    frozen = torch.nn.Parameter(torch.zeros(2), requires_grad=False)
    assert inf_check_is_folded(trainer(Model(extra=frozen)))                                       # ... unless it is not trained
    assert not inf_check_is_folded(trainer(Model(), opt=argparse.Namespace(fold_inf_check=False)))
    m = Model(); m.grad_in_place = False
    assert not inf_check_is_folded(trainer(m))


def test_table_step_is_armed_only_where_one_backward_pass_owns_the_gradient(monkeypatch):
    """ReconTrainer._arm_table_step: the grid table's Adam update moves into the backward scatter only for a single-GPU fused-Adam step whose inf
    check is folded into the gradient producers, outside stream capture, and not when switched off; when it arms, the learning rate of the step
    is in the parameter groups and the un-scaling factor is set before the optimiser is asked"""
    calls = []

    class Opt:
        param_groups = [{'lr': 0.0}, {'lr': 0.0}]
        grad_scale_inv = None

        def arm_in_backward(self, p):
            calls.append((p, [g['lr'] for g in self.param_groups], self.grad_scale_inv))
            return True

    class PosEn:
        embeddings = object()

    class Model:
        pos_en = PosEn()

    def trainer(**kw):
        d = dict(model=Model(), optimizer=Opt(), world_size=1, _dp=None, fused_adam=True, _inf_folded=True, opt=argparse.Namespace(), base_lrs=[1.0, 0.1],
                 loss_scale=1.0, lr_factor=lambda: 0.5)
        d.update(kw)
        return argparse.Namespace(**d)

    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)
    assert ReconTrainer._arm_table_step(trainer()) is True
    assert calls == [(PosEn.embeddings, [0.5, 0.05], 1.0)]
    for kw in (dict(fused_adam=False), dict(_inf_folded=False), dict(world_size=2), dict(_dp=object()), dict(opt=argparse.Namespace(fuse_table_adam=False))):
        assert ReconTrainer._arm_table_step(trainer(**kw)) is False
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: True)
    assert ReconTrainer._arm_table_step(trainer()) is False
    assert len(calls) == 1
