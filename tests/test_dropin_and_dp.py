"""CPU: the drop-in shim packages resolve the reference's import names; the data-parallel layer (gradient all-reduce of
ReconTrainer) is exercised with 2 gloo processes."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dropin_import_names():
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r);"
        "import raymarching, gridencoder, tinycudann as tcnn;"
        "assert raymarching.near_far_from_aabb.__module__.startswith('customnerf_amd.raymarching');"
        "assert gridencoder.GridEncoder.__module__.startswith('customnerf_amd.gridencoder');"
        "n = tcnn.Network(32, 64, {'otype': 'FullyFusedMLP', 'activation': 'ReLU', 'output_activation': 'None', 'n_neurons': 64, 'n_hidden_layers': 2});"
        "assert list(dict(n.named_parameters())) == ['params'] and n.params.numel() == 10240;"
        "names = ['near_far_from_aabb','sph_from_ray','morton3D','morton3D_invert','packbits','march_rays_train','composite_rays_train','composite_rays_train_sdf','march_rays','composite_rays'];"
        "assert all(hasattr(raymarching, k) for k in names); print('ok')"
    ) % (ROOT, os.path.join(ROOT, "customnerf_amd", "dropin"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


DP_WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from customnerf_amd.trainer import ReconTrainer
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)

class Tiny(torch.nn.Module):
    """stand-in field with the ReconTrainer-facing surface (get_params, parameters with persistent .grad)"""
    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(0)
        self.a = torch.nn.Parameter(torch.randn(37, generator=g))
        self.b = torch.nn.Parameter(torch.randn(5, 3, generator=g))
    def get_params(self, lr):
        return [{'params': [self.a], 'lr': lr * 10}, {'params': [self.b], 'lr': lr}]

class Opt: lr = 1e-2; iters = 100; train_conf = 0
m = Tiny()
tr = ReconTrainer(m, Opt(), world_size=world, fused_adam=False)
# each rank produces a different local gradient (its own shard of the "rays")
g = torch.Generator().manual_seed(100 + rank)
m.a.grad.copy_(torch.randn(37, generator=g)); m.b.grad.copy_(torch.randn(5, 3, generator=g))
from customnerf_amd.trainer import _grads_alias_flat
assert _grads_alias_flat([m.a.grad, m.b.grad], tr._flat)          # .grad are views of one flat buffer: the collective runs in place
flat_ptr = tr._flat.data_ptr()
tr.allreduce_grads()
assert tr._flat.data_ptr() == flat_ptr and m.a.grad.data_ptr() == flat_ptr
# reference: sum over ranks of what each rank generated
exp_a = sum(torch.randn(37, generator=torch.Generator().manual_seed(100 + r)) for r in range(world))
assert torch.allclose(m.a.grad, exp_a, atol=1e-6), (m.a.grad - exp_a).abs().max()
gb = [torch.empty_like(m.b.grad) for _ in range(world)]
dist.all_gather(gb, m.b.grad)
assert all(torch.equal(gb[0], x) for x in gb)            # every rank holds the same reduced gradient
# one optimiser step with the 1/world un-scale keeps the replicas bit-identical
torch._foreach_mul_([m.a.grad, m.b.grad], 1.0 / world)
tr.optimizer.step()
pa = [torch.empty_like(m.a.data) for _ in range(world)]
dist.all_gather(pa, m.a.data)
assert all(torch.equal(pa[0], x) for x in pa)
# autograd accumulates into the views in place (they stay bound to the flat buffer) ...
m.a.grad.zero_(); m.b.grad.zero_()
((m.a * (rank + 1)).sum() + (m.b * 2).sum()).backward()
assert _grads_alias_flat([m.a.grad, m.b.grad], tr._flat)
tr.allreduce_grads()
assert torch.allclose(m.a.grad, torch.full((37,), float(sum(r + 1 for r in range(world))))) and torch.allclose(m.b.grad, torch.full((5, 3), 2.0 * world))
# ... and a re-bound gradient falls back to the staging copy with the same result
m.b.grad = torch.full((5, 3), float(rank))
tr.allreduce_grads()
assert torch.allclose(m.b.grad, torch.full((5, 3), float(sum(range(world)))))
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_dp_allreduce_gloo_world2(tmp_path):
    script = tmp_path / "dp_worker.py"
    script.write_text(DP_WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29517", str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    assert out.stdout.count("ok") == 2


SHARDED_WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from customnerf_amd.dp import ShardedExchange, BIG_PARAM_MIN
from customnerf_amd.trainer import flat_grad_buffer
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
class HostScaler:                                                    # duck-typed stand-in for optim.DynamicLossScaler (whose check is a HIP kernel)
    def __init__(self): self.state = torch.tensor([1.0, 0.0, 0.0, 0.0])
    def check(self, g):
        if not bool(torch.isfinite(g).all()): self.state[2] = 1.0
for half_shadow, scaled in ((True, True), (True, False), (False, False)):
    g0 = torch.Generator().manual_seed(0)
    n_big = BIG_PARAM_MIN + 1234                                   # not a multiple of world * 64: the last shard is padded
    big = torch.nn.Parameter(torch.randn(n_big // 2, 2, generator=g0) * 0.1)
    s1 = torch.nn.Parameter(torch.randn(100, generator=g0)); s2 = torch.nn.Parameter(torch.randn(7, 3, generator=g0))
    params = [big, s1, s2]
    flat = flat_grad_buffer(params)
    ref = [p.detach().clone() for p in params]                      # single-process reference: Adam on the MEAN gradient of all ranks
    ref_opt = torch.optim.Adam([{'params': [torch.nn.Parameter(r) for r in ref]}], lr=1.0, betas=(0.9, 0.99), eps=1e-15)
    rp = ref_opt.param_groups[0]['params']
    lrs = {id(big): 5e-3, id(s1): 5e-4, id(s2): 5e-4}
    dp = ShardedExchange(params, flat, lambda p: lrs[id(p)], world, rank, half_shadow=half_shadow, scaler=HostScaler() if scaled else None)
    # the float16 payload exists only under a loss scaler; unscaled (float32 / static-scale) training sends float32 (ADVICE r3)
    assert dp.payload_dtype == (torch.float16 if scaled else torch.float32) and dp.state[0]['send'].dtype == dp.payload_dtype
    dp.async_ops = not half_shadow                                  # both forms of the collectives (blocking is the default)
    small_opt = torch.optim.Adam([s1, s2], lr=5e-4, betas=(0.9, 0.99), eps=1e-15)
    assert len(dp.state) == 1 and dp.small_seg.numel() >= 100 + 21
    for step in range(3):
        grads = []
        for r in range(world):                                     # what every rank "computes": deterministic per (rank, step)
            gr = torch.Generator().manual_seed(1000 * step + r)
            grads.append([torch.randn(p.shape, generator=gr) * (1 + r) for p in params])
        for p, gl in zip(params, grads[rank]):
            p.grad.add_(gl)                                        # accumulate into the persistent flat views, as autograd does
        dp.exchange()
        assert float(flat[:n_big].abs().max()) == 0.0              # the big gradient slot was packed and zeroed
        dp.check()
        dp.step(1.0)
        if half_shadow:                                            # the float32 parameter now lags the master shards: a direct checkpoint must refuse
            from customnerf_amd import checkpoint
            holder_m = torch.nn.Module(); holder_m.big = big
            try:
                checkpoint.checkpoint_state(holder_m); raise AssertionError("stale parameter went into a checkpoint")
            except RuntimeError as e:
                assert "consolidate" in str(e)
        for p in (s1, s2):
            p.grad.mul_(1.0 / world)                               # the small gradients arrive as SUMS
        small_opt.step(); small_opt.zero_grad(set_to_none=False)
        # ---- reference
        mean16 = sum(((gl[0] * (1.0 / world)).half().float() if scaled else gl[0] * (1.0 / world)) for gl in grads)   # payload precision, float32 sum on arrival
        rp[0].grad = mean16.clone(); rp[1].grad = sum(gl[1] for gl in grads) / world; rp[2].grad = sum(gl[2] for gl in grads) / world
        # per-parameter lr: three single-parameter Adams share the state layout of one; emulate with explicit lr scaling
        for q, lr in zip(rp, (5e-3, 5e-4, 5e-4)):
            st = ref_opt.state.setdefault(q, {})
            if not st:
                st['m'] = torch.zeros_like(q); st['v'] = torch.zeros_like(q); st['t'] = 0
            st['t'] += 1
            st['m'].mul_(0.9).add_(q.grad, alpha=0.1); st['v'].mul_(0.99).addcmul_(q.grad, q.grad, value=0.01)
            q.data.addcdiv_(st['m'], (st['v'].sqrt() / (1 - 0.99 ** st['t']) ** 0.5).add_(1e-15), value=-lr / (1 - 0.9 ** st['t']))
        dp.consolidate()
        assert not getattr(big, '_cnerf_stale', False)
        assert torch.allclose(big.detach(), rp[0].detach(), rtol=1e-5, atol=1e-6), (step, (big.detach() - rp[0]).abs().max())
        assert torch.allclose(s1.detach(), rp[1].detach(), rtol=1e-5, atol=1e-6) and torch.allclose(s2.detach(), rp[2].detach(), rtol=1e-5, atol=1e-6)
        if half_shadow:
            sh = dp.shadow_table(big)
            assert sh.dtype == torch.float16 and torch.equal(sh, big.detach().half())     # every rank holds the complete, current shadow
        allp = [torch.empty_like(big.data) for _ in range(world)]
        dist.all_gather(allp, big.data)
        assert all(torch.equal(allp[0], x) for x in allp)          # replicas identical after consolidate
    # the owner keeps 1/world of the moments
    assert dp.state[0]['m'].numel() * world < n_big + world * 64 + 1
    # checkpoint round trip: export in the unsharded layout == the single-process moments; a fresh exchange that imports them continues identically
    holder = torch.optim.Adam([big], lr=5e-3, betas=(0.9, 0.99), eps=1e-15)
    dp.export_optimizer_state(holder)
    est = holder.state[big]
    assert est['step'] == 3 and torch.allclose(est['exp_avg'], ref_opt.state[rp[0]]['m'], rtol=1e-5, atol=1e-7)
    assert torch.allclose(est['exp_avg_sq'], ref_opt.state[rp[0]]['v'], rtol=1e-5, atol=1e-9)
    saved = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in est.items()}
    big2 = torch.nn.Parameter(big.detach().clone())
    params2 = [big2, torch.nn.Parameter(s1.detach().clone()), torch.nn.Parameter(s2.detach().clone())]
    flat2 = flat_grad_buffer(params2)
    dp2 = ShardedExchange(params2, flat2, lambda p: 5e-3 if p is big2 else 5e-4, world, rank, half_shadow=half_shadow, scaler=HostScaler() if scaled else None)
    holder2 = torch.optim.Adam([big2], lr=5e-3, betas=(0.9, 0.99), eps=1e-15)
    holder2.state[big2] = saved
    dp2.import_optimizer_state(holder2)
    assert big2 not in holder2.state and dp2.state[0]['step'] == 3
    for d_, src in ((dp, params), (dp2, params2)):                  # one more step on both, same gradients
        gr = torch.Generator().manual_seed(777 + rank)
        for p in src:
            p.grad.add_(torch.randn(p.shape, generator=gr))
        d_.exchange(); d_.check(); d_.step(1.0); d_.consolidate()
    assert torch.equal(big.detach(), big2.detach())
    if half_shadow:
        assert torch.equal(dp.shadow_table(big), dp2.shadow_table(big2))
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_dp_sharded_exchange_gloo_world2(tmp_path):
    """customnerf_amd.dp.ShardedExchange on two host processes: fp16 all-to-all payload summed in fp32, sharded Adam, all-gather of the
    shadow / master shards, small parameters all-reduced — against a single-process Adam on the mean gradient."""
    script = tmp_path / "dp_sharded_worker.py"
    script.write_text(SHARDED_WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29519")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29519", str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    assert out.stdout.count("ok") == 2
