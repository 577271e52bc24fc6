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
