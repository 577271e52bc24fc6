"""torch.distributed collectives as the data-parallel layer calls them (dp.ShardedExchange, trainer.allreduce_grads_flat, the renderer's sample-count
average, bench.py's timing): RCCL ("nccl") on the GPUs takes the tensors as they are.  A process group on the gloo backend — the CPU tests, and
the two-ranks-on-one-device GPU test (`CNERF_DP_BACKEND=gloo`: tests/test_gpu_dp_two_ranks.py), which runs the N > 1 code path end to end where
the pool has one GPU per box — gets device tensors staged through host memory: same call order, same arithmetic, no links."""
import torch
import torch.distributed as dist


def _staged(t, group):
    return t.is_cuda and dist.get_backend(group) == "gloo"


class _Done:
    def wait(self):
        return True


def all_reduce(t, op=None, group=None, async_op=False):
    op = dist.ReduceOp.SUM if op is None else op
    if _staged(t, group):
        h = t.detach().cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
        return _Done() if async_op else None
    return dist.all_reduce(t, op=op, group=group, async_op=async_op)


def all_to_all_single(out, inp, group=None, async_op=False):
    if _staged(inp, group):
        ho, hi = torch.empty(out.shape, dtype=out.dtype), inp.detach().cpu()
        dist.all_to_all_single(ho, hi, group=group)
        out.copy_(ho)
        return _Done() if async_op else None
    return dist.all_to_all_single(out, inp, group=group, async_op=async_op)


def all_gather_into_tensor(out, inp, group=None):
    if _staged(inp, group):
        hi = inp.detach().cpu()
        ho = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(ho, hi, group=group)
        out.copy_(ho)
        return None
    return dist.all_gather_into_tensor(out, inp, group=group)
