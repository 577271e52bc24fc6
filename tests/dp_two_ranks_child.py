"""Child process of tests/test_gpu_dp_two_ranks.py: one data-parallel rank of a two-rank ReconTrainer (sharded gradient exchange) — both ranks on
device 0, collectives on gloo staged through host memory (customnerf_amd/_coll.py).  usage: python dp_two_ranks_child.py <out_dir> <steps>"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from dp_two_ranks_common import build, view_draws, KW          # noqa: E402


def main():
    out_dir, steps = sys.argv[1], int(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from customnerf_amd.trainer import ReconTrainer
    model, opt, views = build()
    tr = ReconTrainer(model, opt, fp16=True, world_size=world, dp_mode='sharded')
    assert tr._dp is not None
    losses = []
    for s in range(steps):
        v = s * world + rank
        ro, rd, rgb, mask = views[v]
        loss, _ = tr.train_step(ro, rd, rgb, mask, _draws=view_draws(v), **KW)
        losses.append(float(loss))
    tr._dp.consolidate()                                        # collective: the float32 table from the owners' master shards
    if rank == 0:
        torch.save({"params": [p.detach().cpu() for p in model.parameters()], "shadow": model.pos_en.half_table().detach().cpu(),
                    "scale": tr.scaler.state.detach().cpu(), "losses": losses}, os.path.join(out_dir, "rank0.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
