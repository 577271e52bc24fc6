"""Single-rank RCCL sanity check (the GPU box has one GPU): process-group init the way bench.py does it, an in-place all-reduce of a
gradient-sized flat buffer, barrier, destroy."""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
flat = torch.ones(12_262_000, device=dev)
for _ in range(3): dist.all_reduce(flat, op=dist.ReduceOp.SUM)
torch.cuda.synchronize(); t = time.time()
for _ in range(10): dist.all_reduce(flat, op=dist.ReduceOp.SUM)
torch.cuda.synchronize()
print("all_reduce of 49 MB, world 1:", (time.time() - t) / 10 * 1e3, "ms; value", float(flat[0]))
dist.barrier(); dist.destroy_process_group(); print("ok")
