"""Per-step cost of the sharded exchange's own kernels (pack, fp32 sum, shard Adam, RCCL self-copies) on ONE GPU: world-1 RCCL group,
ReconTrainer with the exchange forced on, against the plain trainer.  Link time of a real 8-GPU run is NOT in this number."""
import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from customnerf_amd import scene as sc, tcnn
from customnerf_amd.nerf.network_grid import NeRFNetwork
from customnerf_amd.trainer import ReconTrainer, setup_sharded_dp
from customnerf_amd.gridencoder import grid as ge
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29542")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
tcnn.set_default_dtype(torch.float16)
dev = torch.device("cuda", 0)
for mode in ("plain", "sharded", "plain", "sharded"):
    torch.manual_seed(0)
    opt = sc.make_opt(fp16=True)
    model = NeRFNetwork(opt).to(dev)
    tr = ReconTrainer(model, opt, fp16=True)
    if mode == "sharded":
        tr._dp = setup_sharded_dp(tr, model, True, rank=0)
    H = W = 128
    g = torch.Generator(device=dev); g.manual_seed(1)
    o = torch.zeros(H * W, 3, device=dev); o[:, 2] = -1.5
    d = torch.nn.functional.normalize(torch.randn(H * W, 3, device=dev, generator=g) * 0.2 + torch.tensor([0, 0, 1.0], device=dev), dim=-1)
    rgb = torch.rand(H * W, 3, device=dev, generator=g); mask = (torch.rand(H * W, device=dev, generator=g) > 0.5).float()
    model.density_bitfield.fill_(255) if hasattr(model, 'density_bitfield') else None
    for _ in range(20):
        tr.train_step(o, d, rgb, mask, num_steps=opt.num_steps, upsample_steps=opt.upsample_steps)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(200):
        tr.train_step(o, d, rgb, mask, num_steps=opt.num_steps, upsample_steps=opt.upsample_steps)
    torch.cuda.synchronize()
    print(mode, "ms/step", (time.perf_counter() - t) / 200 * 1e3, flush=True)
    ge.set_pre_scatter_hook(None)
dist.destroy_process_group()
