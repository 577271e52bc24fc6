"""The edit step with a live RCCL communicator (one-rank group) and the sharded exchange forced on: hipGraph capture of the UNet beside the process
group's watchdog thread, collectives between graph replays.  Prints ms/step for plain and sharded."""
import os, sys, copy, time, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29543")
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t = torch.ones(4, device=dev); dist.all_reduce(t); torch.cuda.synchronize()
from customnerf_amd import scene as sc, tcnn
from customnerf_amd.nerf.network_grid import NeRFNetwork
from customnerf_amd.nerf.provider_utils import generate_rays
from customnerf_amd.sd import StableDiffusion
from customnerf_amd.sd.editing import EditTrainer
from customnerf_amd.trainer import setup_sharded_dp
from customnerf_amd.gridencoder import grid as ge
tcnn.set_default_dtype(torch.float16)
torch.manual_seed(0)
opt = sc.make_opt(cuda_ray=False, fp16=True, keep_bg=1000.0, lambda_sd=0.01, cfg=100.0, log_loss_item=False)
H = W = 128; V = 8
c2w = torch.from_numpy(sc.poses(V)).to(dev)
o, d = generate_rays(c2w, *sc.intrinsics(H, W), H, W, 1.0, 'nerfstudio')
o, d = o.view(V, 1, H * W, 3), d.view(V, 1, H * W, 3)
rgb, mask = sc.targets(V, H, W); rgb, mask = rgb.to(dev), mask.to(dev)
guidance = StableDiffusion(dev, '1.5', opt, seed=0)
for mode in ("plain", "sharded", "sharded-sync", "sharded", "sharded-sync", "plain"):
    model = NeRFNetwork(opt).to(dev)
    pre = copy.deepcopy(model).eval()
    tr = EditTrainer(model, pre, guidance, opt, guidance.synthetic_text_embeds(0), guidance.synthetic_text_embeds(1), fp16=True)
    if mode.startswith("sharded"):
        tr._dp = setup_sharded_dp(tr, model, True, rank=0)
        if mode == "sharded-sync":
            tr._dp.async_ops = False
    view = lambda j: (rgb[j % V], mask[j % V], o[j % V], d[j % V], H, W, f"view{j % V}")
    for i in range(2 * V + 2):
        tr.train_step(view(i))
    import customnerf_amd.sd.editing as ed
    marks = []
    orig_apply = ed.apply_optimizer_step
    dummy = torch.zeros(1 << 20, device=dev)
    def timed_apply(trainer):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); orig_apply(trainer)
        if mode == "d-all":
            w = dist.all_reduce(dummy[:1024], async_op=True); w.wait()
            dist.all_to_all_single(dummy[1024:2048], dummy[2048:3072])
            dist.all_reduce(dummy[:4], op=dist.ReduceOp.MAX)
            dist.all_gather_into_tensor(dummy[4096:8192], dummy[4096:8192])
        elif mode == "d-arside":
            side = torch.cuda.Stream() if not hasattr(timed_apply, "side2") else timed_apply.side2
            timed_apply.side2 = side
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                dist.all_reduce(dummy[:1024])
            torch.cuda.current_stream().wait_stream(side)
        elif mode == "d-arasync":
            w = dist.all_reduce(dummy[:1024], async_op=True); w.wait()
        elif mode == "d-armax":
            dist.all_reduce(dummy[:4], op=dist.ReduceOp.MAX)
        elif mode == "d-ar+a2a":
            dist.all_reduce(dummy[:1024])
            dist.all_to_all_single(dummy[1024:2048], dummy[2048:3072])
        elif mode == "d-ar":
            dist.all_reduce(dummy[:1024])
        elif mode == "d-a2a":
            dist.all_to_all_single(dummy[1024:2048], dummy[2048:3072])
        elif mode == "d-ag":
            dist.all_gather_into_tensor(dummy[4096:8192], dummy[4096:8192])
        elif mode == "d-sync":                                  # no RCCL: a cross-stream event round trip like a collective's
            side = torch.cuda.Stream() if not hasattr(timed_apply, "side") else timed_apply.side
            timed_apply.side = side
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                dummy[:1024].add_(1.0)
            torch.cuda.current_stream().wait_stream(side)
        b.record(); marks.append((a, b))
    ed.apply_optimizer_step = timed_apply
    evs = []
    def mark(tag):
        e = torch.cuda.Event(enable_timing=True); e.record(); evs.append((tag, e))
    o_render, o_sd, o_bwd = tr.model.render, tr.train_step_sd, tr.scaler.backward
    o_enc = guidance.encode_imgs
    def enc(*a, **k):
        r = o_enc(*a, **k); mark("vae_fwd"); return r
    guidance.encode_imgs = enc
    def render(*a, **k):
        mark("start"); r = o_render(*a, **k); mark("render"); return r
    def sd(*a, **k):
        r = o_sd(*a, **k); mark("sd_fwd"); return r
    def bwd(*a, **k):
        r = o_bwd(*a, **k); mark("backward"); return r
    tr.model.render, tr.train_step_sd, tr.scaler.backward = render, sd, bwd
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20):
        loss, _ = tr.train_step(view(i))
    ed.apply_optimizer_step = orig_apply
    tr.model.render, tr.train_step_sd, tr.scaler.backward = o_render, o_sd, o_bwd
    guidance.encode_imgs = o_enc
    torch.cuda.synchronize()
    import collections
    seg = collections.defaultdict(list)
    for (ta, ea), (tb, eb) in zip(evs[:-1], evs[1:]):
        seg[f"{ta}->{tb}"].append(ea.elapsed_time(eb) * 1e3)
    print(mode, {k: round(sum(v) / len(v), 1) for k, v in seg.items()}, "total", round(sum(sum(v) / len(v) for v in seg.values()), 1))
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(mode, "optimizer-step section us", sum(a.elapsed_time(b) for a, b in marks) / len(marks) * 1e3, "between sections us",
          sum(marks[i][1].elapsed_time(marks[i + 1][0]) for i in range(len(marks) - 1)) / (len(marks) - 1) * 1e3)
    print(mode, "host enqueue ms/step", t_host / 20 * 1e3, "ms/step", (time.perf_counter() - t0) / 20 * 1e3, "loss", float(loss), "good steps", tr.scaler.good_steps(), flush=True)
    ge.set_pre_scatter_hook(None)
dist.destroy_process_group()
print("done")
