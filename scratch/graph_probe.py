"""Feasibility probe: the reconstruction step's render + loss + backward captured in ONE hipGraph per view (optimiser step stays eager: its learning
rate is a host-side scalar).  Compares ms per step and the loss trajectory with the eager trainer."""
import os, sys, time, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from customnerf_amd import scene as sc, tcnn
from customnerf_amd.nerf.network_grid import NeRFNetwork
from customnerf_amd.nerf.provider_utils import generate_rays
from customnerf_amd.trainer import ReconTrainer, apply_optimizer_step
dev = torch.device("cuda", 0)
tcnn.set_default_dtype(torch.float16)
H = W = 128; V = 8
def make():
    torch.manual_seed(0)
    opt = sc.make_opt(cuda_ray=False, fp16=True)
    model = NeRFNetwork(opt).to(dev)
    return opt, model, ReconTrainer(model, opt, fp16=True, world_size=1)
opt, model, trainer = make()
c2w = torch.from_numpy(sc.poses(V)).to(dev)
rays_o, rays_d = generate_rays(c2w, *sc.intrinsics(H, W), H, W, 1.0, 'nerfstudio')
rays_o, rays_d = rays_o.view(V, 1, H * W, 3), rays_d.view(V, 1, H * W, 3)
rgb, mask = sc.targets(V, H, W); rgb, mask = rgb.to(dev), mask.to(dev)
kw = dict(num_steps=opt.num_steps, upsample_steps=opt.upsample_steps, dt_gamma=0, max_steps=opt.max_steps)

def eager_run(tr, n, off=0):
    out = []
    for i in range(n):
        l, _ = tr.train_step(rays_o[(off + i) % V], rays_d[(off + i) % V], rgb[(off + i) % V], mask[(off + i) % V], **kw)
        out.append(l)
    return out

def fwd_bwd(tr, v):
    tr.model.train()
    ro, rd, tg, mk = tr.select_rays(rays_o[v], rays_d[v], rgb[v], mask[v], None)
    with torch.autocast('cuda', dtype=torch.float16, enabled=True):
        outputs = tr.model.render(ro, rd, staged=False, perturb=True, force_all_rays=True, **kw)
        loss = tr.loss(outputs, tg, mk)
    tr.scaler.backward(loss)
    return loss.detach()

# eager timing
eager_run(trainer, 10)
torch.cuda.synchronize(); t0 = time.perf_counter(); le = eager_run(trainer, 40, 10); torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 40
print(f"eager: {te * 1e3:.3f} ms/step, loss {float(le[0]):.5f} -> {float(le[-1]):.5f}")

# graphed: same initial state
opt, model, trainer = make()
eager_run(trainer, 10)                                                  # warm-up (allocations, workspaces, grads materialised and zeroed by the fused Adam)
graphs, losses = {}, {}
pool = None
side = torch.cuda.Stream()
for v in range(V):
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fwd_bwd(trainer, v); apply_optimizer_step(trainer); trainer.global_step += 1
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, pool=pool):
        losses[v] = fwd_bwd(trainer, v)
    pool = g.pool()
    graphs[v] = g
    apply_optimizer_step(trainer); trainer.global_step += 1            # the captured pass left gradients of a step that was never executed: zero them through a step
print("captured", len(graphs), "graphs")
def graphed_run(n, off=0):
    out = []
    for i in range(n):
        v = (off + i) % V
        graphs[v].replay()
        apply_optimizer_step(trainer); trainer.global_step += 1
        out.append(losses[v].clone())
    return out
graphed_run(5)
torch.cuda.synchronize(); t0 = time.perf_counter(); lg = graphed_run(40, 5); torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 40
print(f"graphed: {tg * 1e3:.3f} ms/step, loss {float(lg[0]):.5f} -> {float(lg[-1]):.5f}")
t0 = time.perf_counter(); graphed_run(40); th = (time.perf_counter() - t0) / 40; torch.cuda.synchronize()
print(f"graphed host enqueue: {th * 1e3:.3f} ms/step")
