#!/usr/bin/env python
"""bench.py — training rays/s + SDS edit-steps/s of the CustomNeRF hot path on MI355X (contract: see the task statement / DESIGN.md §5).

BASELINE.json's metric has two halves and the default invocation (`python bench.py --gpus N --steps K --warmup W`) times BOTH, one
after the other, with the same K / W, and prints them in ONE JSON line:

  * top level = training rays/s.  One step = one full reconstruction step of the reference's loop body (utils_init_nerf.py:194-241,
    599-629) on one 128x128 synthetic view per GPU: ray batch -> NeRFRenderer.render (the `run()` path the reference's -O2 recipe uses,
    64+64 samples; or --path march for the occupancy-march `run_cuda()` path) -> loss -> backward -> [RCCL all-reduce of the gradients
    when N>1] -> Adam.  `roofline` = the hash-grid gather against HBM, `cpu_baseline` = the oracle renderer on the host cores.
  * "secondary" = SDS edit-steps/s (utils_init_nerf.py:353-394, sd.py:115-155), same keys (value / ms_per_step / steps / warmup /
    roofline{bound: mfma} / cpu_baseline / config).

`--task recon` / `--task edit` time one half only (profiling runs).  Inputs (rays, targets, tables, weights) are resident in HBM before
either timed region.  Prints ONE JSON line from rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s spec, ~6.3 TB/s achievable)
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA peak of MI355X (no sparsity)


def measured_traffic(dtype, points_per_launch):
    """HBM bytes per gather launch from the committed rocprofv3 PMC passes (profiles/*_gather_pmc.json: FETCH_SIZE and
    WRITE_SIZE collected in separate --pmc runs of this same command, FETCH_SIZE doubled as MI355X_MICROARCH.md §HBM
    prescribes for gfx950).  The counters are per launch of a given size; scaled linearly to this run's mean launch size."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_gather_pmc.json")))
    if not files:
        return None, None
    try:
        rec = json.load(open(files[-1]))
        e = rec.get(dtype)
        if not e:
            return None, None
        return (2.0 * e["fetch_size_kb"] + e["write_size_kb"]) * 1024.0 * points_per_launch / e["points"], os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


def gather_bytes_per_point(L, C, itemsize, D=3):
    """ALGORITHMIC bytes of the grid gather forward per point: L*2^D*C*s + 4*D + L*C*s."""
    return L * (2 ** D) * C * itemsize + 4 * D + L * C * itemsize


def cpu_baseline(opt, n_rays_side=40, warm=2, steps=5):
    """The oracle (CPU restatement of the reference's pure-PyTorch renderer + C grid encoder) timed on this box's host
    cores on a BOUNDED sample of the same workload: a (side x side)-ray view of the same scene/field, same 64+64 samples,
    forward + backward + Adam.  Protocol of BASELINE.md §2 (as amended in round 3): 2 warm-up steps, median of 5, torch on
    min(32, os.cpu_count()) threads — on the 256-core GPU box ALL cores run this renderer 78x SLOWER (38.8 vs 3022.7 rays/s,
    profiles/r03a_bench.json: the renderer's small elementwise / matmul kernels drown in fork-join overhead), so "all cores" would
    understate the CPU; os.cpu_count() is in the record.  The C grid encode / scatter runs one level per OpenMP thread.
    Reported beside the GPU number, never as the thing measured."""
    from oracle import torch_oracle as to
    from customnerf_amd import scene as sc
    ncpu = os.cpu_count() or 1
    threads = min(32, ncpu)
    H = W = n_rays_side
    o, d = to.generate_rays(torch.from_numpy(sc.poses(8))[:1], *sc.intrinsics(H, W), H, W)
    o, d = o.reshape(1, -1, 3), d.reshape(1, -1, 3)
    rgb, mask = sc.targets(1, H, W)
    aabb = torch.tensor([-opt.bound] * 3 + [opt.bound] * 3)
    torch.set_num_threads(threads)
    ref = to.FieldRef(bound=opt.bound, num_levels=opt.num_levels, level_dim=opt.level_dim, base_resolution=opt.base_resolution,
                      log2_hashmap_size=opt.log2_hashmap_size, desired_resolution=opt.desired_resolution, gridtype='hash',
                      n_hidden_geo=opt.n_hidden_geo, seed=0)
    params = [ref.pos_en.embeddings, ref.network, ref.density_network, ref.rgb_network]
    optim = torch.optim.Adam([{'params': params[:1], 'lr': opt.lr * 10}, {'params': params[1:], 'lr': opt.lr}], betas=(0.9, 0.99), eps=1e-15)
    times = []
    for it in range(warm + steps):
        t0 = time.perf_counter()
        res = to.run(ref, o, d, aabb, opt.min_near, num_steps=opt.num_steps, upsample_steps=opt.upsample_steps, perturb=True, training=True)
        loss = ((res['image'].reshape(-1, 3) - rgb[0]) ** 2).mean() + opt.train_conf * ((res['render_mask'].reshape(-1) - mask[0].reshape(-1)) ** 2).mean()
        optim.zero_grad()
        loss.backward()
        optim.step()
        if it >= warm:
            times.append(time.perf_counter() - t0)
    t = sorted(times)[len(times) // 2]
    return {"value": H * W / t, "unit": "rays/s", "cores": threads, "host_cpu_count": ncpu, "kind": "port",
            "sample": f"{H}x{W}-ray view of the same scene/field ({opt.num_steps}+{opt.upsample_steps} samples, L{opt.num_levels} T2^{opt.log2_hashmap_size} grid), "
                      f"fwd+bwd+Adam, median of {steps} steps after {warm} warm-ups, {threads} torch threads of {ncpu} host cores (all cores measured 78x slower on this "
                      f"renderer: BASELINE.md §2 amendment); grid encode/scatter = C oracle, one level per OpenMP thread; MLP/renderer = torch CPU"}


def cpu_baseline_edit(steps=1):
    """SDS half of one edit step on the host cores with the CPU oracle (oracle/sd_oracle.py): SD-1.5-shaped random weights, VAE encode
    forward + input gradient and the UNet CFG pair, on a BOUNDED sample: a 256x256 VAE input / 32x32 latents (1/4 of the pixels of
    the real 512x512 / 64x64 step); the reported steps/s is the measured rate divided by 4 (convolutions and projections scale
    linearly in pixels; the self-attention share, quadratic, is undercounted — the baseline is therefore optimistic for the CPU)."""
    from oracle import sd_oracle as so
    from customnerf_amd.sd import arch
    ncpu = os.cpu_count() or 1
    threads = min(64, ncpu)              # all 256 cores of the GPU box: 229.6 s for this sample against 3.3 s on 64 threads (profiles/r03a_bench.json)
    torch.set_num_threads(threads)
    usd = arch.random_state_dict(arch.unet_params(arch.UNET_SD15), 1)
    vsd = arch.random_state_dict(arch.vae_encoder_params(arch.VAE_SD15), 2)
    g = torch.Generator().manual_seed(0)
    img = torch.rand(1, 3, 128, 128, generator=g).requires_grad_(True)
    text = torch.randn(2, 77, 768, generator=g)
    alphas = arch.alphas_cumprod()
    times = []
    for it in range(steps):
        t0 = time.perf_counter()
        loss, _, _ = so.train_step_sd(vsd, arch.VAE_SD15, usd, arch.UNET_SD15, img, text, 500, torch.randn(1, 4, 32, 32, generator=g), torch.randn(1, 4, 32, 32, generator=g),
                                      alphas, 100.0, 0.01, size=(256, 256))
        loss.backward()
        times.append(time.perf_counter() - t0)
    t = sorted(times)[len(times) // 2]
    return {"value": 1.0 / (4.0 * t), "unit": "edit-steps/s", "cores": threads, "host_cpu_count": ncpu, "kind": "port",
            "sample": f"SDS half only (VAE encode fwd+input-grad, UNet CFG pair) of one step at 256x256 / 32x32 latents = 1/4 of the pixels, {t:.1f} s measured, "
                      f"rate divided by 4; torch CPU float32 on {threads} threads of {ncpu} host cores (all cores: 70x slower); NeRF render excluded"}


def _dp_selftest(args, trainer, model, fp16):
    """--dp-selftest (one GPU): run the step with the multi-GPU gradient exchange switched on over a ONE-rank RCCL group — the collectives are
    self-copies, every kernel and stream hop of the N > 1 path is real.  What an 8-GPU run adds to this is link time only."""
    if getattr(args, "dp_selftest", False) and trainer.world_size == 1 and args.dp == "sharded":
        from customnerf_amd.trainer import setup_sharded_dp
        trainer._dp = setup_sharded_dp(trainer, model, fp16, rank=0)


class _StageEvents:
    """cnerf_profile_stage_events: the library records these events between the kernels of the grid scatter and of the field backward of the LAST
    step issued (the handles are re-recorded every step); -> milliseconds per stage after a synchronize."""
    NAMES = {(0, 1): "scatter_emit", (1, 2): "scatter_accumulate", (2, 3): "scatter_split_reduce", (4, 5): "field_backward", (5, 6): "field_reduce_partials"}

    def __init__(self):
        import ctypes
        from customnerf_amd import _lib
        self.ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
        for e in self.ev:
            e.record()                                            # (creates the handle)
        torch.cuda.synchronize()
        arr = (ctypes.c_void_p * len(self.ev))(*[e.cuda_event for e in self.ev])
        _lib.check(_lib.lib.cnerf_profile_stage_events(arr, len(self.ev)), "profile_stage_events")

    def finish(self):
        from customnerf_amd import _lib
        torch.cuda.synchronize()
        _lib.check(_lib.lib.cnerf_profile_stage_events(None, 0), "profile_stage_events")
        out = {}
        for (a, b), name in self.NAMES.items():
            try:
                out[name] = self.ev[a].elapsed_time(self.ev[b])
            except Exception:
                out[name] = None
        return out


def _timed(step, args, world, dist):
    """EXACTLY args.steps steps bracketed by barrier + synchronize on both sides; -> (seconds = max over ranks, last step's return value)"""
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = None
    for i in range(args.steps):
        out = step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        from customnerf_amd import _coll
        tt = torch.tensor([dt], device=torch.device("cuda", torch.cuda.current_device()), dtype=torch.float64)
        _coll.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    return dt, out


def run_edit(args, world, rank, dev):
    """SDS edit-steps/s.  One step = EditTrainer.train_step: render one 128x128 view of the edited field (run() path, fg/bg split),
    global/local SDS term (512x512 VAE encode fwd + input gradient, SD-1.5 UNet on the CFG pair; --sds-views V: V views per step through ONE
    UNet batch of 2V — BASELINE.json north_star "optionally SDS camera views"), background-preservation L1 against the cached pretrained
    render, backward through the renderer, [RCCL grad all-reduce], Adam.  View-parallel over ranks.  -> result dict on rank 0, None elsewhere."""
    import copy
    import torch.distributed as dist
    from customnerf_amd import scene as sc, tcnn
    from customnerf_amd.nerf.network_grid import NeRFNetwork
    from customnerf_amd.nerf.provider_utils import generate_rays
    from customnerf_amd.sd import StableDiffusion
    from customnerf_amd.sd import ops as sdops
    from customnerf_amd.sd.editing import EditTrainer
    tcnn.set_default_dtype(torch.float16)
    torch.manual_seed(0)
    opt = sc.make_opt(cuda_ray=False, fp16=True, keep_bg=1000.0, lambda_sd=0.01, cfg=100.0, log_loss_item=False)
    model = NeRFNetwork(opt).to(dev)
    pretrained = copy.deepcopy(model).eval()
    for p in pretrained.parameters():
        p.requires_grad_(False)
    guidance = StableDiffusion(dev, '1.5', opt, seed=0)                 # SD-1.5 shapes, seeded random weights (no checkpoint offline)
    H = W = args.res
    V = 8
    nv = max(1, int(getattr(args, 'sds_views', 1)))
    c2w = torch.from_numpy(sc.poses(V)).to(dev)
    rays_o, rays_d = generate_rays(c2w, *sc.intrinsics(H, W), H, W, 1.0, 'nerfstudio')
    rays_o, rays_d = rays_o.view(V, 1, H * W, 3), rays_d.view(V, 1, H * W, 3)
    rgb, mask = sc.targets(V, H, W)
    rgb, mask = rgb.to(dev), mask.to(dev)
    # the loss scale starts where the GradScaler policy settles on this workload (it backs off 65536 -> 32768 on the first high-noise timestep and
    # stays: profiles/r04*_bench.json) — started at the default, that one back-off lands inside the timed region as a skipped optimiser step
    trainer = EditTrainer(model, pretrained, guidance, opt, guidance.synthetic_text_embeds(0), guidance.synthetic_text_embeds(1), fp16=True, world_size=world, dp_mode=args.dp,
                          init_scale=32768.0)
    _dp_selftest(args, trainer, model, True)

    def view(j):
        v = j % V
        return (rgb[v], mask[v], rays_o[v], rays_d[v], H, W, f"view{v}")

    def step(i):
        base = (i * world + rank) * nv
        if nv == 1:
            return trainer.train_step(view(base))
        return trainer.train_step_multi([view(base + k) for k in range(nv)])

    with torch.no_grad():                                               # both prompts' UNet graphs are captured before the timed region, whichever
        for tz in (trainer.text_z, trainer.text_z_fg):                  # branch (global / local) the warm-up steps happen to draw
            guidance.eps_pred(torch.zeros(2 * nv, 64, 64, 8, device=dev, dtype=torch.float16), 500, tz)   # NHWC CFG pair(s), channels padded to 8 (ops.add_noise)
    n_warm = max(args.warmup, V // (world * nv) + 1)                    # warm-up also fills the per-view cache of the pretrained render (all V views)
    for i in range(n_warm):
        step(i)
    good0 = trainer.scaler.good_steps() if trainer.scaler is not None else 0          # host read, outside the timed region
    dt, (loss, ld) = _timed(step, args, world, dist)
    skipped = args.steps - (trainer.scaler.good_steps() - good0) if trainer.scaler is not None else 0
    # (every rank goes on: the roofline leg below runs two more steps, i.e. two more gradient exchanges — a rank that returned here left its
    # peers waiting in the all-to-all: found by tests/test_gpu_dp_two_ranks.py in round 5, before any multi-GPU hardware saw it)
    result = {"metric": "SDS edit-steps/s", "value": world * args.steps / dt, "unit": "edit-steps/s", "n_gpus": world, "steps": args.steps, "warmup": n_warm,
              "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
              "config": {"workload": f"cfg3 synthetic: {nv} {H}x{W} view(s)/step/GPU of the L16 T2^19 hash-grid field (run() path), SD-1.5-shaped UNet (859.5M params, random weights) on "
                                     f"{nv} CFG pair(s) at 64x64 latents + VAE encoder fwd/input-grad at 512x512, lambda_sd=0.01, keep_bg=1000, LGIE global/local alternation, fwd+bwd+Adam",
                         "sds_views_per_step": nv, "views_per_s": world * nv * args.steps / dt,
                         "parallelism": f"dp{world} (view-parallel SDS, RCCL {trainer.dp_describe()})" if world > 1 else "single GPU", "final_loss": float(loss),
                         "loss_scale": f"dynamic (GradScaler policy on device), now {trainer.scaler.get_scale():g}", "steps_skipped_on_overflow": skipped}}
    if not args.no_roofline:
        prof = []
        use_graph = guidance.use_graph
        guidance.use_graph = False                                       # eager launches so that each GEMM can be bracketed by events
        step(args.warmup + args.steps)                                   # (eager warm-up: workspace growth)
        torch.cuda.synchronize()
        # The event pairs must time the KERNELS, not the host: an eager step is ~600 launches enqueued at the host's pace, and with an idle
        # queue the gap between `event.record()` and the launch call lands inside the bracket (small GEMMs read 17 us instead of the 8 us
        # rocprofv3 shows).  A device-side sleep in front keeps the queue full while the host enqueues the whole step: events and kernels
        # then execute back to back.
        if hasattr(torch.cuda, "_sleep"):
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record(); torch.cuda._sleep(2000000); c1.record(); torch.cuda.synchronize()          # calibrate the spin counter's rate
            per_ms = 2000000 / max(c0.elapsed_time(c1), 1e-3)
            torch.cuda._sleep(int(min(120.0 * per_ms, 2e9)))                                         # ~120 ms head start for the host
        sdops.set_profile(prof)
        step(args.warmup + args.steps + 1)
        sdops.set_profile(None)
        guidance.use_graph = use_graph
        torch.cuda.synchronize()
        ms = sum(r[0].elapsed_time(r[1]) for r in prof)
        fl = sum(r[2] for r in prof)
        ach = fl / (ms * 1e-3) / 1e12
        if getattr(args, "gemm_table", None):                            # per-shape table of the step's GEMM launches (tuning aid)
            import collections
            tab = collections.defaultdict(lambda: [0, 0.0, 0.0])
            for r in prof:
                t = tab[r[3]]
                t[0] += 1; t[1] += r[0].elapsed_time(r[1]); t[2] += r[2]
            with open(args.gemm_table, "w") as fh:
                fh.write("M N K mode batch H_in Cin tstride ups : launches  ms  TFLOP/s\n")
                for k, t in sorted(tab.items(), key=lambda kv: -kv[1][1]):
                    fh.write(f"{k} : {t[0]:4d} {t[1]:8.3f} {t[2] / (t[1] * 1e-3) / 1e12:8.1f}\n")
        result["roofline"] = {"kernel": "k_sd_gemm (implicit-GEMM conv / linear / attention GEMMs of the UNet + VAE)", "bound": "mfma", "achieved": ach,
                              "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_F16_PEAK_TFLOPS, "traffic": None, "launches": len(prof),
                              "gemm_ms_per_step": ms, "algorithmic_tflop_per_step": fl / 1e12,
                              "timing": "per-launch HIP event pairs on the launch stream over one eager step enqueued behind a device-side sleep (queue kept full: "
                                        "the brackets hold kernel time, not host launch pacing); the timed region itself replays the UNet as a hipGraph"}
    if not args.no_cpu_baseline and world == 1:
        try:
            result["cpu_baseline"] = cpu_baseline_edit()
        except Exception as e:
            result["cpu_baseline"] = {"value": None, "unit": "edit-steps/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e!r}"}
    del trainer, guidance, model, pretrained
    torch.cuda.empty_cache()
    return result if rank == 0 else None


def run_recon(args, world, rank, dev):
    """training rays/s -> result dict on rank 0, None elsewhere.  --scaling weak: every rank renders its own 128x128 view (view-parallel);
    --scaling strong: the ranks split ONE view's rays into contiguous chunks (2048 rays per GPU at 8 GPUs)."""
    import torch.distributed as dist
    from customnerf_amd import scene as sc, tcnn
    from customnerf_amd.gridencoder import grid as ge
    from customnerf_amd.nerf.network_grid import NeRFNetwork
    from customnerf_amd.nerf.provider_utils import generate_rays
    from customnerf_amd.trainer import ReconTrainer

    fp16 = args.dtype == "f16"
    tcnn.set_default_dtype(torch.float16 if fp16 else torch.float32)
    torch.manual_seed(0)
    grid_kw = dict(grid_type='tiledgrid', log2_hashmap_size=21, desired_resolution=8192) if args.grid == "bear" else {}
    opt = sc.make_opt(cuda_ray=(args.path == "march"), fp16=fp16, **grid_kw)
    if getattr(args, "no_tune_traversal", False):
        opt.tune_gather_traversal = False
    if getattr(args, "no_packed_weights", False):
        opt.packed_field_weights = False
    if getattr(args, "no_fused_table_adam", False):
        opt.fuse_table_adam = False
    model = NeRFNetwork(opt).to(dev)
    H = W = args.res
    V = 8
    c2w = torch.from_numpy(sc.poses(V)).to(dev)
    rays_o, rays_d = generate_rays(c2w, *sc.intrinsics(H, W), H, W, 1.0, 'nerfstudio')       # HIP ray-gen kernel, resident in HBM
    rays_o, rays_d = rays_o.view(V, 1, H * W, 3), rays_d.view(V, 1, H * W, 3)
    rgb, mask = sc.targets(V, H, W)
    rgb, mask = rgb.to(dev), mask.to(dev)
    if args.path == "march":
        from customnerf_amd import raymarching
        grid = torch.from_numpy(sc.sphere_density_grid(model.cascade, 128, opt.bound, 1.0, 100.0)).to(dev)
        model.density_grid.copy_(grid)
        model.density_bitfield = raymarching.packbits(model.density_grid, 10.0, model.density_bitfield)
    if getattr(args, "rays", 0) and not (args.scaling == "strong" and world > 1):
        # --rays N: the rays of all views regrouped into steps of N rays (2048 = what one of 8 ranks executes when ONE 128x128 view is split
        # over the node; 32768 = two views per step) — the batch-size axis of profiles/r06_batch_sweep.json
        if (V * H * W) % args.rays:
            raise SystemExit(f"--rays {args.rays} does not divide {V * H * W}")
        V = V * H * W // args.rays
        rays_o, rays_d = rays_o.reshape(V, 1, args.rays, 3).contiguous(), rays_d.reshape(V, 1, args.rays, 3).contiguous()
        rgb, mask = rgb.reshape(V, args.rays, 3).contiguous(), mask.reshape(V, args.rays, 1).contiguous()
    trainer = ReconTrainer(model, opt, fp16=fp16, world_size=world, dp_mode=args.dp)
    _dp_selftest(args, trainer, model, fp16)
    render_kw = dict(num_steps=opt.num_steps, upsample_steps=opt.upsample_steps, dt_gamma=0, max_steps=opt.max_steps)
    strong = args.scaling == "strong"
    if strong:                                           # contiguous ray chunk per rank, resident before the timed region
        n_rays = (H * W + world - 1) // world
        lo, hi = rank * n_rays, min((rank + 1) * n_rays, H * W)
        rays_o, rays_d = rays_o[:, :, lo:hi].contiguous(), rays_d[:, :, lo:hi].contiguous()
        rgb, mask = rgb[:, lo:hi].contiguous(), mask[:, lo:hi].contiguous()
        rays_per_step = H * W
    else:
        n_rays = rays_o.shape[2]
        rays_per_step = n_rays * world

    # --graph (one GPU, run() path): render + loss + backward of each view replayed as one hipGraph, eager optimiser step (ReconTrainer.train_step_graphed)
    use_graph = bool(getattr(args, "graph", False)) and args.path == "run" and fp16

    def step(i):
        v = (i % V) if strong else (i * world + rank) % V      # weak: each rank renders its own view (view-parallel data parallelism)
        f = trainer.train_step_graphed if use_graph else trainer.train_step
        return f(rays_o[v], rays_d[v], rgb[v], mask[v], **render_kw)

    if args.prefit > 0:
        # `trained_field`: the benchmark's random-initialised field spreads its importance samples almost uniformly; a fitted field clusters them
        # around its surface, which is the state the reference's gather runs in for all but the first few hundred iterations.  Fit the analytic
        # sphere scene (multi-view consistent targets) for `prefit` untimed steps, then time the SAME step on the SAME targets.
        rgb, mask = sc.sphere_targets(rays_o.reshape(V, -1, 3), rays_d.reshape(V, -1, 3))          # (this rank's rays: already the chunk under --scaling strong)
        for i in range(args.prefit):
            step(i)
    for i in range(max(args.warmup, V) if use_graph else args.warmup):       # (graph mode: every view is captured before the timed region)
        step(i)
    prof = [] if (not args.no_roofline and not use_graph) else None            # per-launch event brackets need eager launches
    ge.set_profile(prof)
    xev = [] if trainer._dp is not None else None
    trainer._exchange_events = xev
    good0 = trainer.scaler.good_steps() if trainer.scaler is not None else 0          # host read, outside the timed region
    stage = _StageEvents() if getattr(args, "stage_events", False) else None       # events between the kernels of the scatter / the field backward
    dt, (loss, out) = _timed(step, args, world, dist)
    ge.set_profile(None)
    trainer._exchange_events = None
    stage_ms = stage.finish() if stage is not None else None

    result = None
    if rank == 0:
        value = rays_per_step * args.steps / dt
        result = {
            "metric": "training rays/s", "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"cfg2 synthetic {H}x{W} view" + ("/GPU, " if not strong else f" split over {world} GPU(s), ")
                                   + ("tiled grid L16 T2^21 desired 8192 (23.97M entries, the reference field's own table), "
                                      if args.grid == "bear" else "hash grid L16 T2^19 (6.12M entries), ")
                                   + ("run() path 64+64 samples/ray" if args.path == "run" else "run_cuda() occupancy-march path, unit-sphere occupancy")
                                   + ", fwd+bwd+Adam", "rays_per_step_per_gpu": n_rays,
                       # all ranks' rays behind ONE optimiser step: weak scaling multiplies it by N (N views per step where the reference takes one,
                       # utils_init_nerf.py:599-629), strong scaling keeps it at one view — so a weak-scaling speed-up is not read as a faster one-view step
                       "rays_per_optimizer_step": rays_per_step,
                       "parallelism": (f"dp{world} ({'ray-chunk' if strong else 'view-parallel'}, RCCL {trainer.dp_describe()})" if world > 1 else "single GPU"),
                       "path": args.path, "final_loss": float(loss),
                       "loss_scale": (f"dynamic (GradScaler policy on device), now {trainer.scaler.get_scale():g}" if trainer.scaler is not None else "none (fp32)"),
                       "steps_skipped_on_overflow": (args.steps - (trainer.scaler.good_steps() - good0)) if trainer.scaler is not None else 0},
        }
        if args.path == "march":
            result["config"]["samples_per_ray"] = out.get('num_points', 0) / n_rays
        if use_graph:
            result["config"]["graph"] = "render + loss + backward replayed as one hipGraph per view; optimiser step eager"
        if args.prefit > 0:
            result["config"]["prefit_steps"] = args.prefit
            result["config"]["workload"] += f", field pre-fitted for {args.prefit} steps to the analytic sphere scene (targets of the timed steps too)"
        if getattr(args, "rays", 0):
            result["config"]["workload"] = result["config"]["workload"].replace(f"{H}x{W} view", f"{H}x{W} views regrouped into {n_rays}-ray steps")
        if stage_ms:
            result["config"]["stage_ms"] = stage_ms
        if xev:
            # gradient exchange (pack + all-to-all + fp32 sum + sharded Adam + shadow all-gather + MLP all-reduce), event pairs on the compute stream
            result["config"]["exchange_ms"] = sum(a.elapsed_time(b) for a, b in xev) / len(xev)
        if prof:
            ms = [e0.elapsed_time(e1) for e0, e1, *_ in prof]
            pts = [p[2] for p in prof]
            L, isz = prof[0][3], prof[0][4]
            bpp = gather_bytes_per_point(L, opt.level_dim, isz)
            tot_ms, tot_b = sum(ms), sum(p * bpp for p in pts)
            achieved = tot_b / (tot_ms * 1e-3) / 1e9
            traffic, traffic_src = measured_traffic(args.dtype, sum(pts) / len(pts))
            result["roofline"] = {"kernel": ge.forward_kernel_name(fp16) + " (hash-grid gather forward)", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                                  "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                  "traffic": traffic, "traffic_source": (traffic_src + " (committed rocprofv3 --pmc passes of this command; not re-measured in this run)") if traffic_src else None,
                                  "launches": len(ms), "avg_launch_ms": tot_ms / len(ms), "points_per_launch": sum(pts) / len(pts),
                                  "algorithmic_bytes_per_point": bpp}
            if args.path == "run" and len(ms) == 2 * args.steps:       # run(): launch 2k = the stratified coarse samples, 2k + 1 = the importance samples
                for nm, sel in (("coarse", ms[0::2]), ("fine", ms[1::2])):
                    a_ = sum(pts[0::2]) * bpp / (sum(sel) * 1e-3) / 1e9
                    result["roofline"][nm] = {"avg_launch_ms": sum(sel) / len(sel), "achieved": a_, "frac": a_ / HBM_PEAK_GBS}
            tuner = model.__dict__.get('_fine_tuner')
            if tuner is not None:                                     # which traversal the importance-sample gather settled on, and the trial that decided it
                result["roofline"]["fine_traversal"] = {"choice": "sample-major" if tuner.choice else "level-major", "trials": len(tuner.history),
                                                        "last_trial_ms": ({"level_major": tuner.history[-1][1], "sample_major": tuner.history[-1][2]} if tuner.history else None)}
            if args.grid == "synthetic" and args.prefit == 0:         # the PMC pass was taken on this table in this (random-initialised) state
                result["roofline"]["l2_fabric"] = l2_fabric(args.dtype, sum(pts) / len(pts), tot_ms / len(ms))
        if not args.no_cpu_baseline and world == 1:
            try:
                result["cpu_baseline"] = cpu_baseline(opt)
            except Exception as e:                                    # the baseline leg must never take the GPU number down
                result["cpu_baseline"] = {"value": None, "unit": "rays/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e!r}"}
    del trainer, model
    torch.cuda.empty_cache()
    return result


def _json_safe(x):
    """strict JSON has no NaN / Infinity: a non-finite float (a diverged loss) goes out as null, so that the one record line always parses"""
    if isinstance(x, float):
        return x if math.isfinite(x) else None
    if isinstance(x, dict):
        return {k: _json_safe(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_json_safe(v) for v in x]
    return x


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def child_envs(n, port):
    """the N child environments of the self-launcher: one process per GPU, rendezvous on 127.0.0.1 (the reference never wired its DDP
    scaffolding — utils_init_nerf.py:76-78 — so there is no launcher to mirror; these are torchrun's variables)"""
    return [{"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1",
             "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")} for r in range(n)]


def visible_gpu_count(env=None, kfd_root="/sys/class/kfd/kfd/topology/nodes"):
    """Number of GPUs a child process would see, WITHOUT touching the HIP / HSA runtime: KFD topology nodes with simd_count > 0 (CPU nodes carry 0),
    narrowed by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when set.  None = cannot tell (no sysfs): the caller skips the check
    and lets the children fail."""
    env = os.environ if env is None else env
    try:
        n = 0
        for node in sorted(os.listdir(kfd_root)):
            with open(os.path.join(kfd_root, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except (OSError, ValueError):
        return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = env.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(args, argv, child_cmd=None, check_devices=True):
    """`python bench.py --gpus N` without torchrun: spawn N fresh worker processes (one per device) BEFORE this process makes any GPU call —
    the parent never initialises HIP (devices are counted from the KFD sysfs topology, not through torch / HIP), the children are new processes (no exec from a process that touched the GPU) — forward rank 0's record,
    exit non-zero if any child fails or if the record does not show N RCCL ranks.  --dry-launch prints the N child environments instead."""
    import subprocess
    n = args.gpus
    envs = child_envs(n, _free_port())
    child_argv = (child_cmd if child_cmd is not None else [sys.executable, os.path.abspath(__file__)]) + [a for a in argv if a != "--dry-launch"]
    if args.dry_launch:
        print(json.dumps({"dry_launch": True, "n_ranks": n, "argv": child_argv, "env": envs}))
        return 0
    if os.environ.get("CNERF_SINGLE_DEVICE") == "1":
        check_devices = False                                      # test mode: every rank on device 0 (tests/test_gpu_dp_two_ranks.py)
    have = visible_gpu_count() if check_devices else n            # sysfs only: the torch query may fall back to hipGetDeviceCount (HSA init)
    if have is not None and have < n:
        sys.stderr.write(f"bench.py: --gpus {n} but only {have} device(s) visible\n")
        return 3
    procs = []
    for r, e in enumerate(envs):
        procs.append(subprocess.Popen(child_argv, env={**os.environ, **e},
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)      # a blocking read here would outlive a dead peer
    reader.start()
    failed = False
    deadline = time.time() + 3600
    pending = list(range(n))
    while pending and time.time() < deadline:
        for r in list(pending):
            rc = procs[r].poll()
            if rc is not None:
                pending.remove(r)
                if rc != 0 and not failed:               # one rank down: the others would wait in a collective forever
                    failed = True
                    for q in pending:
                        procs[q].terminate()
        time.sleep(0.05)
    for r in pending:
        procs[r].kill()
    rcs = [p.wait() for p in procs]
    reader.join(timeout=10)
    out0 = b"".join(chunks).decode()
    if any(rcs):
        sys.stderr.write(f"bench.py: rank exit codes {rcs}\n")
        return 1
    line = [l for l in out0.splitlines() if l.startswith("{")]
    if not line:
        sys.stderr.write("bench.py: rank 0 printed no record\n")
        return 1
    rec = json.loads(line[-1])
    if rec.get("n_gpus") != n or rec.get("rccl_ranks") != n:
        sys.stderr.write(f"bench.py: asked for {n} ranks, the record shows n_gpus={rec.get('n_gpus')} rccl_ranks={rec.get('rccl_ranks')}\n")
        return 1
    print(line[-1])
    return 0


def device_identity(dev):
    """what distinguishes one physical GPU from another on this node: the driver's UUID AND the PCI address (MI355X / torch 2.10: both are
    exposed — uuid=65373131-..., pci_bus_id=220); None when neither is available (the guard below then cannot decide and says so)"""
    pr = torch.cuda.get_device_properties(dev)
    u = getattr(pr, "uuid", None)
    u = str(u) if u is not None else ""
    if u in ("", "00000000-0000-0000-0000-000000000000") or "object at" in u:
        u = ""
    pci = tuple(getattr(pr, a, None) for a in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
    pci = ":".join(str(x) for x in pci) if all(x is not None for x in pci) else ""
    return f"uuid={u}/pci={pci}" if (u or pci) else None


def check_one_gpu_per_rank(dist, dev, world, rank, identities=None):
    """Two ranks on one GPU (a mis-set HIP_VISIBLE_DEVICES, a launcher that hands every rank LOCAL_RANK 0) show up in RCCL only as a hang or a
    'duplicate GPU' abort deep inside communicator creation.  Gather every rank's device identity and refuse to run unless they are all
    different.  A rank whose identity is unknown (no UUID, no PCI address) cannot be judged: warn and go on.  `identities` (tests): the gathered
    list, instead of a collective."""
    if identities is None:
        identities = [None] * world
        dist.all_gather_object(identities, device_identity(dev))
    if any(i is None for i in identities):
        if rank == 0:
            print(f"bench.py: device identities unavailable on some ranks ({identities}): one-GPU-per-rank not verified", file=sys.stderr)
        return identities
    if len(set(identities)) != world:
        raise SystemExit(f"rank {rank}: {world} ranks but {len(set(identities))} distinct GPUs ({identities}) — one process per GPU is required "
                         "(check HIP_VISIBLE_DEVICES / LOCAL_RANK)")
    return identities


def l2_fabric(dtype, points_per_launch, avg_launch_ms):
    """L1 -> L2 request traffic of the gather (what actually binds it: VERDICT r3) from the committed PMC pass: TCP_TCC_READ_REQ x 128-byte
    lines per launch / this run's live launch time, against the guide's L2 ceiling (MI355X_MICROARCH.md §L2: ~34.5 TB/s)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_gather_pmc.json")))
    try:
        e = json.load(open(files[-1]))[dtype]
        gb = e["tcp_tcc_read_req"] * 128.0 * points_per_launch / e["points"] / 1e9
        ach = gb / (avg_launch_ms * 1e-3)
        return {"achieved": ach, "peak": 34500.0, "unit": "GB/s", "frac": ach / 34500.0, "requests_per_point": e["tcp_tcc_read_req"] / e["points"],
                "line_bytes": 128, "source": os.path.relpath(files[-1], ROOT) + " (committed --pmc TCP_TCC_READ_REQ_sum pass; launch time live)"}
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", choices=["both", "recon", "edit"], default="both",
                    help="both (default): training rays/s at the top level + SDS edit-steps/s under \"secondary\", in one JSON line; recon / edit: one half only")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--path", choices=["run", "march"], default="run")
    ap.add_argument("--dtype", choices=["f16", "f32"], default="f16")
    ap.add_argument("--res", type=int, default=128)
    ap.add_argument("--grid", choices=["synthetic", "bear"], default="synthetic",
                    help="synthetic = the benchmark scene of SURVEY.md 8d (hash, T=2^19, desired 2048; the headline number); "
                         "bear = the reference field's own table (tiledgrid, T=2^21, desired 8192; network_grid.py:89-96)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="recon leg at N>1: weak = one 128x128 view per GPU (default; a `strong` sub-record is added), strong = one view's rays split over the GPUs")
    ap.add_argument("--dp", choices=["sharded", "allreduce"], default="sharded",
                    help="N>1 gradient exchange: sharded = fp16 (fp32 without a loss scaler) all-to-all + sharded Adam + all-gather of the shadow; allreduce = one fp32 all-reduce")
    ap.add_argument("--sds-views", type=int, default=1, help="edit leg: camera views per step through one UNet batch of 2V")
    ap.add_argument("--rays", type=int, default=0,
                    help="recon leg: regroup the views' rays into steps of this many rays (default 0 = one whole view per step); must divide 8 * res * res")
    ap.add_argument("--no-fused-table-adam", action="store_true", help="recon leg: the grid table's Adam update as its own launch after the backward pass (A/B of optim.FusedAdam.arm_in_backward)")
    ap.add_argument("--no-packed-weights", action="store_true", help="recon leg: the field kernels stage their weights from the float32 parameters in every launch (A/B of trainer.packed_weights_window)")
    ap.add_argument("--no-tune-traversal", action="store_true", help="recon leg: keep the importance-sample gather level-major (no in-place TraversalTuner trials)")
    ap.add_argument("--stage-events", action="store_true", help="recon leg: per-kernel event times of the last step's scatter and field backward (config.stage_ms)")
    ap.add_argument("--prefit", type=int, default=0,
                    help="recon leg: fit the field to the analytic sphere scene for this many untimed steps first, so that the importance samples cluster around a surface")
    ap.add_argument("--dp-selftest", action="store_true", help="one GPU: force the sharded gradient exchange on over a one-rank RCCL group (no link time)")
    ap.add_argument("--graph", action="store_true", help="recon leg (one GPU, run() path): forward + backward of each view as one hipGraph (ReconTrainer.train_step_graphed)")
    ap.add_argument("--dry-launch", action="store_true", help="print the child environments `--gpus N` would spawn and exit (no GPU needed)")
    ap.add_argument("--no-variants", action="store_true", help="skip the `variants` sub-records (bear table, march path, trained field; strong scaling at N>1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--gemm-table", default=None, help="edit leg: write the per-shape table of one step's GEMM launches to this file (tuning aid)")
    args = ap.parse_args()

    # `python bench.py --gpus N` (no torchrun): this process becomes the launcher — before anything here touches the GPU runtime
    if args.dry_launch or (args.gpus > 1 and "WORLD_SIZE" not in os.environ):
        sys.exit(launch_ranks(args, sys.argv[1:]))

    # stdout carries exactly ONE line, the JSON record.  Native libraries write there too (RCCL prints a version banner at communicator creation and
    # flushes it at exit): keep a private duplicate of the real stdout for the record and point fd 1 at stderr for everybody else.
    sys.stdout.flush()
    record_fd = os.dup(1)
    os.dup2(2, 1)

    if os.environ.get("CNERF_BENCH_WATCHDOG"):
        # a rank that sits in a collective its peer never enters would wait forever: dump every thread's stack and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["CNERF_BENCH_WATCHDOG"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # test-only switches (tests/test_gpu_dp_two_ranks.py: the N > 1 code path end to end on a one-GPU box): CNERF_DP_BACKEND=gloo stages every
    # collective through host memory (customnerf_amd/_coll.py), CNERF_SINGLE_DEVICE=1 puts every rank on device 0
    backend = os.environ.get("CNERF_DP_BACKEND", "nccl")
    if os.environ.get("CNERF_SINGLE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1 or args.dp_selftest:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29544")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        if dist.get_world_size() != world:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, expected {world}")
        if world > 1 and os.environ.get("CNERF_SINGLE_DEVICE") != "1":
            check_one_gpu_per_rank(dist, dev, world, rank)
    rccl_ranks = dist.get_world_size() if dist.is_initialized() else 1

    import copy as _copy

    def variant(fn, **over):
        """the same leg with some arguments replaced — EVERY rank calls it (the legs issue collectives); at world > 1 an exception propagates:
        a rank that swallowed one would leave its peers waiting in a collective"""
        a = _copy.copy(args)
        a.no_cpu_baseline = True
        for k, v in over.items():
            setattr(a, k, v)
        if world > 1:
            return fn(a, world, rank, dev)
        try:
            return fn(a, world, rank, dev)
        except Exception as e:
            return {"error": repr(e)}

    def brief(r):
        if r is None or "error" in r:
            return r
        out = {"value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"], "steps": r["steps"], "warmup": r["warmup"], "workload": r["config"]["workload"]}
        for k in ("samples_per_ray", "prefit_steps", "exchange_ms", "graph", "stage_ms", "rays_per_step_per_gpu"):
            if k in r["config"]:
                out[k] = r["config"][k]
        if "roofline" in r:
            out["roofline"] = {k: r["roofline"][k] for k in ("achieved", "peak", "unit", "frac", "avg_launch_ms", "points_per_launch", "coarse", "fine", "fine_traversal") if k in r["roofline"]}
        return out

    result = None
    if args.task in ("both", "recon"):
        result = run_recon(args, world, rank, dev)
        plain = args.path == "run" and args.grid == "synthetic" and args.prefit == 0 and not args.no_variants and not args.graph and not args.rays  # rank-independent conditions only
        if plain and world == 1 and not args.dp_selftest:
            # the claims that used to live in builder-run profiles/, under the driver's clock: the reference field's own table
            # (network_grid.py:89-96), the occupancy-march path (renderer.py:597-718) and the gather on a FITTED field
            v = {"bear_table": brief(variant(run_recon, grid="bear")),
                 "march_path": brief(variant(run_recon, path="march", no_roofline=True)),
                 "trained_field": brief(variant(run_recon, prefit=300)),
                 "graphed_step": brief(variant(run_recon, graph=True, no_roofline=True))}
            # the N > 1 code path on this one GPU: the sharded gradient exchange forced on over a ONE-rank RCCL group (pack, all-to-all, fp32 sum,
            # sharded Adam, shadow all-gather — the collectives are self-copies, everything else is what an 8-GPU run executes): its local cost
            # (`exchange_ms`, event pairs) under the driver's clock
            try:
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ["MASTER_PORT"] = str(_free_port())
                dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
                v["dp_selftest"] = brief(variant(run_recon, dp_selftest=True, no_roofline=True))
                # what ONE of 8 ranks executes per step when one 128x128 view is split over the node (north_star's own partition, SURVEY.md 8e:
                # 2048 rays per GPU), exchange on: the strong-scaling bound of this design, measured instead of estimated
                sb = brief(variant(run_recon, dp_selftest=True, no_roofline=True, rays=2048, stage_events=True))
                # ... and the same with render + loss + backward replayed as one hipGraph per ray chunk (exchange + optimiser eager): at 2048 rays the
                # eager step is bound by the host's enqueue (~1 ms), not by its 0.5 ms of kernels
                sg = brief(variant(run_recon, dp_selftest=True, no_roofline=True, rays=2048, graph=True))
                if sb and "error" not in sb and rank == 0:
                    sb["implied_strong_scaling_bound_8gpu"] = result["ms_per_step"] / sb["ms_per_step"]
                    sb["note"] = ("one rank's share of a 16384-ray step under --scaling strong at 8 GPUs, gradient exchange on over a one-rank RCCL group "
                                  "(no link time); bound = 16384-ray step time / this step time")
                    if sg and "error" not in sg:
                        sb["graphed"] = {"ms_per_step": sg["ms_per_step"], "value": sg["value"], "exchange_ms": sg.get("exchange_ms"),
                                         "implied_strong_scaling_bound_8gpu": result["ms_per_step"] / sg["ms_per_step"]}
                    else:
                        sb["graphed"] = sg
                v["small_batch"] = sb
            except Exception as e:
                v.setdefault("dp_selftest", {"error": repr(e)})
                v.setdefault("small_batch", {"error": repr(e)})
            finally:
                if dist.is_initialized():
                    dist.destroy_process_group()
            if rank == 0:
                result["variants"] = v
        if plain and world > 1 and args.scaling == "weak":
            st = brief(variant(run_recon, scaling="strong", no_roofline=True))
            sg = brief(variant(run_recon, scaling="strong", no_roofline=True, graph=True))     # the same with each rank's chunk replayed as a hipGraph
            if rank == 0:
                result["strong"] = st
                result["strong_graphed"] = sg
    if args.task in ("both", "edit"):
        if world > 1 or args.task == "edit":
            edit = run_edit(args, world, rank, dev)
        else:
            try:
                edit = run_edit(args, world, rank, dev)
            except Exception as e:                                        # one GPU, combined line: a failing edit leg must not take the rays/s half down
                edit = {"metric": "SDS edit-steps/s", "value": None, "unit": "edit-steps/s", "error": repr(e)}
        if world == 1 and edit is not None and not edit.get("error") and edit["config"].get("steps_skipped_on_overflow") == args.steps:
            # every timed step was a skipped (non-finite) step: the number above is not a training step.  Seen ONCE in round 6 (r06a) and never
            # reproduced (tests/test_gpu_uninitialised.py poisons the allocator to make that class of bug deterministic): say so in the record,
            # keep the failed attempt, and measure again on a fresh trainer.
            first = {"ms_per_step": edit["ms_per_step"], "final_loss": edit["config"].get("final_loss"), "steps_skipped_on_overflow": args.steps}
            edit = run_edit(args, world, rank, dev)
            edit["nonfinite_first_attempt"] = first
        if args.task == "both" and args.sds_views == 1 and not args.no_variants and not (world == 1 and edit.get("error")):
            # the same leg with 4 camera views per step through one UNet batch of 8 (north_star: "optionally SDS camera views"), reported beside
            # the single-view figure: fixed per-launch costs amortise and the GEMMs' M quadruples.  The condition is rank-independent.
            mv = variant(run_edit, sds_views=4)
            if rank == 0:
                edit["multi_view"] = mv if (mv is None or "error" in mv) else {
                    "sds_views_per_step": 4, "edit_steps_per_s": mv["value"], "views_per_s": mv["config"]["views_per_s"],
                    "ms_per_step": mv["ms_per_step"], "roofline": mv.get("roofline")}
        if args.task == "edit":
            result = edit
        elif rank == 0:
            result["secondary"] = edit
    if rank == 0:
        result["rccl_ranks"] = rccl_ranks
        if backend != "nccl":
            result["collective_backend"] = backend + " (host-staged: test mode, no links)"
        os.write(record_fd, (json.dumps(_json_safe(result)) + "\n").encode())
    os.close(record_fd)
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
