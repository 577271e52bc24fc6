"""A training TRAJECTORY with and without early termination (and twice with): 1200 steps on the analytic sphere scene from one seed, the table
compared bit for bit at checkpoints — the per-step bit-identity of tests/test_gpu_train.py::test_early_termination_leaves_every_gradient_bit_identical
compounded over a run in which the field sharpens, bins crowd (spill runs) and 70 % of the rows die.  usage: python scratch/soak_early_term.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from customnerf_amd import scene as sc, tcnn
from customnerf_amd.nerf.network_grid import NeRFNetwork
from customnerf_amd.nerf.provider_utils import generate_rays
from customnerf_amd.trainer import ReconTrainer
dev = torch.device('cuda')
H = W = 128; V = 8; STEPS = int(os.environ.get("STEPS", "1200")); EVERY = 200

def run(early):
    tcnn.set_default_dtype(torch.float16)
    torch.manual_seed(0)
    torch.cuda.manual_seed(0)
    opt = sc.make_opt(cuda_ray=False, fp16=True)
    opt.early_termination = early
    model = NeRFNetwork(opt).to(dev)
    c2w = torch.from_numpy(sc.poses(V)).to(dev)
    ro, rd = generate_rays(c2w, *sc.intrinsics(H, W), H, W, 1.0, 'nerfstudio')
    ro, rd = ro.view(V, 1, H * W, 3), rd.view(V, 1, H * W, 3)
    rgb, mask = sc.sphere_targets(ro.reshape(V, -1, 3), rd.reshape(V, -1, 3))
    tr = ReconTrainer(model, opt, fp16=True)
    kw = dict(num_steps=opt.num_steps, upsample_steps=opt.upsample_steps, dt_gamma=0, max_steps=opt.max_steps)
    snaps, losses = [], []
    for i in range(STEPS):
        loss, _ = tr.train_step(ro[i % V], rd[i % V], rgb[i % V], mask[i % V], **kw)
        if (i + 1) % EVERY == 0:
            snaps.append([p.detach().clone() for p in model.parameters()])
            losses.append(float(loss))
    return snaps, losses, tr.scaler.get_scale()

a, la, sa = run(True)
b, lb, sb = run(True)
c, lc, sc_ = run(False)
ok = True
for k in range(len(a)):
    same_ab = all(torch.equal(x, y) for x, y in zip(a[k], b[k]))
    same_ac = all(torch.equal(x, y) for x, y in zip(a[k], c[k]))
    print(f"step {(k + 1) * EVERY:5d}: loss {la[k]:.6f} | repeat run bit-identical: {same_ab} | without early termination bit-identical: {same_ac} (loss {lc[k]:.6f})", flush=True)
    ok &= same_ab and same_ac
print("loss scale", sa, sb, sc_)
print("ALL BIT-IDENTICAL" if ok else "MISMATCH")
