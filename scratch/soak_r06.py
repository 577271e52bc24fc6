"""A training TRAJECTORY with the round-6 step-level switches on and off: 800 steps on the analytic sphere scene from one seed, every parameter compared bit
for bit at checkpoints — packed field weights (trainer.packed_weights_window) and the table's Adam update inside the scatter (optim.FusedAdam.arm_in_backward)
against the plain step (`opt.packed_field_weights = False`, `opt.fuse_table_adam = False`), while the field sharpens, bins crowd and split, rows die and the
loss scale grows.  usage: python scratch/soak_r06.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from customnerf_amd import scene as sc, tcnn
from customnerf_amd.nerf.network_grid import NeRFNetwork
from customnerf_amd.nerf.provider_utils import generate_rays
from customnerf_amd.trainer import ReconTrainer
dev = torch.device('cuda')
H = W = 128; V = 8; STEPS = int(os.environ.get("STEPS", "800")); EVERY = 200


def run(on):
    tcnn.set_default_dtype(torch.float16)
    torch.manual_seed(0)
    torch.cuda.manual_seed(0)
    opt = sc.make_opt(cuda_ray=False, fp16=True)
    opt.packed_field_weights = on
    opt.fuse_table_adam = on
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
            st = tr.optimizer.state[model.pos_en.embeddings]
            snaps.append([p.detach().clone() for p in model.parameters()] + [st['exp_avg'].clone(), st['exp_avg_sq'].clone(), tr.scaler.state.clone()])
            losses.append(float(loss))
    return snaps, losses, tr.scaler.get_scale(), tr.scaler.good_steps()


a, la, sa, ga = run(True)
b, lb, sb, gb = run(False)
ok = True
for k in range(len(a)):
    same = all(torch.equal(x, y) for x, y in zip(a[k], b[k]))
    print(f"step {(k + 1) * EVERY:5d}: loss {la[k]:.6f} | plain step bit-identical (parameters, table moments, scaler state): {same} (loss {lb[k]:.6f})", flush=True)
    ok &= same
print("loss scale", sa, sb, "| counted steps", ga, gb)
print("ALL BIT-IDENTICAL" if ok else "MISMATCH")
