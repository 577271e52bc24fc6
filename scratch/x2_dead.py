"""Cost of a dead phase of k_field_bwd_x2: the run() list size (81920 tiles) with synthetic liveness patterns."""
import sys, json, numpy as np, torch
sys.path.insert(0, '.')
from oracle import torch_oracle as to
from customnerf_amd.field import field
L, n_geo = 16, 2
rays, spr = 16384, 160
P = rays * spr
ref = to.FieldRef(bound=2.0, num_levels=L, n_hidden_geo=n_geo, seed=0)
g = torch.Generator(device='cuda').manual_seed(1)
enc = (torch.rand(L, P, 2, device='cuda', generator=g) - 0.5).half().requires_grad_(True)
xyz = (torch.rand(P, 3, device='cuda', generator=g) * 2 - 1) * 1.9
dirs = torch.nn.functional.normalize(torch.randn(rays, 3, device='cuda', generator=g), dim=-1)
pn, pd, pr = (torch.nn.Parameter(p.detach().cuda().float().contiguous()) for p in (ref.network, ref.density_network, ref.rgb_network))
n_tiles = P // 32
idx = torch.arange(n_tiles, device='cuda')
pats = {
    'all_live': torch.ones(n_tiles, dtype=torch.uint8, device='cuda'),
    'all_dead': torch.zeros(n_tiles, dtype=torch.uint8, device='cuda'),
    'half_dead_blocks_of_512': ((idx // 512) % 2).to(torch.uint8),
    'half_dead_pairs': ((idx // 2) % 2).to(torch.uint8),
    'half_dead_tiles': (idx % 2).to(torch.uint8),
    'half_dead_random_pairs': (torch.rand(n_tiles // 2, device='cuda', generator=g) < 0.5).repeat_interleave(2).to(torch.uint8),
    'columns_center_live': (((idx // 2) % 128 >= 32) & ((idx // 2) % 128 < 96)).to(torch.uint8),
}
out = {}
for name, live in pats.items():
    rows = live.repeat_interleave(32).float()
    gs = (torch.randn(P, device='cuda', generator=g) * rows).contiguous()
    gc = (torch.randn(P, 4, device='cuda', generator=g) * rows[:, None]).contiguous()
    def run(flag):
        s, c = field(enc, xyz, dirs, spr, 2 * L, n_geo, 4, pn, pd, pr)
        if flag: gs._cnerf_tile_live = live
        elif hasattr(gs, '_cnerf_tile_live'): del gs._cnerf_tile_live
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        enc.grad = None
        e0.record()
        torch.autograd.backward([s, c], [gs, gc])
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3, enc.grad.clone(), [p.grad.clone() for p in (pn, pd, pr)]
    for p in (pn, pd, pr): p.grad = None
    ts, tn = [], []
    for it in range(6):
        for p in (pn, pd, pr): p.grad = None
        t1, ge1, gp1 = run(True)
        for p in (pn, pd, pr): p.grad = None
        t0, ge0, gp0 = run(False)
        if it: ts.append(t1); tn.append(t0)
    same = torch.equal(ge0, ge1) and all(torch.equal(a, b) for a, b in zip(gp0, gp1))
    out[name] = {'live_frac': float(live.float().mean()), 'us_with_flags': float(np.median(ts)), 'us_without': float(np.median(tn)), 'bit_identical': same}
    print(name, out[name], flush=True)
json.dump(out, open(sys.argv[1], 'w'), indent=1)
