"""per-role work / barrier-wait cycles of k_field_bwd_w8 (tuning build, CNERF_W8_ABLATE |= 32)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CNERF_W8_ABLATE"] = str(int(os.environ.get("CNERF_W8_ABLATE", "0")) | 32)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from test_gpu_field import make_case
from customnerf_amd import field as fmod
L, n_geo, P = 16, 2, 1 << 21
ref, enc, x, d = make_case(L, n_geo, 1024, seed=5)
g = torch.Generator(device='cuda').manual_seed(0)
xs = (torch.rand(P, 3, device='cuda', generator=g) * 2 - 1) * 1.9
ds = torch.nn.functional.normalize(torch.randn(P // 64, 3, device='cuda', generator=g), dim=-1)
pn, pd, pr = (t.detach().clone().cuda().requires_grad_(True) for t in (ref.network, ref.density_network, ref.rgb_network))
with torch.no_grad():
    e = enc.encode(xs, bound=2.0, half=True)
gs = torch.randn(P, device='cuda', generator=g) * 0.05
gc = torch.randn(P, 4, device='cuda', generator=g)
for it in range(3):
    s, c = fmod.field(e, xs, ds, 64, 2 * L, n_geo, 4, pn, pd, pr)
    torch.autograd.backward([s, c], [gs, gc])
torch.cuda.synchronize()
ws = fmod._WS[next(iter(fmod._WS))]
total = 64 * 32 + 4096 * 3 + 1024 + 64 * 96 + 1024
tt = ws[510 * total * 4: 510 * total * 4 + 255 * 8 * 2 * 8].view(torch.int64).view(255, 8, 2).cpu().numpy()
n_phase = (P // 16 + 509) // 510 + 3
for r, name in enumerate(["Af0", "Bf0", "Af1", "Bf1", "Ab0", "Bb0", "Ab1", "Bb1"]):
    print(name, "work cycles/phase %.0f  barrier wait/phase %.0f" % (tt[:, r, 0].mean() / n_phase, tt[:, r, 1].mean() / n_phase))
