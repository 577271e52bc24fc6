"""kernel time of the fused field backward on 2 M ray-ordered samples (event brackets around the autograd backward; the reduce launch is included)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from test_gpu_field import make_case
from customnerf_amd import field as fmod
L, n_geo, P = 16, int(os.environ.get("NGEO", "2")), 1 << 21
ref, enc, x, d = make_case(L, n_geo, 1024, seed=5)
g = torch.Generator(device='cuda').manual_seed(0)
xs = (torch.rand(P, 3, device='cuda', generator=g) * 2 - 1) * 1.9
ds = torch.nn.functional.normalize(torch.randn(P // 64, 3, device='cuda', generator=g), dim=-1)
pn, pd, pr = (t.detach().clone().cuda().requires_grad_(True) for t in (ref.network, ref.density_network, ref.rgb_network))
with torch.no_grad():
    e = enc.encode(xs, bound=2.0, half=True)
gs = torch.randn(P, device='cuda', generator=g) * 0.05
gc = torch.randn(P, 4, device='cuda', generator=g)
ts = []
for it in range(8):
    s, c = fmod.field(e, xs, ds, 64, 2 * L, n_geo, 4, pn, pd, pr)
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    torch.autograd.backward([s, c], [gs, gc])
    t1.record(); torch.cuda.synchronize()
    ts.append(t0.elapsed_time(t1))
print("field backward (kernel + partial reduce + autograd glue) ms:", " ".join(f"{t:.3f}" for t in ts), " min", min(ts))
