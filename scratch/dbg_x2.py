"""per-layer error of the fused field backward against the oracle (debug aid for field_bwd_x2.hip)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from test_gpu_field import make_case, cuda
from customnerf_amd.field import field

L, n_geo = 16, int(os.environ.get("NGEO", "2"))
P = int(os.environ.get("P", "2077"))
ref, enc, x, d = make_case(L, n_geo, P, seed=5)
ref.half = True; ref.pos_en.half = True
rng = np.random.default_rng(9)
gs = (rng.standard_normal(P) * 0.05).astype(np.float32)
gc = rng.standard_normal((P, 4)).astype(np.float32)
s_ref, c_ref, _ = ref(torch.from_numpy(x), torch.from_numpy(d))
torch.autograd.backward([s_ref, c_ref], [torch.from_numpy(gs), torch.from_numpy(gc)])
pn, pd, pr = (t.detach().clone().cuda().requires_grad_(True) for t in (ref.network, ref.density_network, ref.rgb_network))
e = enc.encode(cuda(x), bound=2.0, half=True)
s, c = field(e, cuda(x), cuda(d), 1, 2 * L, n_geo, 4, pn, pd, pr)
torch.autograd.backward([s, c], [cuda(gs), cuda(gc)])
def rel(a, b):
    a, b = a.cpu().numpy(), b.numpy()
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)), float(np.abs(a).max()), float(np.abs(b).max())
net, rnet = pn.grad, ref.network.grad
o = 0
for name, n in (("n0", 64 * 32),) + ((("n1", 4096),) if n_geo == 2 else ()) + (("n2", 4096),):
    print(name, rel(net[o:o + n], rnet[o:o + n])); o += n
print("d0", rel(pd.grad[:4096], ref.density_network.grad[:4096]))
print("dO", rel(pd.grad[4096:4096 + 64], ref.density_network.grad[4096:4096 + 64]))
r0, rr0 = pr.grad[:64 * 96].view(64, 96), ref.rgb_network.grad[:64 * 96].view(64, 96)
print("r0.dir", rel(r0[:, :27].contiguous(), rr0[:, :27].contiguous()))
print("r0.fea", rel(r0[:, 27:91].contiguous(), rr0[:, 27:91].contiguous()))
print("rO", rel(pr.grad[64 * 96:64 * 96 + 256], ref.rgb_network.grad[64 * 96:64 * 96 + 256]))
print("grid", rel(enc.embeddings.grad, ref.pos_en.embeddings.grad))
