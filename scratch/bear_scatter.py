"""Gather / scatter times at the reference's bear table (tiled, T = 2^21, desired 8192) on 128x128x128 ray-ordered samples."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from customnerf_amd import scene as sc
from customnerf_amd._lib import lib, check, ptr, stream, dtype_id
from customnerf_amd.gridencoder import GridEncoder
from customnerf_amd.gridencoder import grid as ge
from customnerf_amd.nerf.provider_utils import generate_rays
dev = torch.device('cuda')
H = W = 128
o, d = generate_rays(torch.from_numpy(sc.poses(1)).to(dev), *sc.intrinsics(H, W), H, W, 1.0, 'nerfstudio')
o, d = o.view(-1, 3), d.view(-1, 3)
N = o.shape[0]
z = 1.5 + 4.0 * (torch.arange(128, device=dev)[None, :] + torch.rand(N, 128, device=dev)) / 128
x = (((o[:, None, :] + d[:, None, :] * z[..., None]).clamp(-2, 2) + 2) / 4).reshape(-1, 3).contiguous()
B = x.shape[0]
def timeit(f, n=10, w=3):
    for _ in range(w): f()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.time() - t) / n * 1e3
for name, kw in (("hash T=2^19 desired 2048", dict(log2_hashmap_size=19, desired_resolution=2048, gridtype='hash')),
                 ("tiled T=2^21 desired 8192 (bear)", dict(log2_hashmap_size=21, desired_resolution=8192, gridtype='tiled'))):
    enc = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, **kw).to(dev)
    L, C, D = 16, 2, 3
    S_, Hres = float(np.log2(enc.per_level_scale)), int(enc.base_resolution)
    oh = enc._offsets_host
    grad = (torch.randn(L, B, C, device=dev) * 1e-3).half()
    gemb = torch.zeros(enc.embeddings.shape, device=dev)
    ws, wsb = ge._bwd_workspace(oh, B, D, C, L, L, S_, Hres, 1, dev)
    t_fwd = timeit(lambda: enc.encode(x * 2 - 1, bound=1, half=True))
    t_bin = timeit(lambda: check(lib.cnerf_grid_encode_backward(ptr(grad), ptr(x), oh.ctypes.data, ptr(gemb), B, D, C, L, L, S_, Hres, None, None, enc.gridtype_id, 0, 0, dtype_id(grad), ptr(ws), wsb, stream()), "bwd"))
    t_atm = timeit(lambda: check(lib.cnerf_grid_encode_backward(ptr(grad), ptr(x), oh.ctypes.data, ptr(gemb), B, D, C, L, L, S_, Hres, None, None, enc.gridtype_id, 0, 0, dtype_id(grad), None, 0, stream()), "bwd"), n=3, w=1)
    print(f"{name}: entries {enc.embeddings.shape[0]}, B {B}: gather (incl. coordinate map) {t_fwd:.3f} ms, binned scatter {t_bin:.3f} ms (workspace {wsb / 2**30:.2f} GiB), atomic scatter {t_atm:.2f} ms")
