"""Share of the dense (coarse) levels in the binned grid backward: time the backward with max_level = 5 / 16 on ray-ordered samples."""
import sys, time, torch, numpy as np, ctypes
sys.path.insert(0, '/root/repo')
from customnerf_amd import scene as sc
from customnerf_amd._lib import lib, check, ptr, stream, dtype_id
from customnerf_amd.gridencoder import GridEncoder
from customnerf_amd.gridencoder import grid as ge
from customnerf_amd.nerf.provider_utils import generate_rays
dev = torch.device('cuda')
enc = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048).to(dev)
H = W = 128
o, d = generate_rays(torch.from_numpy(sc.poses(1)).to(dev), *sc.intrinsics(H, W), H, W, 1.0, 'nerfstudio')
o, d = o.view(-1, 3), d.view(-1, 3)
N = o.shape[0]
def samples(S, jitter):
    z = 1.5 + 4.0 * (torch.arange(S, device=dev)[None, :] + (torch.rand(N, S, device=dev) if jitter else 0.5)) / S
    x = (o[:, None, :] + d[:, None, :] * z[..., None]).clamp(-2, 2)
    return ((x + 2) / 4).reshape(-1, 3).contiguous()
coarse = samples(64, True)
fine_z = 1.5 + 4.0 * torch.rand(N, 64, device=dev)
fine = (((o[:, None, :] + d[:, None, :] * fine_z[..., None]).clamp(-2, 2) + 2) / 4).reshape(-1, 3).contiguous()
x = torch.cat([coarse, fine], 0).contiguous()
B = x.shape[0]
L, C, D = 16, 2, 3
S_, Hres = float(np.log2(enc.per_level_scale)), int(enc.base_resolution)
oh = enc._offsets_host
for mode in ("ray-ordered", "shuffled"):
    xs = x if mode == "ray-ordered" else x[torch.randperm(B, device=dev)].contiguous()
    for ml in (5, 16, 11):
        grad = (torch.randn(L, B, C, device=dev) * 1e-3).half()
        gemb = torch.zeros(enc.embeddings.shape, device=dev)
        ws, wsb = ge._bwd_workspace(oh, B, D, C, L, ml, S_, Hres, 1, dev)
        def run():
            check(lib.cnerf_grid_encode_backward(ptr(grad), ptr(xs), oh.ctypes.data, ptr(gemb), B, D, C, L, ml, S_, Hres, None, None, 0, 0, 0, dtype_id(grad), ptr(ws), wsb, stream()), "bwd")
        for _ in range(3): run()
        torch.cuda.synchronize(); t = time.time()
        for _ in range(10): run()
        torch.cuda.synchronize()
        print(f"{mode:12s} levels 0..{ml-1}: {(time.time()-t)/10*1e3:.3f} ms")
