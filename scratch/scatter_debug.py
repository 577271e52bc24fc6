"""per-level comparison of the round-5 scatter against the float-atomic kernel (debug aid)"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from customnerf_amd._lib import lib, ptr, stream, check
from customnerf_amd.gridencoder import GridEncoder
enc = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048, gridtype='hash').cuda()
L, C = enc.num_levels, enc.level_dim
S = float(np.log2(enc.per_level_scale))
offs = enc._offsets_host


def scatter(x, g, binned=True):
    B = x.shape[0]
    need = ctypes.c_uint64(0)
    lib.cnerf_grid_encode_backward_workspace_bytes(offs.ctypes.data, B, 3, C, L, L, S, enc.base_resolution, 1, ctypes.addressof(need))
    ws = torch.empty(int(need.value) + 256, dtype=torch.uint8, device='cuda') if binned else None
    out = torch.zeros(enc.embeddings.shape, device='cuda')
    check(lib.cnerf_grid_encode_backward(ptr(g), ptr(x), offs.ctypes.data, ptr(out), B, 3, C, L, L, S, enc.base_resolution, None, None, 0, 0, 0, 1,
                                         ptr(ws), ws.numel() if binned else 0, stream()))
    torch.cuda.synchronize()
    return out


def per_level(a, b):
    return [float((a[offs[l]:offs[l + 1]] - b[offs[l]:offs[l + 1]]).abs().max()) for l in range(L)]


rng = np.random.default_rng(21)
for B, frac in ((70001, 0.0), (300001, 0.7), (2097152, 0.0)):
    x = torch.rand(B, 3, device='cuda', generator=torch.Generator(device='cuda').manual_seed(5))
    g = torch.from_numpy(rng.standard_normal((L, B, C)).astype(np.float32)).cuda().half()
    dead = torch.from_numpy(rng.random(B) < frac).cuda()
    g[:, dead] = 0
    a = scatter(x, g.contiguous())
    a2 = scatter(x, g.contiguous())
    r = scatter(x, g.contiguous(), binned=False)
    keep = (~dead).nonzero().squeeze(1)
    c = scatter(x[keep].contiguous(), g[:, keep].contiguous())
    print(f"B {B} dead {frac}: rerun equal {torch.equal(a, a2)}; max |binned - atomic| per level:", [f"{v:.1e}" for v in per_level(a, r)])
    print("      max |full - compact| per level:", [f"{v:.1e}" for v in per_level(a, c)], " scale", float(r.abs().max()))
