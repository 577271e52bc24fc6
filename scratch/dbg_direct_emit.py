"""which table entries differ under the direct (unstaged) record emit, smoothstep + align_corners config"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from test_gpu_gridencoder import build, make_inputs, cuda, BINNED_CONFIGS, co
from customnerf_amd.gridencoder import grid as G
name, kw = BINNED_CONFIGS[int(os.environ.get("CFG", "2"))]
enc = build(kw)
B = 70001
x = make_inputs(B, 3, seed=11)
x[100:200] = x[50]
L, C = enc.num_levels, enc.level_dim
g = np.random.default_rng(12).standard_normal((B, L * C)).astype(np.float32)
g = co.h2f(co.f2h(g))
ge_ref, _ = co.grid_encode_backward(g, x, tuple(enc.embeddings.shape), enc._offsets_host, enc.per_level_scale, enc.base_resolution, None,
                                    enc.gridtype_id, enc.align_corners, enc.interp_id)
table = enc.half_table()
out = G._grid_encode.apply(cuda(x), enc.embeddings, table, enc._offsets_host, enc.per_level_scale, enc.base_resolution, False,
                           enc.gridtype_id, enc.align_corners, enc.interp_id, None)
glbc = cuda(g).view(B, L, C).permute(1, 0, 2).contiguous().to(out.dtype)
out.backward(glbc)
got = enc.embeddings.grad.cpu().numpy()
d = np.abs(got - ge_ref)
bad = np.argwhere(d > 0.02 + 2e-3 * np.abs(ge_ref))
off = enc._offsets_host
print(name, "mismatches", len(bad))
for e, c in bad[:20]:
    lvl = int(np.searchsorted(off, e, side='right') - 1)
    print("entry", e, "level", lvl, "in-level", e - off[lvl], "chunk", (e - off[lvl]) >> 12, "ch", c, "got", got[e, c], "ref", ge_ref[e, c], "size", off[lvl + 1] - off[lvl])
