import sys, torch
sys.path.insert(0, '.')
from customnerf_amd.sd import ops
def t(f, n=20, w=3):
    for _ in range(w): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for B in (1, 2, 4, 8):
    for T, C in ((4096, 320), (1024, 640)):
        qkv = torch.randn(B, T, 3 * C, device='cuda').half()
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        print(B, T, f"{t(lambda: ops.attention(q, k, v, 8)):.1f} us")
