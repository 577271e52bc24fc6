# time of the fused attention kernel at the UNet's self-attention shapes, V row-major (cnerf_sd_attention_v) vs pre-transposed: python scratch/attn_bench.py
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
torch.manual_seed(0)
for B, T, C, H in [(2, 4096, 320, 8), (2, 1024, 640, 8), (2, 256, 1280, 8), (2, 64, 1280, 8)]:
    qkv = torch.randn(B, T, 3 * C, device='cuda').half()
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    vT = ops.transpose_v(v)
    o1 = ops.attention(q, k, v, H); o2 = ops.attention_vt(q, k, vT, H)
    ref = torch.nn.functional.scaled_dot_product_attention(*(x.reshape(B, T, H, C // H).transpose(1, 2) for x in (q, k, v))).transpose(1, 2).reshape(B, T, C)
    print(f"T={T} d={C // H}: v row-major {t(lambda: ops.attention(q, k, v, H)):7.1f} us   v^T {t(lambda: ops.attention_vt(q, k, vT, H)):7.1f} us   transpose {t(lambda: ops.transpose_v(v)):5.1f} us"
          f"   equal {torch.equal(o1, o2)}  max err vs sdpa {(o1.float() - ref.float()).abs().max().item():.2e}")
