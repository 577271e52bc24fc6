"""fixed cost and per-K-step cost of k_sd_gemm for a single workgroup (graph-timed)"""
import sys, torch
sys.path.insert(0, '/root/repo')
from customnerf_amd.sd import ops
def graph_time(f, n=50):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * n) * 1e3
for M, N in [(128, 128), (128, 64), (8192, 128), (32768, 128)]:
    row = []
    for K in [64, 128, 256, 512, 1024, 2048, 4096]:
        x = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
        row.append(graph_time(lambda: ops.linear(x, w)))
    print(f"M{M} N{N}: " + "  ".join(f"K{K}:{t:.1f}" for K, t in zip([64, 128, 256, 512, 1024, 2048, 4096], row)), flush=True)
