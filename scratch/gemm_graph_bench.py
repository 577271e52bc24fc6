"""true GPU time of small GEMMs: 50 launches captured in one HIP graph (no host pacing), per-launch = total / 50"""
import sys, torch
sys.path.insert(0, '/root/repo')
from customnerf_amd.sd import ops, pack
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
for M, N, K in [(8192, 320, 320), (2048, 640, 640), (512, 1280, 1280), (154, 320, 768), (154, 1280, 768), (128, 1280, 1280), (8192, 2560, 320), (8192, 320, 1280), (512, 10240, 1280), (4096, 4096, 512), (4096, 512, 4096)]:
    x = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    t = graph_time(lambda: ops.linear(x, w))
    print(f"dense M{M} N{N} K{K}: {t:.1f} us  {2*M*N*K/t/1e6:.1f} TFLOP/s", flush=True)
for B, C, H, Co in [(1, 512, 64, 512), (2, 1280, 8, 1280), (2, 1280, 16, 1280), (2, 640, 32, 640), (2, 320, 64, 320), (1, 128, 512, 128), (1, 256, 256, 256), (1, 512, 128, 512)]:
    x = torch.randn(B, H, H, C, device="cuda").half(); w = pack.pack_conv(torch.randn(Co, C, 3, 3) / (9 * C) ** 0.5).cuda()
    t = graph_time(lambda: ops.conv2d(x, w, None, 3), n=20)
    print(f"conv B{B} C{C} H{H} Co{Co}: {t:.1f} us  {2*B*H*H*Co*9*C/t/1e6:.1f} TFLOP/s", flush=True)
