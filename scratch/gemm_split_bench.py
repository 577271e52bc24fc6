"""graph-timed small-M (split-K) shapes of the UNet"""
import sys, torch
sys.path.insert(0, '/root/repo')
from customnerf_amd.sd import ops, pack
def graph_time(f, n=20):
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
for B, C, H, Co in [(2, 1280, 8, 1280), (2, 2560, 8, 1280), (2, 640, 16, 1280), (2, 1280, 16, 1280), (2, 2560, 16, 1280), (2, 1920, 16, 1280), (2, 320, 32, 640), (2, 640, 32, 640), (2, 1280, 32, 640),
                    (2, 1920, 32, 640), (2, 960, 32, 640), (2, 320, 64, 320), (2, 640, 64, 320), (2, 960, 64, 320), (1, 512, 64, 512)]:
    x = torch.randn(B, H, H, C, device="cuda").half(); w = pack.pack_conv(torch.randn(Co, C, 3, 3) / (9 * C) ** 0.5).cuda()
    t = graph_time(lambda: ops.conv2d(x, w, None, 3))
    print(f"conv B{B} C{C} H{H} Co{Co}: {t:.1f} us  {2*B*H*H*Co*9*C/t/1e6:.1f} TFLOP/s", flush=True)
for M, N, K in [(512, 1280, 5120), (2048, 640, 2560), (8192, 320, 1280), (512, 1280, 1280), (128, 1280, 1280), (128, 1280, 5120), (4096, 512, 4096)]:
    x = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    t = graph_time(lambda: ops.linear(x, w))
    print(f"dense M{M} N{N} K{K}: {t:.1f} us  {2*M*N*K/t/1e6:.1f} TFLOP/s", flush=True)
