import sys, time, torch
sys.path.insert(0, '/root/repo')
from customnerf_amd.sd import ops, pack
def timeit(f, n=20, w=3):
    for _ in range(w): f()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.time() - t) / n
which = sys.argv[1] if len(sys.argv) > 1 else "all"
res = []
if which in ("all", "dense"):
    for M, N, K in [(4096, 4096, 4096), (8192, 320, 320), (8192, 1280, 320), (8192, 320, 1280)]:
        x = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
        t = timeit(lambda: ops.linear(x, w))
        print(f"dense M{M} N{N} K{K}: {t*1e6:.1f} us  {2*M*N*K/t/1e12:.1f} TFLOP/s")
if which in ("all", "conv"):
    for B, C, H, Co in [(1, 128, 512, 128), (1, 256, 256, 256), (1, 512, 128, 512), (2, 320, 64, 320), (2, 640, 32, 640), (2, 1280, 16, 1280), (2, 1280, 8, 1280)]:
        x = torch.randn(B, H, H, C, device="cuda").half(); w = pack.pack_conv(torch.randn(Co, C, 3, 3) / (9 * C) ** 0.5).cuda()
        t = timeit(lambda: ops.conv2d(x, w, None, 3))
        print(f"conv B{B} C{C} H{H} Co{Co}: {t*1e6:.1f} us  {2*B*H*H*Co*9*C/t/1e12:.1f} TFLOP/s")
