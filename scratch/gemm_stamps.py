import sys, torch
sys.path.insert(0, '/root/repo')
from customnerf_amd.sd import ops
for M, N, K in [(128, 128, 64), (128, 128, 1024), (8192, 320, 320), (512, 1280, 1280), (2048, 640, 640)]:
    x = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    for _ in range(3): ops.linear(x, w)
    torch.cuda.synchronize()
    ops.linear(x, w, alpha=1.0009765625) if 'alpha' in ops.linear.__code__.co_varnames else ops.linear(x, w)
    torch.cuda.synchronize()
