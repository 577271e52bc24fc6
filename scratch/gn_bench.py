"""GroupNorm forward on the UNet's shapes: microseconds per call (events, 50 calls).  Run with a TUNING build and CNERF_GN_FUSED=0 / 1."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from customnerf_amd.sd import ops
shapes = [(2, 4096, 320), (2, 4096, 640), (2, 4096, 960), (2, 1024, 320), (2, 1024, 640), (2, 1024, 960), (2, 1024, 1280), (2, 1024, 1920),
          (2, 256, 640), (2, 256, 1280), (2, 256, 1920), (2, 256, 2560), (2, 64, 1280), (2, 64, 2560), (8, 4096, 320), (8, 1024, 640), (8, 256, 1280),
          (1, 4096, 512), (1, 16384, 512)]
print("fused =", os.environ.get("CNERF_GN_FUSED", "default"))
for B, HW, C in shapes:
    x = torch.randn(B, HW, C, device="cuda").half()
    gamma, beta = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
    for _ in range(5):
        ops.groupnorm(x, gamma, beta, 32, 1e-5, True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.groupnorm(x, gamma, beta, 32, 1e-5, True)
    e1.record(); torch.cuda.synchronize()
    print(f"  B={B} HW={HW:5d} C={C:4d}: {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us", flush=True)
