"""per-node cost of tiny kernels: in a HIP graph, and eager back-to-back on a stream"""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from customnerf_amd.sd import ops
x = torch.zeros(1024, device="cuda")
y = torch.randn(64, 320, device="cuda").half(); w = torch.randn(320, 320, device="cuda").half()
def run(f, n, label):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    tg = e0.elapsed_time(e1) / n * 1e3
    # eager: enqueue behind a long kernel so that the host gets ahead
    big = torch.randn(8192, 8192, device="cuda")
    torch.cuda.synchronize()
    for _ in range(4): big @ big
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    te = e0.elapsed_time(e1) / n * 1e3
    print(f"{label}: graph {tg:.2f} us/node, eager (host ahead) {te:.2f} us/launch", flush=True)
run(lambda: x.add_(1.0), 200, "torch add_ 1024 elements")
run(lambda: ops.linear(y, w), 200, "k_sd_gemm M64 N320 K320")
