"""Sweep tile width / split-K of k_sd_gemm with graph-timed launches (true GPU time) over UNet + VAE shapes."""
import os, sys, subprocess, json
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    import torch
    sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
    from customnerf_amd.sd import ops, pack
    def graph_time(f, n=20):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(2): f()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(n): f()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (2 * n) * 1e3
    out = {}
    for B, C, H, Co in [(1, 512, 64, 512), (1, 128, 512, 128), (1, 256, 256, 256), (1, 512, 128, 512), (1, 256, 128, 512), (1, 128, 256, 256),
                        (2, 320, 64, 320), (2, 640, 64, 320), (2, 960, 64, 320), (2, 320, 32, 640), (2, 640, 32, 640), (2, 1280, 32, 640), (2, 1920, 32, 640),
                        (2, 640, 16, 1280), (2, 1280, 16, 1280), (2, 2560, 16, 1280), (2, 1280, 8, 1280), (2, 2560, 8, 1280)]:
        x = torch.randn(B, H, H, C, device="cuda").half(); w = pack.pack_conv(torch.randn(Co, C, 3, 3) / (9 * C) ** 0.5).cuda()
        t = graph_time(lambda: ops.conv2d(x, w, None, 3))
        out[f"conv B{B} C{C} H{H} Co{Co}"] = (t, 2 * B * H * H * Co * 9 * C / t / 1e6)
    for M, N, K in [(8192, 320, 320), (8192, 2560, 320), (8192, 320, 1280), (8192, 960, 320), (2048, 640, 640), (2048, 5120, 640), (2048, 640, 2560), (2048, 1920, 640),
                    (512, 1280, 1280), (512, 10240, 1280), (512, 1280, 5120), (512, 3840, 1280), (128, 1280, 1280), (154, 320, 768), (154, 1280, 768), (154, 2560, 768),
                    (4096, 512, 512), (4096, 512, 4096), (4096, 4096, 512)]:
        x = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
        t = graph_time(lambda: ops.linear(x, w))
        out[f"dense M{M} N{N} K{K}"] = (t, 2 * M * N * K / t / 1e6)
    print("RESULT " + json.dumps(out))
    sys.exit(0)
res = {}
for nt in (0, 1, 2):
    for sp in (0, 1, 2, 3, 4, 6, 8, 12):
        if nt == 0 and sp != 0: continue
        if nt != 0 and sp == 0: continue
        env = dict(os.environ)
        if nt: env['CNERF_SG_NT'] = str(nt)
        if sp: env['CNERF_SG_SPLITS'] = str(sp)
        o = subprocess.run([sys.executable, __file__, 'child'], env=env, capture_output=True, text=True).stdout
        line = [l for l in o.splitlines() if l.startswith('RESULT ')]
        if line: res[(nt, sp)] = json.loads(line[0][7:])
keys = list(res[(0, 0)].keys())
cols = sorted(res)
for k in keys:
    base = res[(0, 0)][k]
    best = min(res.items(), key=lambda kv: kv[1][k][0])
    row = ' '.join(f"{res[c][k][0]:6.1f}" for c in cols)
    print(f"{k:28s} auto {base[0]:7.1f} us {base[1]:6.1f} TF | best nt={best[0][0]} sp={best[0][1]:2d} {best[1][k][0]:7.1f} us ({base[0]/best[1][k][0]:.2f}x) | {row}")
print("columns:", cols)
