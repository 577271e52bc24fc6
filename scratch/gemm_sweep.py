"""Sweep tile width / split-K of k_sd_gemm over the UNet's conv and linear shapes (tuning hooks CNERF_SG_NT / CNERF_SG_SPLITS)."""
import os, sys, time, subprocess, json
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    import torch
    sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
    from customnerf_amd.sd import ops, pack
    def timeit(f, n=30, w=5):
        for _ in range(w): f()
        torch.cuda.synchronize(); t = time.time()
        for _ in range(n): f()
        torch.cuda.synchronize(); return (time.time() - t) / n
    out = {}
    for B, C, H, Co in [(2, 320, 64, 320), (2, 640, 64, 320), (2, 960, 64, 320), (2, 320, 32, 640), (2, 640, 32, 640), (2, 1280, 32, 640), (2, 1920, 32, 640),
                        (2, 640, 16, 1280), (2, 1280, 16, 1280), (2, 2560, 16, 1280), (2, 1280, 8, 1280), (2, 2560, 8, 1280)]:
        x = torch.randn(B, H, H, C, device="cuda").half(); w = pack.pack_conv(torch.randn(Co, C, 3, 3) / (9 * C) ** 0.5).cuda()
        t = timeit(lambda: ops.conv2d(x, w, None, 3))
        out[f"conv B{B} C{C} H{H} Co{Co}"] = (t * 1e6, 2 * B * H * H * Co * 9 * C / t / 1e12)
    for M, N, K in [(8192, 320, 320), (8192, 2560, 320), (8192, 320, 1280), (2048, 640, 640), (2048, 5120, 640), (2048, 640, 2560), (512, 1280, 1280), (512, 10240, 1280),
                    (512, 1280, 5120), (154, 320, 768), (154, 1280, 768)]:
        x = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
        t = timeit(lambda: ops.linear(x, w))
        out[f"dense M{M} N{N} K{K}"] = (t * 1e6, 2 * M * N * K / t / 1e12)
    print("RESULT " + json.dumps(out))
    sys.exit(0)
res = {}
auto_only = len(sys.argv) > 1 and sys.argv[1] == 'auto'
for nt in ((0,) if auto_only else (0, 1, 2)):
    for sp in ((0,) if auto_only else (0, 1, 2, 3, 4, 6, 8)):
        env = dict(os.environ)
        if nt: env['CNERF_SG_NT'] = str(nt)
        if sp: env['CNERF_SG_SPLITS'] = str(sp)
        o = subprocess.run([sys.executable, __file__, 'child'], env=env, capture_output=True, text=True).stdout
        line = [l for l in o.splitlines() if l.startswith('RESULT ')]
        if line: res[(nt, sp)] = json.loads(line[0][7:])
keys = list(res[(0, 0)].keys())
for k in keys:
    base = res[(0, 0)][k]
    best = min(res.items(), key=lambda kv: kv[1][k][0])
    row = ' '.join(f"{res[c][k][0]:6.0f}" for c in sorted(res))
    print(f"{k:28s} auto {base[0]:7.1f} us {base[1]:6.1f} TF | best nt={best[0][0]} sp={best[0][1]} {best[1][k][0]:7.1f} us ({base[0]/best[1][k][0]:.2f}x) | {row}")
print("columns:", sorted(res))
