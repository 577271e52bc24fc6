import torch, sys
sys.path.insert(0, '/root/repo')
from customnerf_amd.sd import arch, ops
from customnerf_amd.sd.unet import UNet
cfg = arch.UNET_TINY
sd = arch.random_state_dict(arch.unet_params(cfg), seed=3)
net = UNet(cfg, sd, "cuda")
g = torch.Generator().manual_seed(0)
x = torch.zeros(2, 32, 32, 8, dtype=torch.float16); x[..., :4] = torch.randn(2, 32, 32, 4, generator=g).half(); x = x.cuda()
ctx = torch.randn(2, 77, 96, generator=g).half().cuda()
t = torch.tensor([481.0, 481.0]).cuda()
G, eps = 32, 1e-5

def stage(n):
    def f():
        temb = ops.timestep_embedding(t, 128)
        temb = ops.linear(ops.linear(temb, net.t1w, bias=net.t1b, act=ops.ACT_SILU), net.t2w, bias=net.t2b)
        if n == 0: return temb
        ta = ops.silu(temb)
        h = ops.conv2d(x, net.ciw, net.cib, 3)
        if n == 1: return h
        r = net.down[0][0][0]
        h1, _ = ops.groupnorm(h, r.n1w, r.n1b, G, eps, True)
        if n == 2: return h1
        tb = ops.linear(ta, r.tw, bias=r.tb, out32=True)
        if n == 3: return tb
        h2 = ops.conv2d(h1, r.c1w, r.c1b, 3, bias_rows=tb)
        if n == 4: return h2
        h = r(h, ta, G, eps)
        if n == 5: return h
        h = net.down[0][1][0](h, ctx, G)
        if n == 6: return h
        return net.forward(x, t, ctx)
    return f

for n in range(8):
    f = stage(n)
    with torch.no_grad():
        e1 = f().clone(); e2 = f().clone()
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s): f()
        torch.cuda.current_stream().wait_stream(s)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            o = f()
        gr.replay(); torch.cuda.synchronize()
        g1 = o.clone()
        gr.replay(); torch.cuda.synchronize()
        g2 = o.clone()
    print(n, 'eager-eager', float((e1.float()-e2.float()).abs().max()), 'eager-graph', float((e1.float()-g1.float()).abs().max()), 'graph-graph', float((g1.float()-g2.float()).abs().max()), 'scale', float(e1.float().abs().max()))
