import sys, ctypes; sys.path.insert(0,'.')
import numpy as np, torch
from customnerf_amd.gridencoder import GridEncoder, grid as G
from customnerf_amd._lib import lib, ptr, stream, check
from customnerf_amd import scene as sc
from customnerf_amd.nerf.provider_utils import generate_rays
from customnerf_amd import raymarching
torch.manual_seed(0)
enc = GridEncoder(num_levels=16, log2_hashmap_size=19, desired_resolution=2048, gridtype='hash').cuda()
H=W=128
o,d = generate_rays(torch.from_numpy(sc.poses(8)[:1]).cuda(), *sc.intrinsics(H,W), H, W)
o=o.view(-1,3); d=d.view(-1,3)
aabb=torch.tensor([-2.,-2,-2,2,2,2]).cuda()
n,f = raymarching.near_far_from_aabb(o,d,aabb,0.01)
z = n[:,None] + (f-n)[:,None]*torch.linspace(0,1,128,device='cuda')[None]
xyz = (o[:,None]+d[:,None]*z[...,None]).clamp(-2,2).reshape(-1,3)
x_ray = ((xyz+2)/4).contiguous()
x_rand = torch.rand_like(x_ray)
B=x_ray.shape[0]; L=16; C=2
S=float(np.log2(enc.per_level_scale))
def run(x, half, label, nl=16):
    table = enc.half_table() if half else enc.embeddings.detach()
    out = torch.empty(L,B,C,device='cuda',dtype=table.dtype)
    def go():
        check(lib.cnerf_grid_encode_forward(ptr(x),ptr(table),enc._offsets_host.ctypes.data,ptr(out),B,3,C,L,nl,S,16,None,0,0,0,1 if half else 0,stream()))
    for _ in range(3): go()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): go()
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/10
    bpp = nl*8*2*(2 if half else 4)+12+nl*2*(2 if half else 4)
    print(f"{label:34s} half={half} nl={nl:2d}: {ms:7.3f} ms  {B*bpp/ms/1e6:8.1f} GB/s algorithmic")
for half in (True, False):
    run(x_ray, half, "ray-structured 2.1M pts")
    run(x_rand, half, "uniform random 2.1M pts")
run(x_ray, True, "ray-structured levels 0-4", 5)
