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
need=ctypes.c_uint64(0)
lib.cnerf_grid_encode_backward_workspace_bytes(enc._offsets_host.ctypes.data,B,3,C,L,L,S,16,1,ctypes.addressof(need))
ws=torch.empty(need.value+256,dtype=torch.uint8,device='cuda')
g=torch.randn(L,B,C,device='cuda').half()
ge=torch.zeros_like(enc.embeddings)
def run(x, nl, label, use_ws=True):
    for _ in range(2):
        check(lib.cnerf_grid_encode_backward(ptr(g),ptr(x),enc._offsets_host.ctypes.data,ptr(ge),B,3,C,L,nl,S,16,None,None,0,0,0,1,ptr(ws) if use_ws else None,ws.numel() if use_ws else 0,stream()))
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        check(lib.cnerf_grid_encode_backward(ptr(g),ptr(x),enc._offsets_host.ctypes.data,ptr(ge),B,3,C,L,nl,S,16,None,None,0,0,0,1,ptr(ws) if use_ws else None,ws.numel() if use_ws else 0,stream()))
    e1.record(); torch.cuda.synchronize()
    print(f"{label:40s} nl={nl:2d}: {e0.elapsed_time(e1)/3:8.3f} ms")
for nl in (1,2,3,5,8,16):
    run(x_ray, nl, "binned ray-structured points")
for nl in (5,16):
    run(x_rand, nl, "binned uniform random points")
run(x_ray, 5, "atomic ray-structured", False)
run(x_rand, 16, "atomic uniform random", False)
