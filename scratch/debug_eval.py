import sys; sys.path.insert(0,'.')
import numpy as np, torch
from oracle import torch_oracle as to, c_oracle as co
from customnerf_amd import scene as sc, raymarching as rm
sys.path.insert(0,'tests')
from test_gpu_render import _fields, cuda, T
model, ref, opt = _fields(cuda_ray=True)
grid = sc.sphere_density_grid(2, 128, 2.0, 1.0, 100.0)
bitfield = co.packbits(grid, 10.0)
model.density_bitfield.copy_(cuda(bitfield))
H=W=32
pose = torch.eye(4).unsqueeze(0).clone()
pose[0, :3, :4] = T(sc.camera_pose(6, opencv=True))
o, d = to.get_rays(pose, sc.intrinsics(H, W), H, W)
o=o.reshape(-1,3).contiguous(); d=d.reshape(-1,3).contiguous()
aabb = np.array([-2.0, -2, -2, 2, 2, 2],np.float32)
N=o.shape[0]
nears,fars = co.near_far_from_aabb(o.numpy(), d.numpy(), aabb, 0.2)
ws_r, dep_r, img_r = np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros((N, 3), np.float32)
alive_r, t_r = np.arange(N, dtype=np.int32), nears.copy()
ws, dep, img = torch.zeros(N).cuda(), torch.zeros(N).cuda(), torch.zeros(N, 3).cuda()
alive, t = torch.arange(N, dtype=torch.int32).cuda(), cuda(nears)
alive_next = torch.empty_like(alive); count = torch.zeros(1, dtype=torch.int32).cuda()
og, dg, bf, ng, fg = o.cuda(), d.cuda(), cuda(bitfield), cuda(nears), cuda(fars)
model.eval()
step=0; it=0
while step<1024:
    n_alive = alive_r.shape[0]
    if n_alive<=0: break
    n_step = max(min(N//n_alive,8),1)
    xr,dr,lr = co.march_rays(n_alive,n_step,alive_r,t_r,o.numpy(),d.numpy(),2.0,bitfield,2,128,nears,fars,128,None,0,1024)
    x,dd,l = rm.march_rays(n_alive,n_step,alive,t,og,dg,2.0,bf,2,128,ng,fg,128,False,0,1024)
    assert np.array_equal(x.cpu().numpy(), xr), it
    with torch.no_grad():
        s_ref,c_ref,_ = ref(torch.from_numpy(xr), torch.from_numpy(dr))
        s,c,_ = model(x,dd)
    ds = (s.cpu()-s_ref).abs().max().item(); dc=(c.cpu()-c_ref).abs().max().item()
    co.composite_rays(n_alive,n_step,alive_r,t_r,s_ref.numpy(),c_ref[...,:3].contiguous().numpy(),lr,ws_r,dep_r,img_r,1e-4)
    rm.composite_rays(n_alive,n_step,alive,t,s,c,l,ws,dep,img,1e-4)
    alive_r = np.ascontiguousarray(alive_r[alive_r>=0])
    rm.compact_rays_alive(alive,n_alive,alive_next,count); alive,alive_next = alive_next,alive
    k=int(count.item())
    di = np.abs(img.cpu().numpy()-img_r).max(); dw=np.abs(ws.cpu().numpy()-ws_r).max()
    same = (k==alive_r.shape[0]) and np.array_equal(alive[:k].cpu().numpy(), alive_r)
    if it<6 or di>1e-5 or not same:
        print(it, n_alive, n_step, "dsig",ds,"dc",dc,"dimg",di,"dws",dw,"alive_same",same, k, alive_r.shape[0])
    if not same: break
    step+=n_step; it+=1
print("final", np.abs(img.cpu().numpy()-img_r).max())
