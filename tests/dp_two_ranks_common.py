"""shared by tests/test_gpu_dp_two_ranks.py and its child ranks: the same seeded field, views and jitter draws in every process"""
import torch

H = W = 64                                                      # 4096 rays x 32 samples x 16 levels >= 2^20 pairs: the binned (bit-reproducible) scatter
V = 4
KW = dict(num_steps=16, upsample_steps=16, dt_gamma=0, max_steps=1024)


def build():
    from customnerf_amd import scene as sc, tcnn
    from customnerf_amd.nerf.network_grid import NeRFNetwork
    from customnerf_amd.nerf.provider_utils import generate_rays
    tcnn.set_default_dtype(torch.float16)
    torch.manual_seed(0)
    opt = sc.make_opt(fp16=True, num_steps=16, upsample_steps=16, iters=100)
    model = NeRFNetwork(opt).cuda()
    c2w = torch.from_numpy(sc.poses(V)).cuda()
    ro, rd = generate_rays(c2w, *sc.intrinsics(H, W), H, W, 1.0, 'nerfstudio')
    ro, rd = ro.view(V, 1, H * W, 3), rd.view(V, 1, H * W, 3)
    rgb, mask = sc.targets(V, H, W)
    rgb, mask = rgb.cuda(), mask.cuda()
    views = [(ro[v].contiguous(), rd[v].contiguous(), rgb[v].contiguous(), mask[v].contiguous()) for v in range(V)]
    return model, opt, views


def view_draws(v):
    g = torch.Generator(device="cuda").manual_seed(1000 + v)
    return dict(z=torch.rand(H * W, KW["num_steps"], device="cuda", generator=g), u=torch.rand(H * W, KW["upsample_steps"], device="cuda", generator=g))
