"""GPU parity of the renderer-level path: ray generation, Adam, NeRFRenderer.run with the closed-form field against the
reference's golden vectors, the real field (grid + MLPs) against the oracle field, run()/run_cuda() end to end incl.
gradients.  Tolerance for rendered rgb/depth/weights: 1e-4 absolute in fp32 (north_star)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import c_oracle as co          # noqa: E402
from oracle import torch_oracle as to      # noqa: E402
from oracle.toy_field import ToyField      # noqa: E402


def cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_generate_rays_against_reference_golden(golden):
    from customnerf_amd.nerf.provider_utils import generate_rays, get_rays
    g = golden("rays")
    for tag in ("32", "64", "24x40"):
        fx, fy, cx, cy, H, W = g[f"get_rays_{tag}__intr"]
        H, W = int(H), int(W)
        pose = torch.eye(4).unsqueeze(0).clone()
        pose[0, :3, :4] = T(g[f"get_rays_{tag}__c2w"])
        r = get_rays(pose.cuda(), (fx, fy, cx, cy), H, W)
        np.testing.assert_allclose(r['rays_d'].cpu().numpy(), g[f"get_rays_{tag}__d"], rtol=0, atol=1e-6)
        np.testing.assert_array_equal(r['rays_o'].cpu().numpy(), g[f"get_rays_{tag}__o"])
        c2w = T(g[f"gen_rays_{tag}__c2w"])[None].cuda()
        for level in (1, 2):
            o, d = generate_rays(c2w, fx, fy, cx, cy, H, W, level, 'nerfstudio')
            np.testing.assert_array_equal(o[0].cpu().numpy(), g[f"gen_rays_{tag}_l{level}__o"])
            np.testing.assert_allclose(d[0].cpu().numpy(), g[f"gen_rays_{tag}_l{level}__d"], rtol=0, atol=1e-6)


def test_adam_step_matches_torch():
    from customnerf_amd.optim import adam_step
    torch.manual_seed(0)
    n = 100003
    p = torch.randn(n).cuda(); g = torch.randn(n).cuda() * 1e-3
    ref_p = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref_p], lr=5e-3, betas=(0.9, 0.99), eps=1e-15)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    ph = torch.empty(n, dtype=torch.half).cuda()
    for step in range(1, 6):
        gi = g * step
        ref_p.grad = gi.clone()
        opt.step()
        gbuf = (gi * 128.0).clone()                       # as if produced under a 128x loss scale
        adam_step(p, gbuf, m, v, lr=5e-3, betas=(0.9, 0.99), eps=1e-15, step=step, grad_scale_inv=1 / 128.0, zero_grad=True, p_half=ph)
        assert torch.all(gbuf == 0)
    np.testing.assert_allclose(p.cpu().numpy(), ref_p.detach().cpu().numpy(), rtol=1e-5, atol=1e-6)
    assert torch.equal(ph, p.half())


CASES = {
    "train_T8": dict(training=True, kw=dict(num_steps=8, upsample_steps=8, perturb=True), opt={}),
    "train_T64": dict(training=True, kw=dict(num_steps=64, upsample_steps=64, perturb=True), opt={}),
    "eval_T64": dict(training=False, kw=dict(num_steps=64, upsample_steps=64, perturb=False), opt={}),
    "train_T16_hardmask": dict(training=True, kw=dict(num_steps=16, upsample_steps=16, perturb=True), opt=dict(soft_mask=False)),
    "train_T16_detach": dict(training=True, kw=dict(num_steps=16, upsample_steps=16, perturb=True), opt=dict(detach_bg=True, detach_mask_from_field=True)),
}


def _toy_renderer(**opt_kw):
    from customnerf_amd.nerf.renderer import NeRFRenderer
    from customnerf_amd.scene import make_opt

    class ToyRenderer(NeRFRenderer):
        def __init__(self, opt):
            super().__init__(opt)
            self.f = ToyField()

        def forward(self, x, d, *a, **k):
            return self.f(x, d)

        def density(self, x):
            return self.f.density(x)
    return ToyRenderer(make_opt(**opt_kw)).cuda()


@pytest.mark.parametrize("tag", list(CASES))
def test_run_matches_reference_golden(golden, tag):
    """NeRFRenderer.run on the GPU (HIP near_far + renderer logic) replaying the reference's own RNG draws."""
    g = golden("run")
    c = CASES[tag]
    st = int(g[f"{tag}__stride"])
    rays_o, rays_d = T(g["rays_o"])[:, ::st].contiguous().cuda(), T(g["rays_d"])[:, ::st].contiguous().cuda()
    draws = {"light": T(g[f"{tag}__light"])}
    if f"{tag}__z" in g:
        draws["z"] = T(g[f"{tag}__z"])
    if f"{tag}__u" in g:
        draws["u"] = T(g[f"{tag}__u"])
    model = _toy_renderer(**c["opt"])
    model.train(c["training"])
    res = model.run(rays_o, rays_d, _draws=draws, **c["kw"])
    for k in ("image", "depth", "render_mask", "weights_sum", "weights"):
        np.testing.assert_allclose(res[k].cpu().numpy(), g[f"{tag}__{k}"], rtol=0, atol=1e-4, err_msg=f"{tag}:{k}")
    np.testing.assert_array_equal(res["mask"].cpu().numpy(), g[f"{tag}__mask"])
    for sub in ("fg", "bg"):
        for k in ("image", "depth", "render_mask", "weights_sum"):
            np.testing.assert_allclose(res[sub][k].cpu().numpy(), g[f"{tag}__{sub}_{k}"], rtol=0, atol=1e-4, err_msg=f"{tag}:{sub}.{k}")


def _fields(seed=0, **kw):
    """A NeRFNetwork (product, GPU) and a FieldRef (oracle, CPU) with identical parameters, cfg1-like geometry."""
    from customnerf_amd.nerf.network_grid import NeRFNetwork
    from customnerf_amd.scene import make_opt
    from customnerf_amd import tcnn
    tcnn.set_default_dtype(torch.float32)
    geo = dict(num_levels=4, log2_hashmap_size=19, desired_resolution=2048, n_hidden_geo=1)
    geo.update(kw)
    opt = make_opt(cuda_ray=geo.pop("cuda_ray", False), **geo)
    model = NeRFNetwork(opt).cuda()
    ref = to.FieldRef(bound=opt.bound, num_levels=opt.num_levels, level_dim=2, base_resolution=16, log2_hashmap_size=opt.log2_hashmap_size,
                      desired_resolution=opt.desired_resolution, gridtype='hash', n_hidden_geo=opt.n_hidden_geo, seed=seed)
    g = torch.Generator().manual_seed(seed + 100)
    with torch.no_grad():
        ref.pos_en.embeddings.copy_((torch.rand(ref.pos_en.embeddings.shape, generator=g) * 2 - 1) * 0.5)   # U(-.5,.5): visible features
        model.pos_en.embeddings.copy_(ref.pos_en.embeddings.cuda())
        model.network.params.copy_(ref.network.cuda())
        model.density_network.params.copy_(ref.density_network.cuda())
        model.rgb_network.params.copy_(ref.rgb_network.cuda())
    return model, ref, opt


def test_field_forward_backward_vs_oracle():
    model, ref, opt = _fields()
    rng = np.random.default_rng(0)
    P = 3000
    x = ((rng.random((P, 3)) * 2 - 1) * 1.9).astype(np.float32)
    d = rng.standard_normal((P, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    s_ref, c_ref, _ = ref(T(x), T(d))
    s, c, _ = model(cuda(x), cuda(d))
    np.testing.assert_allclose(s.detach().cpu().numpy(), s_ref.detach().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(c.detach().cpu().numpy(), c_ref.detach().numpy(), rtol=0, atol=1e-5)
    np.testing.assert_allclose(model.density(cuda(x))['sigma'].detach().cpu().numpy(), s_ref.detach().numpy(), rtol=1e-4, atol=1e-5)
    gs = rng.standard_normal(P).astype(np.float32) * 0.1
    gc = rng.standard_normal((P, 4)).astype(np.float32)
    torch.autograd.backward([s_ref, c_ref], [T(gs), T(gc)])
    torch.autograd.backward([s, c], [cuda(gs), cuda(gc)])
    for name, a, b in (("grid", model.pos_en.embeddings.grad, ref.pos_en.embeddings.grad), ("net", model.network.params.grad, ref.network.grad),
                       ("den", model.density_network.params.grad, ref.density_network.grad), ("rgb", model.rgb_network.params.grad, ref.rgb_network.grad)):
        b = b.numpy()
        np.testing.assert_allclose(a.cpu().numpy(), b, rtol=1e-3, atol=1e-4 * max(1.0, float(np.abs(b).max())), err_msg=name)


def test_run_end_to_end_vs_oracle_cfg1():
    """cfg1 (32x32 view, L=4 grid, one-hidden-layer MLPs): full run() forward + loss + backward vs the CPU oracle."""
    from customnerf_amd import scene as sc
    model, ref, opt = _fields()
    model.train()
    H = W = 32
    pose = torch.eye(4).unsqueeze(0).clone()
    pose[0, :3, :4] = T(sc.camera_pose(1, opencv=True))
    o, d = to.get_rays(pose, sc.intrinsics(H, W), H, W)
    g = torch.Generator().manual_seed(5)
    draws = dict(light=torch.randn(3, generator=g), z=torch.rand(H * W, 16, generator=g), u=torch.rand(H * W, 16, generator=g))
    kw = dict(num_steps=16, upsample_steps=16, perturb=True)
    r_ref = to.run(ref, o, d, torch.tensor([-2.0, -2, -2, 2, 2, 2]), opt.min_near, training=True, draws=draws, **kw)
    r = model.run(o.cuda(), d.cuda(), _draws=draws, **kw)
    for k in ("image", "depth", "render_mask", "weights_sum", "weights"):
        np.testing.assert_allclose(r[k].detach().cpu().numpy(), r_ref[k].detach().numpy(), rtol=0, atol=1e-4, err_msg=k)
    # fg/bg split the density with sigmoid((conf - thr) * 100) (renderer.py:387): the x100 gain amplifies fp32 rounding
    # differences of the confidence channel (different matmul summation order GPU vs CPU) by two orders of
    # magnitude, so these two composites are compared at 1e-3; the unsplit image above holds 1e-4.
    for sub in ("fg", "bg"):
        for k in ("image", "depth", "weights_sum"):
            np.testing.assert_allclose(r[sub][k].detach().cpu().numpy(), r_ref[sub][k].detach().numpy(), rtol=0, atol=1e-3, err_msg=sub + k)
    rgb_gt, m_gt = sc.targets(1, H, W, seed=3)

    def loss_of(res, dev):
        return ((res['image'].reshape(-1, 3) - rgb_gt[0].to(dev)) ** 2).mean() + 0.01 * ((res['render_mask'].reshape(-1) - m_gt[0].reshape(-1).to(dev)) ** 2).mean()
    l_ref, l = loss_of(r_ref, 'cpu'), loss_of(r, 'cuda')
    assert abs(l.item() - l_ref.item()) < 1e-5
    l_ref.backward(); l.backward()
    for name, a, b in (("grid", model.pos_en.embeddings.grad, ref.pos_en.embeddings.grad), ("net", model.network.params.grad, ref.network.grad),
                       ("den", model.density_network.params.grad, ref.density_network.grad), ("rgb", model.rgb_network.params.grad, ref.rgb_network.grad)):
        b = b.numpy()
        np.testing.assert_allclose(a.cpu().numpy(), b, rtol=2e-3, atol=2e-4 * max(1e-3, float(np.abs(b).max())), err_msg=name)


def test_run_cuda_train_and_eval_vs_oracle():
    from customnerf_amd import scene as sc
    model, ref, opt = _fields(cuda_ray=True)
    grid = sc.sphere_density_grid(2, 128, 2.0, 1.0, 100.0)
    bitfield = co.packbits(grid, 10.0)
    model.density_grid.copy_(cuda(grid))
    model.density_bitfield.copy_(cuda(bitfield))
    H = W = 32
    pose = torch.eye(4).unsqueeze(0).clone()
    pose[0, :3, :4] = T(sc.camera_pose(6, opencv=True))
    o, d = to.get_rays(pose, sc.intrinsics(H, W), H, W)
    aabb = torch.tensor([-2.0, -2, -2, 2, 2, 2])
    noises = torch.rand(H * W, generator=torch.Generator().manual_seed(9))
    # --- training branch
    model.train()
    r_ref = to.run_cuda_train(ref, o, d, aabb, 2.0, bitfield, 2, 128, noises.numpy(), 0, 1024, 1e-4)
    r = model.run_cuda(o.cuda(), d.cuda(), perturb=True, force_all_rays=True, _noises=noises.cuda())
    np.testing.assert_array_equal(r['rays'].cpu().numpy(), r_ref['rays'])          # compacted ray indices: bit-exact
    assert model.step_counter[0].cpu().numpy().tolist() == r_ref['counter'].tolist()
    for k in ("image", "depth", "weights_sum"):
        np.testing.assert_allclose(r[k].detach().cpu().numpy().reshape(r_ref[k].shape), r_ref[k].detach().numpy(), rtol=0, atol=1e-4, err_msg=k)
    np.testing.assert_array_equal(r['mask'].cpu().numpy().reshape(-1), r_ref['mask'].numpy())
    l_ref = (r_ref['image'] ** 2).sum() + r_ref['weights_sum'].sum()
    l = (r['image'] ** 2).sum() + r['weights_sum'].sum()
    l_ref.backward(); l.backward()
    for name, a, b in (("grid", model.pos_en.embeddings.grad, ref.pos_en.embeddings.grad), ("net", model.network.params.grad, ref.network.grad),
                       ("den", model.density_network.params.grad, ref.density_network.grad), ("rgb", model.rgb_network.params.grad, ref.rgb_network.grad)):
        b = b.numpy()
        np.testing.assert_allclose(a.cpu().numpy(), b, rtol=2e-3, atol=2e-4 * max(1e-3, float(np.abs(b).max())), err_msg=name)
    # --- inference branch
    model.eval()
    with torch.no_grad():
        e = model.run_cuda(o.cuda(), d.cuda(), perturb=False, T_thresh=1e-4)
    e_ref = to.run_cuda_eval(ref, o, d, aabb, 2.0, bitfield, 2, 128, 0, 1024, 1e-4)
    for k in ("image", "depth", "weights_sum"):
        np.testing.assert_allclose(e[k].cpu().numpy().reshape(e_ref[k].shape), e_ref[k], rtol=0, atol=1e-4, err_msg="eval " + k)
    # training and inference composites agree with each other too (same samples, same field)
    np.testing.assert_allclose(e['image'].cpu().numpy().reshape(-1, 3), r_ref['image'].detach().numpy(), rtol=0, atol=5e-3)


def test_update_extra_state_builds_occupancy():
    """update_extra_state (renderer.py:1658-1715): morton3D + density query + EMA + packbits on the device."""
    model, ref, opt = _fields(cuda_ray=True)
    torch.manual_seed(0)
    model.update_extra_state()
    dg = model.density_grid.cpu().numpy()
    assert dg.min() >= 0 and model.mean_density > 0
    thr = min(model.mean_density, model.density_thresh)
    np.testing.assert_array_equal(model.density_bitfield.cpu().numpy(), co.packbits(dg, thr))
    # the gaussian blob (network_grid.py:150-156) makes the centre dense: the centre cell must be occupied in cascade 0
    centre = co.morton3D(np.array([[64, 64, 64]], np.int32))[0]
    assert dg[0, centre] > thr
