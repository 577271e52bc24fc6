"""CPU: pin the oracle (oracle/) against the golden vectors produced by the reference's own Python
(tests/golden/make_golden.py).  SURVEY.md §8c items (1)-(7)."""
import numpy as np
import torch

from oracle import torch_oracle as to
from oracle import c_oracle as co
from oracle.toy_field import ToyField


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_sample_pdf(golden):
    g = golden("sample_pdf")
    out = to.sample_pdf(T(g["bins"]), T(g["weights"]), 16, det=True)
    np.testing.assert_allclose(out.numpy(), g["out_det"], rtol=0, atol=1e-6)
    out = to.sample_pdf(T(g["bins"]), T(g["weights"]), 16, det=False, u=T(g["u"]))
    np.testing.assert_allclose(out.numpy(), g["out_rnd"], rtol=0, atol=1e-6)


def test_fisheye_rays(golden):
    """oracle restatement of the OPENCV_FISHEYE ray branch (provider.py:421-433, provider_utils.py:128-234) against what the reference's
    own radial_and_tangential_undistort produced"""
    g = golden("rays_fisheye")
    for tag in ("24x40", "32"):
        fx, fy, cx, cy, H, W, level = g[f"{tag}__intr"]
        H, W = int(H), int(W)
        xs = torch.linspace(0, W * level - 1, W)
        ys = torch.linspace(0, H * level - 1, H)
        x, y = torch.meshgrid(xs, ys, indexing='ij')
        coord = torch.stack([((x + 0.5).reshape(-1) - cx) / fx, -((y + 0.5).reshape(-1) - cy) / fy], -1)
        np.testing.assert_allclose(to.undistort(coord, g[f"{tag}__dist"]).numpy(), g[f"{tag}__undist"], rtol=0, atol=1e-6)
        o, d = to.generate_rays_fisheye(T(g[f"{tag}__c2w"])[None], fx, fy, cx, cy, H, W, level, g[f"{tag}__dist"])
        np.testing.assert_array_equal(o[0].numpy(), g[f"{tag}__o"])
        np.testing.assert_allclose(d[0].numpy(), g[f"{tag}__d"], rtol=0, atol=1e-6)


def test_trunc_exp(golden):
    g = golden("trunc_exp")
    x = T(g["x"]).requires_grad_(True)
    y = to.trunc_exp(x)
    y.backward(T(g["g"]))
    np.testing.assert_array_equal(y.detach().numpy(), g["y"])
    np.testing.assert_array_equal(x.grad.numpy(), g["gx"])


def test_embedder_and_safe_normalize(golden):
    g = golden("embedder")
    assert int(g["out_dim"]) == 27
    np.testing.assert_array_equal(to.freq_embed(T(g["d"])).numpy(), g["out"])
    np.testing.assert_array_equal(to.safe_normalize(T(g["sn_in"])).numpy(), g["sn_out"])


def test_grid_offsets(golden):
    g = golden("grid_offsets")
    cfgs = {"hash_L16_T19_2048": dict(num_levels=16, log2_hashmap_size=19, desired_resolution=2048),
            "tiled_L16_T21_8192": dict(num_levels=16, log2_hashmap_size=21, desired_resolution=8192),
            "hash_L4_T19_2048": dict(num_levels=4, log2_hashmap_size=19, desired_resolution=2048),
            "hash_default": dict()}
    for tag, kw in cfgs.items():
        off, pls = to.grid_offsets(**kw)
        np.testing.assert_array_equal(off, g[tag + "__offsets"])
        assert float(pls) == float(g[tag + "__pls"])
        assert int(off[-1]) == int(g[tag + "__shape"][0])
        assert int(g[tag + "__n_params"]) == int(off[-1]) * 2
        assert float(g[tag + "__absmax"]) <= 1e-4
    # SURVEY.md §8 sizes
    off, _ = to.grid_offsets(**cfgs["hash_L16_T19_2048"])
    assert int(off[-1]) == 6119864
    off, _ = to.grid_offsets(**cfgs["tiled_L16_T21_8192"])
    assert int(off[-1]) == 23967296


def test_rays(golden):
    g = golden("rays")
    for tag in ("32", "64", "24x40"):
        c2w = T(g[f"get_rays_{tag}__c2w"])
        fx, fy, cx, cy, H, W = g[f"get_rays_{tag}__intr"]
        H, W = int(H), int(W)
        pose = torch.eye(4).unsqueeze(0).clone()
        pose[0, :3, :4] = c2w
        o, d = to.get_rays(pose, (fx, fy, cx, cy), H, W)
        np.testing.assert_array_equal(o.numpy(), g[f"get_rays_{tag}__o"])
        np.testing.assert_array_equal(d.numpy(), g[f"get_rays_{tag}__d"])
        c2w = T(g[f"gen_rays_{tag}__c2w"])
        for level in (1, 2):
            o, d = to.generate_rays(c2w[None], fx, fy, cx, cy, H, W, level)
            np.testing.assert_array_equal(o[0].numpy(), g[f"gen_rays_{tag}_l{level}__o"])
            np.testing.assert_allclose(d[0].numpy(), g[f"gen_rays_{tag}_l{level}__d"], rtol=0, atol=1e-7)


CASES = {
    "train_T8": dict(training=True, kw=dict(num_steps=8, upsample_steps=8, perturb=True)),
    "train_T64": dict(training=True, kw=dict(num_steps=64, upsample_steps=64, perturb=True)),
    "eval_T64": dict(training=False, kw=dict(num_steps=64, upsample_steps=64, perturb=False)),
    "train_T16_hardmask": dict(training=True, kw=dict(num_steps=16, upsample_steps=16, perturb=True, soft_mask=False)),
    "train_T16_detach": dict(training=True, kw=dict(num_steps=16, upsample_steps=16, perturb=True, detach_bg=True,
                                                    detach_mask_from_field=True)),
}


def test_run_against_reference(golden):
    g = golden("run")
    aabb = torch.tensor([-2.0, -2, -2, 2, 2, 2])
    for tag, c in CASES.items():
        st = int(g[f"{tag}__stride"])
        rays_o, rays_d = T(g["rays_o"])[:, ::st].contiguous(), T(g["rays_d"])[:, ::st].contiguous()
        draws = {"light": T(g[f"{tag}__light"])}
        if f"{tag}__z" in g:
            draws["z"] = T(g[f"{tag}__z"])
        if f"{tag}__u" in g:
            draws["u"] = T(g[f"{tag}__u"])
        res = to.run(ToyField(), rays_o, rays_d, aabb, 0.01, training=c["training"], draws=draws, **c["kw"])
        for k in ("image", "depth", "render_mask", "weights_sum", "weights", "sigma", "rgbs"):
            np.testing.assert_allclose(res[k].numpy(), g[f"{tag}__{k}"], rtol=1e-5, atol=1e-6, err_msg=f"{tag}:{k}")
        np.testing.assert_array_equal(res["mask"].numpy(), g[f"{tag}__mask"])
        np.testing.assert_allclose(res["edit_mask"].float().numpy(), g[f"{tag}__edit_mask"].astype(np.float32), rtol=1e-5, atol=1e-6)
        for sub in ("fg", "bg"):
            for k in ("image", "depth", "render_mask", "weights_sum", "weights"):
                np.testing.assert_allclose(res[sub][k].numpy(), g[f"{tag}__{sub}_{k}"], rtol=1e-5, atol=1e-6,
                                           err_msg=f"{tag}:{sub}.{k}")
        # dropping the output-dead fine density pass (renderer.py:353) changes nothing
        res2 = to.run(ToyField(), rays_o, rays_d, aabb, 0.01, training=c["training"], draws=draws, skip_fine_density=True, **c["kw"])
        np.testing.assert_array_equal(res2["image"].numpy(), res["image"].numpy())


def test_weights_sum_i_and_grads(golden):
    g = golden("weights_sum_i")
    N = g["sigmas"].shape[0]
    for tag, kw in (("plain", {}), ("detach", dict(detach_bg=True, detach_mask_from_field=True))):
        s, c = T(g["sigmas"]).requires_grad_(True), T(g["rgbs"]).requires_grad_(True)
        res = to.weights_sum_i(T(g["sample_dist"]), s, T(g["z"]), T(g["nears"]), T(g["fars"]), c, (1, N), T(g["masks"]),
                               is_all=True, **kw)
        loss = (res['image'] ** 2).sum() + res['weights_sum'].sum() + (res['render_mask'] * 0.3).sum() + res['depth'].sum()
        loss.backward()
        for k in ("image", "depth", "render_mask", "weights_sum", "weights"):
            np.testing.assert_allclose(res[k].detach().numpy(), g[f"{tag}__{k}"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(s.grad.numpy(), g[f"{tag}__grad_sigmas"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(c.grad.numpy(), g[f"{tag}__grad_rgbs"], rtol=1e-5, atol=1e-7)


def test_near_far_c_oracle_matches_reference_path(golden):
    """The C near_far restatement against the numpy slab test the golden run() used (its nears/fars fix `mask`
    and every sample position, so image parity above already depends on it); here directly, incl. misses."""
    g = golden("run")
    o, d = g["rays_o"].reshape(-1, 3), g["rays_d"].reshape(-1, 3)
    aabb = np.array([-0.5, -0.5, -0.5, 0.5, 0.5, 0.5], np.float32)     # small box: many rays miss
    nears, fars = co.near_far_from_aabb(o, d, aabb, 0.05)
    big = np.finfo(np.float32).max
    miss = nears == big
    assert miss.any() and (~miss).any()
    assert np.all(fars[miss] == big)
    hit = ~miss
    assert hit.sum() > 50
    p_near = o[hit] + nears[hit, None] * d[hit]
    p_far = o[hit] + fars[hit, None] * d[hit]
    assert np.all(np.abs(p_far).max(-1) <= 0.5 + 1e-5) and np.all(np.abs(p_near).max(-1) <= 0.5 + 1e-5)
    assert np.all(nears[hit] <= fars[hit]) and np.all(nears[hit] > 2.0)
    # the golden run() itself must be a non-degenerate scene: most rays hit the [-2,2]^3 box
    assert g["train_T8__mask"].mean() > 0.9 and g["train_T8__weights_sum"].max() > 0.5


def _occupancy_rounds(g):
    grid = g("occupancy")
    H, cas = int(grid["grid_size"]), int(grid["cascade"])
    return grid, H, cas


def test_update_extra_state_against_reference(golden):
    """The oracle's occupancy refresh against the reference's own update_extra_state loop (renderer.py:1658-1715, run on a 16^3 grid with the
    toy density; tests/golden/make_golden.py records the jitter draws): two refreshes, the second on the first one's grid.  The one stated
    difference — `2 * coords / (H - 1)` as a multiplication by the float reciprocal (what the GPU the reference runs on computes) against the
    true division of the CPU run that made the vectors — moves a query position by at most one ulp."""
    g, H, cas = _occupancy_rounds(golden)
    grid = g["grid0"]
    for rnd in range(2):
        rand = [T(r) for r in g[f"r{rnd}__rand"]]
        grid, mean, bits = to.update_extra_state(ToyField(), grid, float(g["bound"]), cas, H, float(g["decay"]), float(g["density_thresh"]), rand)
        want = g[f"r{rnd}__grid"]
        assert np.array_equal(grid < 0, want < 0) and np.array_equal(grid[want < 0], want[want < 0])       # invalid cells untouched
        np.testing.assert_allclose(grid, want, rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(mean, float(g[f"r{rnd}__mean_density"]), rtol=1e-6)
        thr = min(float(g[f"r{rnd}__mean_density"]), float(g["density_thresh"]))
        diff = np.unpackbits(bits ^ g[f"r{rnd}__bitfield"], bitorder="little").astype(bool)
        assert not diff[np.abs(want.reshape(-1) - thr) > 1e-4].any()                 # bits may differ only for cells at the threshold
        assert diff.sum() <= 2


SDS_CASES = ["plain", "local", "stage_late", "stage_early", "nonfinite"]


def _poisoned_toy_eps(x, t, ctx):
    from oracle.toy_field import toy_eps
    e = toy_eps(x, t, ctx).clone()
    e[1, 0, 0, 0], e[1, 1, 2, 3], e[0, 2, 1, 1] = float("nan"), float("inf"), float("-inf")
    return e


def test_sds_train_step_against_reference(golden):
    """oracle/sd_oracle.sds_grad against the reference's own StableDiffusion.train_step (nerf/sd.py:115-155, run by make_golden.py with a
    closed-form epsilon predictor and the published scheduler formula): timestep range and `t * t_ratio` truncation, add_noise, the CFG
    combination, (1 - abar_t) weighting, lambda_sd, nan_to_num, and the loss whose latent gradient is the SDS gradient."""
    from oracle import sd_oracle as so
    from oracle.toy_field import toy_eps
    g = golden("sds")
    alphas, text = T(g["alphas_cumprod"]), T(g["text"])
    for tag in SDS_CASES:
        lo, hi = [int(v) for v in g[f"{tag}__randint_lo_hi"]]
        late = bool(g[f"{tag}__stage_time"]) and int(g[f"{tag}__global_step"]) > 1000 / 2
        assert lo == int(1000 * 0.02) and hi == (int(int(1000 * 0.98) * 0.5) if late else int(1000 * 0.98)) + 1
        t = int(int(g[f"{tag}__t_draw"][0]) * float(g[f"{tag}__t_ratio"]))               # (t * t_ratio).to(torch.long): truncation
        lat, noise = T(g[f"{tag}__latents"]), T(g[f"{tag}__noise"])
        grad = so.sds_grad(None, None, lat, text, t, noise, alphas, 100.0, 0.01, eps_fn=_poisoned_toy_eps if tag == "nonfinite" else toy_eps)
        assert torch.isfinite(grad).all()                                                  # nan_to_num
        lat_g = lat.clone().requires_grad_(True)                                           # sd.py:150-152, as oracle.train_step_sd does
        loss = 0.5 * torch.nn.functional.mse_loss(lat_g, (lat_g - grad).detach(), reduction="sum")
        loss.backward()
        want = g[f"{tag}__grad_latents"]
        np.testing.assert_allclose(lat_g.grad.numpy(), want, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(float(loss.detach()), float(g[f"{tag}__loss"]), rtol=1e-5)
        assert float(g[f"{tag}__loss"]) == float(g[f"{tag}__loss_sds"])
        if tag != "nonfinite":
            np.testing.assert_allclose(grad.numpy(), want, rtol=1e-5, atol=1e-7)           # d loss / d latents = grad
    # what the reference does with a non-finite prediction: NaN -> 0; +-inf -> +-FLT_MAX in `grad`, which the mse backward (2 (x - y) first)
    # turns back into an infinite latent gradient — the loss scaler then skips the step
    w = g["nonfinite__grad_latents"]
    assert w[0, 0, 0, 0] == 0.0 and np.isinf(w[0, 1, 2, 3]) and np.isinf(w[0, 2, 1, 1]) and np.isinf(float(g["nonfinite__loss"]))


def test_noise_schedule_matches_the_published_one(golden):
    """customnerf_amd.sd.arch.alphas_cumprod (host logic of the product: the table `train_step` indexes) against the scheduler stand-in the
    reference's train_step ran with: scaled-linear betas 0.00085 .. 0.012 over 1000 steps, cumulative product of (1 - beta)."""
    from customnerf_amd.sd import arch
    np.testing.assert_allclose(arch.alphas_cumprod(1000).numpy(), golden("sds")["alphas_cumprod"], rtol=2e-6)


def test_editing_step_against_reference(golden):
    """oracle/edit_oracle.train_step_editing against the reference's own Trainer_Nerf.train_step_editing / train_step_sd / get_pt
    (nerf/utils_init_nerf.py:243-308, 353-394; tests/golden/editing.npz: toy field with six parameters, closed-form VAE and epsilon predictor):
    rendered image, the SDS and the background term, the total loss and its gradient with respect to the field's parameters, for the global
    (`g_only`) and the local (`l_only`, `local_t_ratio`) branch.  `ori_bg`: the reference raises on its own broadcast (recorded)."""
    import argparse
    from oracle import edit_oracle as eo
    from oracle.toy_field import toy_eps, toy_encode_imgs
    g = golden("editing")
    H, W = int(g["H"]), int(g["W"])
    assert int(g["ori_bg__raises"]) == 1 and "must match" in str(g["ori_bg__message"])
    aabb = torch.tensor([-2.0, -2, -2, 2, 2, 2])
    for tag in ("g_only", "l_only"):
        opt = argparse.Namespace(num_steps=int(g["opt__num_steps"]), upsample_steps=int(g["opt__upsample_steps"]), train_conf=float(g["opt__train_conf"]),
                                 soft_mask=True, conf_thr=float(g["opt__conf_thr"]), detach_bg=False, detach_mask_from_field=False,
                                 min_near=float(g["opt__min_near"]), lambda_sd=float(g["opt__lambda_sd"]), keep_bg=float(g["opt__keep_bg"]),
                                 local_t_ratio=float(g["opt__local_t_ratio"]), cfg=float(g["opt__cfg"]), ori_bg=False)
        theta = T(g["theta_edit"]).clone().requires_grad_(True)
        draws = dict(light=T(g[f"{tag}__light"]), z=T(g[f"{tag}__z"]), u=T(g[f"{tag}__u"]))
        draws_pt = dict(light=T(g[f"{tag}__pt_light"]), z=T(g[f"{tag}__pt_z"]), u=T(g[f"{tag}__pt_u"]))
        loss, ld, out = eo.train_step_editing(ToyField(theta), ToyField(T(g["theta_pre"])), T(g["rays_o"]), T(g["rays_d"]), T(g["rgbs"]), H, W, aabb, opt,
                                              None, None, None, None, T(g["text_z"]), T(g["text_z_fg"]), T(g["alphas_cumprod"]), draws, draws_pt,
                                              "global" if tag == "g_only" else "local", int(g[f"{tag}__t_draw"][0]), None, T(g[f"{tag}__noise"]),
                                              encode_fn=toy_encode_imgs, eps_fn=toy_eps)
        loss.backward()
        np.testing.assert_allclose(out["image"].detach().numpy().reshape(1, H, W, 3).transpose(0, 3, 1, 2), g[f"{tag}__pred_rgb"], rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(float(ld["loss_sds"]), float(g[f"{tag}__loss_sds"]), rtol=2e-4)
        np.testing.assert_allclose(float(ld["loss_bg"]), float(g[f"{tag}__loss_bg"]), rtol=1e-5)
        np.testing.assert_allclose(float(loss.detach()), float(g[f"{tag}__loss"]), rtol=1e-5)
        np.testing.assert_allclose(theta.grad.numpy(), g[f"{tag}__grad_theta"], rtol=2e-4, atol=1e-5)


def test_pretrain_step_against_reference(golden):
    """The reconstruction step — oracle run() + train_rgb * mse(image, rgbs) + train_conf * mse(render_mask, mask) — against the reference's own
    Trainer_Nerf.train_step_pretrain (nerf/utils_init_nerf.py:194-241; tests/golden/editing.npz, toy field with six parameters)."""
    import torch.nn.functional as F
    g = golden("editing")
    aabb = torch.tensor([-2.0, -2, -2, 2, 2, 2])
    for tag in ("conf", "conf2", "batch"):                       # "batch": --batch_rays subsampling (:210-215) with the recorded index draw
        theta = T(g["theta_edit"]).clone().requires_grad_(True)
        draws = dict(light=T(g[f"pre_{tag}__light"]), z=T(g[f"pre_{tag}__z"]), u=T(g[f"pre_{tag}__u"]))
        sel = torch.from_numpy(g[f"pre_{tag}__select_inds"]) if int(g[f"pre_{tag}__batch_rays"]) else torch.arange(T(g["rays_o"]).reshape(-1, 3).shape[0])
        pick = lambda a, c: T(a).reshape(1, -1, c)[:, sel]
        out = to.run(ToyField(theta), pick(g["rays_o"], 3), pick(g["rays_d"], 3), aabb, float(g["opt__min_near"]), num_steps=int(g["opt__num_steps"]),
                     upsample_steps=int(g["opt__upsample_steps"]), perturb=True, training=True, train_conf=float(g[f"pre_{tag}__train_conf"]), draws=draws)
        loss_c = float(g[f"pre_{tag}__train_rgb"]) * F.mse_loss(out["image"].reshape(-1, 3), pick(g["rgbs"], 3).reshape(-1, 3))
        loss_m = float(g[f"pre_{tag}__train_conf"]) * F.mse_loss(out["render_mask"].reshape(-1), pick(g["mask"], 1).reshape(-1))
        (loss_c + loss_m).backward()
        np.testing.assert_allclose(out["image"].detach().numpy(), g[f"pre_{tag}__pred_rgb"], rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(out["weights_sum"].detach().numpy().reshape(1, -1).clip(1e-5, 1 - 1e-5), g[f"pre_{tag}__mask_volume"], rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(float(loss_c.detach()), float(g[f"pre_{tag}__loss_c"]), rtol=1e-5)
        np.testing.assert_allclose(float(loss_m.detach()), float(g[f"pre_{tag}__loss_m"]), rtol=1e-5)
        np.testing.assert_allclose(theta.grad.numpy(), g[f"pre_{tag}__grad_theta"], rtol=2e-4, atol=1e-7)


def _field_table(n):
    idx = torch.arange(n, dtype=torch.float64)
    return torch.stack([torch.sin(idx * 0.37) * 0.5, torch.cos(idx * 0.11 + 1.3) * 0.5], -1).float()


def test_field_glue_against_reference(golden):
    """oracle FieldRef (the field as the oracle states it) against the reference's own NeRFNetwork.forward / density (nerf/network_grid.py:66-193)
    with its real GridEncoder wrapper, get_encoder, get_embedder and trunc_exp; the native kernel and tinycudann underneath were this build's
    restatements when tests/golden/field.npz was made, so what is pinned here is the glue: input mapping, [L, B, C] layout, the gaussian blob,
    the [dir embedding, features] order, activations, and the wrapper's backward."""
    g = golden("field")
    ref = to.FieldRef(bound=float(g["bound"]), num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=21, desired_resolution=8192,
                      gridtype=str(g["gridtype"]), n_hidden_geo=2)
    assert ref.pos_en.embeddings.shape[0] == int(g["n_embeddings"]) and np.array_equal(np.asarray(ref.pos_en.offsets), g["offsets"])
    with torch.no_grad():
        ref.pos_en.embeddings.copy_(_field_table(int(g["n_embeddings"])))
        for nm in ("network", "density_network", "rgb_network"):
            getattr(ref, nm).copy_(T(g[f"{nm}__params"]))
    assert tuple(g["network__cfg"]) == ref.cfg_net and tuple(g["density_network__cfg"]) == ref.cfg_den and tuple(g["rgb_network__cfg"]) == ref.cfg_rgb
    x, d = T(g["x"]), T(g["d"])
    sigma, rad, _ = ref(x, d)
    np.testing.assert_allclose(sigma.detach().numpy(), g["sigma"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(rad.detach().numpy(), g["radiances"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(ref.density(x)["sigma"].detach().numpy(), g["density_sigma"], rtol=1e-5, atol=1e-7)
    loss = (sigma * T(g["w_sigma"]) * 0.01).sum() + (rad * T(g["w_rad"])).sum()
    np.testing.assert_allclose(float(loss.detach()), float(g["loss"]), rtol=1e-5)
    loss.backward()
    for nm in ("network", "density_network", "rgb_network"):
        np.testing.assert_allclose(getattr(ref, nm).grad.numpy(), g[f"{nm}__grad"], rtol=2e-4, atol=1e-7, err_msg=nm)
    ge = ref.pos_en.embeddings.grad
    nz = torch.nonzero(ge.abs().sum(-1)).reshape(-1).numpy()
    assert np.array_equal(nz, g["grad_emb_idx"])
    np.testing.assert_allclose(ge[nz].numpy(), g["grad_emb_val"], rtol=2e-4, atol=1e-8)
