"""GPU: the LGIE editing step (customnerf_amd.sd.editing.EditTrainer, counterpart of utils_init_nerf.py:243-308, 353-394) end to
end on a small field and small SD-shaped networks: the SDS gradient reaches the grid table and the MLPs, the background term
pulls the edited field's bg render towards the cached pretrained one, the per-view cache is filled once."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(**optkw):
    from customnerf_amd import scene as sc, tcnn
    from customnerf_amd.nerf.network_grid import NeRFNetwork
    from customnerf_amd.nerf.provider_utils import generate_rays
    from customnerf_amd.sd import arch
    from customnerf_amd.sd.guidance import StableDiffusion
    from customnerf_amd.sd.editing import EditTrainer
    tcnn.set_default_dtype(torch.float16)
    torch.manual_seed(0)
    opt = sc.make_opt(fp16=True, num_levels=8, num_steps=16, upsample_steps=16, cfg=7.5, log_loss_item=False, **optkw)
    model = NeRFNetwork(opt).cuda()
    with torch.no_grad():
        model.pos_en.embeddings.uniform_(-0.5, 0.5)
    pre = copy.deepcopy(model).eval()
    usd = arch.random_state_dict(arch.unet_params(arch.UNET_TINY), 1)
    vsd = arch.random_state_dict(arch.vae_encoder_params(arch.VAE_TINY), 2)
    guide = StableDiffusion("cuda", "1.5", opt, unet_state=usd, vae_state=vsd, unet_cfg=arch.UNET_TINY, vae_cfg=arch.VAE_TINY)
    H = W = 32
    c2w = torch.from_numpy(sc.poses(4)).cuda()
    o, d = generate_rays(c2w, *sc.intrinsics(H, W), H, W, 1.0, 'nerfstudio')
    o, d = o.view(4, 1, H * W, 3), d.view(4, 1, H * W, 3)
    rgb, mask = sc.targets(4, H, W)
    tr = EditTrainer(model, pre, guide, opt, guide.synthetic_text_embeds(0), guide.synthetic_text_embeds(1))
    data = lambda v: (rgb[v].cuda(), mask[v].cuda(), o[v], d[v], H, W, f"v{v}")
    return tr, model, pre, data


def test_editing_step_updates_field_and_caches_pretrained_render():
    tr, model, pre, data = _setup(keep_bg=1000.0, lambda_sd=0.01)
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    pre_before = {n: p.detach().clone() for n, p in pre.named_parameters()}
    losses = []
    for i in range(6):
        loss, ld = tr.train_step(data(i % 2))
        losses.append(float(loss))
        assert set(ld) == {"loss_sds", "loss_bg"}
    assert len(tr.pt_dict) == 2 and tr.global_step == 6
    assert all(np.isfinite(losses))
    changed = {n: float((p.detach() - before[n]).abs().max()) for n, p in model.named_parameters()}
    assert changed["pos_en.embeddings"] > 0 and changed["network.params"] > 0 and changed["rgb_network.params"] > 0, changed
    for n, p in pre.named_parameters():
        assert torch.equal(p, pre_before[n])                      # the pretrained field is frozen
    for p in model.parameters():
        assert torch.isfinite(p).all()


def test_editing_global_and_local_terms():
    """g_only / l_only select the prompt, the image (full vs fg) and the timestep ratio (utils_init_nerf.py:291-301)."""
    tr, model, pre, data = _setup(keep_bg=0.0, lambda_sd=0.01, g_only=True)
    loss, ld = tr.train_step(data(0))
    assert "loss_bg" not in ld and np.isfinite(float(loss))
    tr2, *_ = _setup(keep_bg=0.0, lambda_sd=0.01, l_only=True, local_t_ratio=0.25)
    ts = []
    orig = tr2.guidance.draw_timestep
    tr2.guidance.draw_timestep = lambda system, t_ratio=1: ts.append(orig(system, t_ratio)) or ts[-1]
    tr2.train_step(data(1))
    assert len(ts) == 1 and ts[0] <= 0.25 * 980 + 1
    # keep_bg only (lambda_sd = 0): pure background-preservation step, identical fields -> zero bg loss when unperturbed nets agree
    tr3, model3, pre3, data3 = _setup(keep_bg=1000.0, lambda_sd=0.0)
    loss3, ld3 = tr3.train_step(data3(0))
    assert set(ld3) == {"loss_bg"} and float(loss3) < 50.0
