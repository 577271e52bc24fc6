"""NeRFNetwork — counterpart of the reference's nerf/network_grid.py:70-206 (grid backbone).

Same module tree and parameter names as the reference so its state dicts line up (SURVEY.md §5 checkpoint row):
`pos_en.embeddings`, `pos_en.offsets`, `network.params`, `density_network.params`, `rgb_network.params`.
Grid geometry defaults to the reference's hard-coded values (tiled, log2 T = 21, finest 8192; network_grid.py:89-96)
and can be overridden through `opt.grid_type / opt.log2_hashmap_size / opt.desired_resolution / opt.num_levels`.
"""
import torch
from torch import nn

from .renderer import NeRFRenderer
from .encoding import get_encoder, get_embedder
from .provider_utils import trunc_exp
from .. import tcnn
from ..field import field, field_attach, field_forward_raw, field_forward_rows, packed_weights
from ..gridencoder import GridEncoder


class RGB_network(nn.Module):
    """network_grid.py:13-68: separate colour / confidence heads (used with --detach_mask_from_field / --mask_no_dir)."""

    def __init__(self, input_ch_views, opt=None):
        super().__init__()
        self.opt = opt
        cfg = {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "Sigmoid", "n_neurons": 64, "n_hidden_layers": 1}
        self.rgb_network = tcnn.Network(input_ch_views + 64, 3, cfg, seed=13)
        if getattr(opt, 'mask_no_dir', False):
            self.conf_network = tcnn.Network(64, 1, cfg, seed=14)
        else:
            ndim = 1 if getattr(opt, 'keyword2', None) is None else 2
            self.conf_network = tcnn.Network(input_ch_views + 64, ndim, cfg, seed=14)

    def forward(self, x):
        if getattr(self.opt, 'mask_no_dir', False):
            dir_dim = x.shape[-1] - 64
            _, fea = x.split([dir_dim, 64], dim=-1)
            rgb = self.rgb_network(x)
            rgb2 = self.conf_network(fea if getattr(self.opt, 'mask_no_dir_nodetach', False) else fea.detach())
        else:
            rgb = self.rgb_network(x)
            rgb2 = self.conf_network(x.detach())
        return torch.cat([rgb, rgb2], dim=-1)


class NeRFNetwork(NeRFRenderer):
    def __init__(self, opt, device=None, **unused):
        super().__init__(opt)
        grid_type = getattr(opt, 'grid_type', 'tiledgrid')
        self.pos_en, self.pos_en_dim = get_encoder(
            grid_type, input_dim=3,
            num_levels=getattr(opt, 'num_levels', 16), level_dim=getattr(opt, 'level_dim', 2),
            base_resolution=getattr(opt, 'base_resolution', 16),
            log2_hashmap_size=getattr(opt, 'log2_hashmap_size', 21),
            desired_resolution=getattr(opt, 'desired_resolution', 8192))
        n_geo = getattr(opt, 'n_hidden_geo', 2)
        self.network = tcnn.Network(self.pos_en_dim, 64, {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None",
                                                          "n_neurons": 64, "n_hidden_layers": n_geo}, seed=10)
        self.density_network = tcnn.Network(64, 1, {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None",
                                                    "n_neurons": 64, "n_hidden_layers": 1}, seed=11)
        self.dir_en, input_ch_views = get_embedder(4)
        sig = {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "Sigmoid", "n_neurons": 64, "n_hidden_layers": 1}
        if getattr(opt, 'train_conf', 0):
            if getattr(opt, 'detach_mask_from_field', False) or getattr(opt, 'mask_no_dir', False):
                self.rgb_network = RGB_network(input_ch_views, opt=opt)
            else:
                self.rgb_network = tcnn.Network(input_ch_views + 64, 3 + 1, sig, seed=12)
        else:
            self.rgb_network = tcnn.Network(input_ch_views + 64, 3, sig, seed=12)
        self.bg_net = None
        self.supports_dir_group = True

    def background(self, d):
        return torch.zeros(d.size(), dtype=d.dtype, device=d.device)

    def gaussian(self, x):
        """network_grid.py:150-156: density blob at the scene centre."""
        d = (x ** 2).sum(-1)
        return 5 * torch.exp(-d / (2 * 0.2 ** 2))

    # ---- fused MFMA path (cnerf_field_forward / _backward): the standard CustomNeRF field, one launch per evaluation
    def _fused_cfg(self):
        cfg = getattr(self, '_fused_cfg_cache', None)
        if cfg is None:
            ok = (getattr(self.opt, 'fused_field', True) and isinstance(self.pos_en, GridEncoder) and self.pos_en.level_dim == 2
                  and self.pos_en.input_dim == 3 and self.pos_en.output_dim <= 64 and isinstance(self.rgb_network, tcnn.Network)
                  and self.network.n_hidden_layers in (1, 2) and self.density_network.n_hidden_layers == 1
                  and self.rgb_network.n_hidden_layers == 1 and self.rgb_network.n_output_dims in (3, 4))
            cfg = (self.pos_en.output_dim, self.network.n_hidden_layers, self.rgb_network.n_output_dims) if ok else False
            self._fused_cfg_cache = cfg
        return cfg

    def _half(self):
        """Precision of the fused field (grid gather + the three MLPs as ONE unit): the tcnn compute dtype, NOT the autocast state.
        Deliberate deviation from the reference: there the grid encoder casts its table to half only under autocast (grid.py:45) while
        tinycudann always computes in half, so outside autocast (evaluation loops that forget the context manager) float32 grid features
        are rounded to half at the MLP input.  Here the gather feeds the fused kernel in the MLP's own precision — half features from the
        fp16 shadow table when the networks are half, exact float32 end to end when `tcnn.set_default_dtype(torch.float32)` — and the
        result does not depend on whether a caller wrapped the call in autocast.  (`GridEncoder.forward` / `grid_encode`, the drop-in
        surface, keep the reference's autocast rule.)"""
        return self.network.compute_dtype == torch.float16

    def _weight_image(self):
        """The fp16 fragment image of the three MLPs for the fused kernels (field.packed_weights), or None = the kernels stage from the float32
        parameters.  Off unless `packed_field_weights` is set: the image is keyed on the parameters' version counters, which a write through
        `.data` does not move (torch_ema's copy_to / restore, a drop-in trainer's `param.data.copy_`) — the trainers of this package, whose
        every parameter write goes through an optimiser, switch it on for the duration of a training step (trainer.packed_weights_window)."""
        if not self.__dict__.get('packed_field_weights') or not self._half():
            return None
        enc_dim, n_geo, n_rgb = self._fused_cfg()
        return packed_weights(self.__dict__.setdefault('_wimg_cache', {}), enc_dim, n_geo, n_rgb, self.network.params, self.density_network.params,
                              self.rgb_network.params)

    def _geo(self, x):
        x_en = self.pos_en(x, bound=self.opt.bound)
        fea = self.network(x_en)
        sigma = self.density_network(fea)
        sigma = trunc_exp(sigma.squeeze(-1) + self.gaussian(x))
        return fea, sigma

    def forward(self, x, d, l=None, ratio=1, shading='albedo', dir_group=1):
        """network_grid.py:159-177 -> (sigma [P], radiances [P, 3(+1)], None).
        dir_group > 1: `d` holds one direction per dir_group consecutive samples (the samples of a ray share rays_d)."""
        cfg = self._fused_cfg()
        if cfg and x.is_cuda:
            enc_dim, n_geo, n_rgb = cfg
            x = x.reshape(-1, 3)
            enc = self.pos_en.encode(x, bound=self.opt.bound, half=self._half())
            sigma, rgbc = field(enc, x, d.reshape(-1, 3), dir_group, enc_dim, n_geo, n_rgb, self.network.params, self.density_network.params,
                                self.rgb_network.params, wimg=self._weight_image())
            return sigma, (rgbc if n_rgb == 4 else rgbc[:, :3]), None
        if dir_group != 1:
            d = d.reshape(-1, 3).repeat_interleave(dir_group, dim=0)[:x.reshape(-1, 3).shape[0]]
        fea, sigma = self._geo(x)
        view_en = self.dir_en(d)
        rgb_input = torch.cat([view_en.to(fea.dtype), fea], dim=-1)
        radiances = self.rgb_network(rgb_input)
        return sigma, radiances, None

    def density(self, x):
        """network_grid.py:180-193"""
        cfg = self._fused_cfg()
        if cfg and x.is_cuda and not torch.is_grad_enabled():
            enc_dim, n_geo, n_rgb = cfg
            x = x.reshape(-1, 3).contiguous().float()
            enc = self.pos_en.encode(x, bound=self.opt.bound, half=self._half())
            sigma, _ = field_forward_raw(enc, x, None, 1, enc_dim, n_geo, n_rgb, self.network.params, self.density_network.params, None,
                                         with_rgb=False, wimg=self._weight_image())
            return {'sigma': sigma}
        return {'sigma': self._geo(x)[1]}

    # ---- split evaluation (renderer._run_fused): the grid features of a sample list are gathered block by block into ONE kernel-layout
    # buffer, so the coarse samples of run() — whose features the density pass needs anyway — are not gathered a second time for the
    # full evaluation (the reference encodes them twice: renderer.py:327 and :375).  Same arithmetic per sample, one gather less.
    @property
    def supports_split_eval(self):
        return bool(self._fused_cfg())

    def split_buffers(self, P, device):
        """-> (enc [L, P, 2] feature buffer, unit [P, 3] grid coordinates) to be filled by split_encode"""
        dt = torch.float16 if self._half() else torch.float32
        L = self.pos_en.num_levels
        return torch.empty(L, P, 2, dtype=dt, device=device), torch.empty(P, 3, dtype=torch.float32, device=device)

    @torch.no_grad()
    def split_encode(self, enc, unit, x, row0, unit_ready=False, importance=False):
        """gather the features of x [B, 3] into rows row0.. of enc (and their [0,1] grid coordinates into unit; unit_ready: the sampling
        kernel already wrote them).  importance: x are importance samples (renderer.py:340-352) — whether they are spread or hug a surface
        depends on the field's state, so the gather's traversal is measured in place (gridencoder.grid.TraversalTuner)"""
        B = x.shape[0]
        u = unit[row0:row0 + B]
        if not unit_ready:
            torch.add(x, self.opt.bound, out=u)                                     # grid.py:156: (x + bound) / (2 bound)
            u.div_(2 * self.opt.bound)
        trav = 0
        if importance and getattr(self.opt, 'tune_gather_traversal', True):
            trav = self.__dict__.get('_fine_tuner')
            if trav is None:
                from ..gridencoder.grid import TraversalTuner
                trav = self.__dict__['_fine_tuner'] = TraversalTuner()
        self.pos_en.encode_into(u, enc, row0, half=self._half(), traversal=trav)

    @torch.no_grad()
    def split_density(self, enc, x):
        """sigma of the first len(x) rows of enc (network_grid.py:180-193)"""
        enc_dim, n_geo, n_rgb = self._fused_cfg()
        sigma, _ = field_forward_raw(enc, x, None, 1, enc_dim, n_geo, n_rgb, self.network.params, self.density_network.params, None, with_rgb=False,
                                     wimg=self._weight_image())
        return sigma

    def _overlap_plan(self):
        return getattr(self.opt, 'overlap_scatter_plan', True)

    def split_prepare(self, unit, grad_enabled):
        """As soon as every row of `unit` is written (i.e. before the last split_encode): start the coordinate-only half of the backward
        scatter on the side stream, where it overlaps that gather instead of the field kernels.  -> plan for split_forward, or None."""
        if not (grad_enabled and self.pos_en.embeddings.requires_grad and self._overlap_plan()):
            return None
        if not getattr(self.opt, 'early_scatter_plan', True):
            return None
        return self.pos_en.prepare_backward(unit, self._half())

    def split_prepare_rows(self, state, unit, grad_enabled, row0, rows, finish):
        """split_prepare in pieces: the histogram of rows [row0, row0 + rows) as soon as those coordinates exist (the coarse block right after the
        coarse sampling, the fine block after the importance sampling), so that the whole plan is ready before the field backward starts.
        -> (state | plan | None, piecewise): piecewise False = not available for this shape / precision: use split_prepare."""
        if not (grad_enabled and self.pos_en.embeddings.requires_grad and self._overlap_plan() and getattr(self.opt, 'early_scatter_plan', True)):
            return None, True
        blk = self.pos_en.hist_block_points(self._half())
        if not blk or row0 % blk or (rows % blk and row0 + rows != unit.shape[0]) or not getattr(self.opt, 'piecewise_scatter_plan', True):
            return None, False
        return self.pos_en.prepare_backward_rows(state, unit, self._half(), row0, rows, finish), True

    def split_forward(self, enc, unit, x, d, dir_group, plan=None):
        """forward() on a complete feature buffer: (sigma [P], rgbc [P, 4]); gradients reach the table through attach_backward"""
        enc_dim, n_geo, n_rgb = self._fused_cfg()
        if torch.is_grad_enabled() and self.pos_en.embeddings.requires_grad:
            enc = self.pos_en.attach_backward(enc, unit, overlap=self._overlap_plan(), plan=plan)
        sigma, rgbc = field(enc, x, d.reshape(-1, 3), dir_group, enc_dim, n_geo, n_rgb, self.network.params, self.density_network.params,
                            self.rgb_network.params, grad_in_place=bool(getattr(self, 'grad_in_place', False)), wimg=self._weight_image())
        return sigma, rgbc

    # ---- block-wise full evaluation (renderer._run_fused): the coarse block is evaluated in full as soon as its features exist (its sigma is
    # what the importance sampling needs, so the reference's separate density pass — renderer.py:326 — disappears), the fine block after its
    # gather; split_attach then makes the pair differentiable.
    def split_outputs(self, P, device):
        return torch.empty(P, dtype=torch.float32, device=device), torch.empty(P, 4, dtype=torch.float32, device=device)

    @torch.no_grad()
    def split_forward_rows(self, enc, row0, x_rows, d, dir_group, sigma_all, rgbc_all):
        enc_dim, n_geo, n_rgb = self._fused_cfg()
        n = x_rows.shape[0]
        field_forward_rows(enc, row0, x_rows, d.reshape(-1, 3).contiguous().float(), dir_group, enc_dim, n_geo, n_rgb, self.network.params,
                           self.density_network.params, self.rgb_network.params, sigma_all[row0:row0 + n], rgbc_all[row0:row0 + n],
                           wimg=self._weight_image())

    def split_attach(self, enc, unit, x, d, dir_group, sigma_all, rgbc_all, plan=None):
        """(sigma [P], rgbc [P, 4]) computed by split_forward_rows -> the same values, differentiable in the table and the MLP parameters"""
        if not torch.is_grad_enabled():
            return sigma_all, rgbc_all
        enc_dim, n_geo, n_rgb = self._fused_cfg()
        if self.pos_en.embeddings.requires_grad:
            enc = self.pos_en.attach_backward(enc, unit, overlap=self._overlap_plan(), plan=plan)
        return field_attach(enc, x, d.reshape(-1, 3), dir_group, enc_dim, n_geo, n_rgb, self.network.params, self.density_network.params,
                            self.rgb_network.params, sigma_all, rgbc_all, grad_in_place=bool(getattr(self, 'grad_in_place', False)),
                            wimg=self._weight_image())

    def get_params(self, lr):
        """network_grid.py:196-206: grid lr x10."""
        return [
            {'params': self.pos_en.parameters(), 'lr': lr * 10},
            {'params': self.network.parameters(), 'lr': lr},
            {'params': self.density_network.parameters(), 'lr': lr},
            {'params': self.rgb_network.parameters(), 'lr': lr},
        ]
