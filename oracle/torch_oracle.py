"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product path).

torch-CPU restatement of the reference's *Python* hot path: the pure-PyTorch renderer
(`nerf/renderer.py` run / weights_sum_i / sample_pdf / run_cuda / update_extra_state), the field
(`nerf/network_grid.py`), the small helpers (`nerf/provider_utils.py`, `nerf/base.py`) and the ray
generators (`nerf/provider.py`, `nerf/provider_utils.py`).  Every function cites the reference lines it
follows.  It is pinned against golden vectors produced by importing the reference itself
(tests/golden/make_golden.py -> tests/golden/*.npz, checked by tests/test_oracle_golden.py): sample_pdf, weights_sum_i, run() in five
configurations, the ray generators incl. fisheye, trunc_exp, the embedder, the offset tables, update_extra_state (occupancy.npz), and the
field's glue — NeRFNetwork.forward / density over the real GridEncoder wrapper (field.npz; the kernel and the MLP underneath being the
restatements of this directory).

The grid encoder / ray-marching arithmetic comes from the C restatement (oracle/c_oracle.py).

tinycudann (the reference's MLP engine) is third-party, un-vendored and unpinned (README.md:49-50,
requirements.txt:19-20): PARITY UNPINNED for the MLP arithmetic.  The MLP semantic defined here is the
published FullyFusedMLP contract: bias-free dense layers, widths padded to multiples of 16, ReLU hidden
activation, optional sigmoid on the output, fp16 weights/activations with fp32 accumulation in `half`
mode (fp32 everywhere otherwise), one flat parameter vector holding the row-major [out, in] matrices.
"""
import math

import numpy as np
import torch

from . import c_oracle as co


# ------------------------------------------------------------------ grid encoder (grid.py)
def grid_offsets(input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16,
                 log2_hashmap_size=19, desired_resolution=None, align_corners=False):
    """grid.py:106-137: per-level offset table and the effective per_level_scale."""
    if desired_resolution is not None:
        per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
    offsets, offset = [], 0
    max_params = 2 ** log2_hashmap_size
    for i in range(num_levels):
        resolution = int(np.ceil(base_resolution * per_level_scale ** i))
        params_in_level = min(max_params, (resolution if align_corners else resolution + 1) ** input_dim)
        params_in_level = int(np.ceil(params_in_level / 8) * 8)
        offsets.append(offset)
        offset += params_in_level
    offsets.append(offset)
    return np.array(offsets, dtype=np.int32), per_level_scale


class _GridEncodeRef(torch.autograd.Function):
    """grid.py:24-95 (calc_grad_inputs omitted on purpose: positions never require grad on this path)."""

    @staticmethod
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, gridtype, align_corners, interp, half):
        out, _ = co.grid_encode_forward(inputs.numpy(), embeddings.detach().numpy(), offsets, per_level_scale,
                                        base_resolution, False, gridtype, align_corners, interp, None, half)
        ctx.save_for_backward(inputs)
        ctx.cfg = (tuple(embeddings.shape), offsets, per_level_scale, base_resolution, gridtype, align_corners, interp)
        return torch.from_numpy(out)

    @staticmethod
    def backward(ctx, grad):
        (inputs,) = ctx.saved_tensors
        shape, offsets, pls, H, gridtype, ac, interp = ctx.cfg
        ge, _ = co.grid_encode_backward(grad.contiguous().numpy(), inputs.numpy(), shape, offsets, pls, H, None, gridtype, ac, interp)
        return None, torch.from_numpy(ge), None, None, None, None, None, None, None


class GridEncoderRef(torch.nn.Module):
    """grid.py:102-168."""

    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16, log2_hashmap_size=19,
                 desired_resolution=None, gridtype='hash', align_corners=False, interpolation='linear', half=False):
        super().__init__()
        self.offsets, self.per_level_scale = grid_offsets(input_dim, num_levels, level_dim, per_level_scale, base_resolution,
                                                          log2_hashmap_size, desired_resolution, align_corners)
        self.input_dim, self.num_levels, self.level_dim = input_dim, num_levels, level_dim
        self.base_resolution = base_resolution
        self.output_dim = num_levels * level_dim
        self.gridtype_id = {'hash': 0, 'tiled': 1}[gridtype]
        self.interp_id = {'linear': 0, 'smoothstep': 1}[interpolation]
        self.align_corners = align_corners
        self.half = half
        self.embeddings = torch.nn.Parameter(torch.empty(int(self.offsets[-1]), level_dim).uniform_(-1e-4, 1e-4))

    def forward(self, inputs, bound=1):
        inputs = (inputs + bound) / (2 * bound)                      # grid.py:156
        prefix = list(inputs.shape[:-1])
        inputs = inputs.reshape(-1, self.input_dim).contiguous()
        out = _GridEncodeRef.apply(inputs.detach(), self.embeddings, self.offsets, self.per_level_scale, self.base_resolution,
                                   self.gridtype_id, self.align_corners, self.interp_id, self.half)
        return out.view(prefix + [self.output_dim])


# ------------------------------------------------------------------ MLP (tinycudann contract, see module docstring)
def pad16(n):
    return (n + 15) // 16 * 16


def mlp_layer_dims(n_in, n_out, n_neurons, n_hidden_layers):
    """[(out, in)] per layer with tcnn's padding: input -> multiple of 16, output -> multiple of 16."""
    dims, cur = [], pad16(n_in)
    for _ in range(n_hidden_layers):
        dims.append((n_neurons, cur))
        cur = n_neurons
    dims.append((pad16(n_out), cur))
    return dims


def mlp_n_params(n_in, n_out, n_neurons, n_hidden_layers):
    return sum(o * i for o, i in mlp_layer_dims(n_in, n_out, n_neurons, n_hidden_layers))


def _q(x, half):
    return x.half().float() if half else x


def mlp_forward(x, params, n_in, n_out, n_neurons, n_hidden_layers, output_activation='None', half=False):
    """x [P, n_in] float32 -> [P, n_out] float32 (values exactly representable in fp16 when half=True)."""
    dims = mlp_layer_dims(n_in, n_out, n_neurons, n_hidden_layers)
    if x.shape[-1] < dims[0][1]:
        x = torch.nn.functional.pad(x, (0, dims[0][1] - x.shape[-1]))
    h = _q(x, half)
    off = 0
    for li, (o, i) in enumerate(dims):
        W = _q(params[off:off + o * i].view(o, i), half)
        off += o * i
        h = h @ W.t()
        if li < len(dims) - 1:
            h = torch.relu(h)
        elif output_activation == 'Sigmoid':
            h = torch.sigmoid(h)
        h = _q(h, half)
    return h[:, :n_out]


def xavier_params(n_in, n_out, n_neurons, n_hidden_layers, generator=None):
    """U(+-sqrt(6/(in+out))) per matrix (SURVEY.md §8d synthetic-scene recipe)."""
    chunks = []
    for o, i in mlp_layer_dims(n_in, n_out, n_neurons, n_hidden_layers):
        s = math.sqrt(6.0 / (i + o))
        chunks.append((torch.rand(o * i, generator=generator) * 2 - 1) * s)
    return torch.cat(chunks)


# ------------------------------------------------------------------ small helpers
class _TruncExp(torch.autograd.Function):
    """provider_utils.py:16-29"""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-15, 15))


trunc_exp = _TruncExp.apply


def safe_normalize(x, eps=1e-20):
    """provider_utils.py:125-126"""
    return x / torch.sqrt(torch.clamp(torch.sum(x * x, -1, keepdim=True), min=eps))


def freq_embed(d, multires=4):
    """base.py:10-77 get_embedder(4): [x, sin(2^k x), cos(2^k x)]_k=0..3 -> 27-d"""
    out = [d]
    for f in (2.0 ** torch.linspace(0.0, multires - 1, multires)).tolist():
        out.append(torch.sin(d * f))
        out.append(torch.cos(d * f))
    return torch.cat(out, dim=-1)


def gaussian(x):
    """network_grid.py:150-156"""
    return 5 * torch.exp(-(x ** 2).sum(-1) / (2 * 0.2 ** 2))


# ------------------------------------------------------------------ field (network_grid.py:70-206)
class FieldRef(torch.nn.Module):
    """NeRFNetwork restated (train_conf branch: rgb_network 91 -> 64 -> 3+1 sigmoid, network_grid.py:116-129)."""

    def __init__(self, bound=2.0, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048,
                 gridtype='hash', hidden=64, n_hidden_geo=2, half=False, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.bound, self.half, self.hidden = bound, half, hidden
        self.pos_en = GridEncoderRef(3, num_levels, level_dim, 2, base_resolution, log2_hashmap_size, desired_resolution,
                                     gridtype, half=half)
        with torch.no_grad():
            self.pos_en.embeddings.copy_((torch.rand(self.pos_en.embeddings.shape, generator=g) * 2 - 1) * 1e-4)
        self.enc_dim = num_levels * level_dim
        self.cfg_net = (self.enc_dim, hidden, hidden, n_hidden_geo)      # network_grid.py:98-104
        self.cfg_den = (hidden, 1, hidden, 1)                            # :106-112
        self.cfg_rgb = (27 + hidden, 4, hidden, 1)                       # :121-129
        self.network = torch.nn.Parameter(xavier_params(*self.cfg_net, generator=g))
        self.density_network = torch.nn.Parameter(xavier_params(*self.cfg_den, generator=g))
        self.rgb_network = torch.nn.Parameter(xavier_params(*self.cfg_rgb, generator=g))

    def _geo(self, x):
        x_en = self.pos_en(x, bound=self.bound)
        fea = mlp_forward(x_en, self.network, *self.cfg_net, 'None', self.half)
        raw = mlp_forward(fea, self.density_network, *self.cfg_den, 'None', self.half)
        sigma = trunc_exp(raw.squeeze(-1) + gaussian(x))                # network_grid.py:166 / :189
        return fea, sigma

    def density(self, x):
        return {'sigma': self._geo(x)[1]}

    def forward(self, x, d):
        fea, sigma = self._geo(x)
        rgb_in = torch.cat([freq_embed(d), fea], dim=-1)                # network_grid.py:168-172
        rgbc = mlp_forward(rgb_in, self.rgb_network, *self.cfg_rgb, 'Sigmoid', self.half)
        return sigma, rgbc, None


# ------------------------------------------------------------------ renderer (renderer.py)
def sample_pdf(bins, weights, n_samples, det=False, u=None):
    """renderer.py:21-55.  `u` (optional) replaces the torch.rand draw at :37."""
    weights = weights + 1e-5
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    if det:
        u = torch.linspace(0. + 0.5 / n_samples, 1. - 0.5 / n_samples, steps=n_samples)
        u = u.expand(list(cdf.shape[:-1]) + [n_samples])
    elif u is None:
        u = torch.rand(list(cdf.shape[:-1]) + [n_samples])
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    cdf_b, cdf_a = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    bins_b, bins_a = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    denom = cdf_a - cdf_b
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_b) / denom
    return bins_b + t * (bins_a - bins_b)


def weights_sum_i(sample_dist, sigmas, z_vals, nears, fars, rgbs, prefix, masks, train_conf=True, detach_bg=False,
                  detach_mask_from_field=False, is_all=False, if_fg=False, bg_color=None):
    """renderer.py:407-474 (normals are always None on this path)."""
    if is_all and detach_bg:                                            # :409-418
        edit_points = masks.mean(-1, keepdims=True) >= 0.5
        sigmas = torch.where(edit_points, sigmas, sigmas.detach())
        rgbs = torch.where(edit_points, rgbs, rgbs.detach())
    deltas = z_vals[..., 1:] - z_vals[..., :-1]
    deltas = torch.cat([deltas, sample_dist * torch.ones_like(deltas[..., :1])], dim=-1)
    alphas = 1 - torch.exp(-deltas * sigmas.squeeze(-1))
    alphas_shifted = torch.cat([torch.ones_like(alphas[..., :1]), 1 - alphas + 1e-15], dim=-1)
    weights = alphas * torch.cumprod(alphas_shifted, dim=-1)[..., :-1]
    results = {}
    weights_sum = weights.sum(dim=-1)
    ori_z_vals = ((z_vals - nears) / (fars - nears)).clamp(0, 1)
    depth = torch.sum(weights * ori_z_vals, dim=-1)
    image = torch.sum(weights.unsqueeze(-1) * rgbs, dim=-2)
    image = image.view(*prefix, 3)
    depth = depth.view(*prefix)
    if if_fg and bg_color is not None:
        results['black_image'] = image
        image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
    results['image'] = image
    if train_conf:
        w = weights.unsqueeze(-1).detach() if detach_mask_from_field else weights.unsqueeze(-1)
        results['render_mask'] = torch.sum(w * masks, dim=-2).view(*prefix, -1)
    results['depth'] = depth
    results['weights_sum'] = weights_sum
    results['weights'] = weights
    results['mask'] = (nears < fars).reshape(*prefix)
    return results


def run(field, rays_o, rays_d, aabb, min_near, num_steps=64, upsample_steps=64, perturb=False, training=True,
        train_conf=0.01, soft_mask=True, conf_thr=0.5, detach_bg=False, detach_mask_from_field=False,
        draws=None, skip_fine_density=False):
    """renderer.py:278-405.  `draws` = dict(light=randn(3), z=rand(N,T), u=rand(N,t)) replaces the RNG draws at
    :305, :317 and sample_pdf:37 (same order); None draws them here in that order.
    skip_fine_density=True drops the output-dead density pass at :353 (SURVEY.md §2 'Known reference defects')."""
    prefix = rays_o.shape[:-1]
    rays_o = rays_o.contiguous().view(-1, 3)
    rays_d = rays_d.contiguous().view(-1, 3)
    N = rays_o.shape[0]
    nears, fars = co.near_far_from_aabb(rays_o.numpy(), rays_d.numpy(), aabb.numpy(), min_near)
    nears, fars = torch.from_numpy(nears).unsqueeze(-1), torch.from_numpy(fars).unsqueeze(-1)
    draws = dict(draws or {})
    if 'light' not in draws:
        draws['light'] = torch.randn(3)                                   # :305 (value unused downstream)
    z_vals = torch.linspace(0.0, 1.0, num_steps).unsqueeze(0).expand((N, num_steps))
    z_vals = nears + (fars - nears) * z_vals
    sample_dist = (fars - nears) / num_steps
    if perturb:
        zr = draws['z'] if 'z' in draws else torch.rand(z_vals.shape)
        z_vals = z_vals + (zr - 0.5) * sample_dist
    xyzs = rays_o.unsqueeze(-2) + rays_d.unsqueeze(-2) * z_vals.unsqueeze(-1)
    xyzs = torch.min(torch.max(xyzs, aabb[:3]), aabb[3:])
    sig_c = field.density(xyzs.reshape(-1, 3))['sigma'].view(N, num_steps)
    if upsample_steps > 0:
        with torch.no_grad():
            deltas = z_vals[..., 1:] - z_vals[..., :-1]
            deltas = torch.cat([deltas, sample_dist * torch.ones_like(deltas[..., :1])], dim=-1)
            alphas = 1 - torch.exp(-deltas * sig_c)
            alphas_shifted = torch.cat([torch.ones_like(alphas[..., :1]), 1 - alphas + 1e-15], dim=-1)
            weights = alphas * torch.cumprod(alphas_shifted, dim=-1)[..., :-1]
            z_vals_mid = (z_vals[..., :-1] + 0.5 * deltas[..., :-1])
            new_z_vals = sample_pdf(z_vals_mid, weights[:, 1:-1], upsample_steps, det=not training, u=draws.get('u')).detach()
            new_xyzs = rays_o.unsqueeze(-2) + rays_d.unsqueeze(-2) * new_z_vals.unsqueeze(-1)
            new_xyzs = torch.min(torch.max(new_xyzs, aabb[:3]), aabb[3:])
        if not skip_fine_density:
            field.density(new_xyzs.reshape(-1, 3))                        # :353, result never used
        z_vals = torch.cat([z_vals, new_z_vals], dim=1)
        z_vals, z_index = torch.sort(z_vals, dim=1)
        xyzs = torch.cat([xyzs, new_xyzs], dim=1)
        xyzs = torch.gather(xyzs, dim=1, index=z_index.unsqueeze(-1).expand_as(xyzs))
    dirs = rays_d.view(-1, 1, 3).expand_as(xyzs)
    sigmas, rgbs, _ = field(xyzs.reshape(-1, 3), dirs.reshape(-1, 3))
    rgbs, masks = rgbs.split([3, rgbs.shape[-1] - 3], dim=-1)
    masks = masks.reshape(N, -1, masks.shape[-1])
    sigmas = sigmas.view(N, -1, 1)
    rgbs = rgbs.reshape(N, -1, 3)
    if not train_conf:
        return {}
    kw = dict(train_conf=True, detach_bg=detach_bg, detach_mask_from_field=detach_mask_from_field)
    results = weights_sum_i(sample_dist, sigmas, z_vals, nears, fars, rgbs, prefix, masks, is_all=True, **kw)
    if soft_mask:                                                          # :386-389
        edit_mask = torch.sigmoid((masks - conf_thr) * 100)
        sigmas_fg = sigmas * edit_mask
        sigmas_bg = sigmas * (1 - edit_mask)
    else:                                                                  # :391-395
        edit_mask = masks > 0.5
        sigmas_bg = torch.where(edit_mask, torch.zeros_like(sigmas), sigmas)
        sigmas_fg = torch.where(edit_mask, sigmas, torch.zeros_like(sigmas))
    results['sigma'], results['rgbs'], results['edit_mask'] = sigmas, rgbs, edit_mask
    results['z_vals'] = z_vals
    results['fg'] = weights_sum_i(sample_dist, sigmas_fg, z_vals, nears, fars, rgbs, prefix, masks, if_fg=True, **kw)
    results['bg'] = weights_sum_i(sample_dist, sigmas_bg, z_vals, nears, fars, rgbs, prefix, masks, **kw)
    return results


# ------------------------------------------------------------------ ray generation
def get_rays(poses, intrinsics, H, W):
    """provider_utils.py:239-302 (N=-1 branch: all pixels)."""
    B = poses.shape[0]
    fx, fy, cx, cy = intrinsics
    i, j = torch.meshgrid(torch.linspace(0, W - 1, W), torch.linspace(0, H - 1, H), indexing='ij')
    i = i.t().reshape([1, H * W]).expand([B, H * W]) + 0.5
    j = j.t().reshape([1, H * W]).expand([B, H * W]) + 0.5
    zs = torch.ones_like(i)
    xs = (i - cx) / fx * zs
    ys = (j - cy) / fy * zs
    directions = safe_normalize(torch.stack((xs, ys, zs), dim=-1))
    rays_d = directions @ poses[:, :3, :3].transpose(-1, -2)
    rays_o = poses[..., :3, 3][..., None, :].expand_as(rays_d)
    return rays_o, rays_d


def generate_rays(c2w, fx, fy, cx, cy, H, W, level=1):
    """provider.py:402-464, pinhole branch: c2w [V,3,4] -> origins, directions [V, H, W, 3]."""
    tx = torch.linspace(0, W * level - 1, W)
    ty = torch.linspace(0, H * level - 1, H)
    x, y = torch.meshgrid(tx, ty, indexing='ij')
    x = (x + 0.5).reshape(-1)
    y = (y + 0.5).reshape(-1)
    d = torch.stack([(x - cx) / fx, -(y - cy) / fy, -torch.ones_like(x)], -1)        # [W*H, 3]
    origins, directions = [], []
    for v in range(c2w.shape[0]):
        R = c2w[v, :3, :3]
        dv = torch.sum(d[:, None, :] * R[None], dim=-1)
        dv = torch.nn.functional.normalize(dv, dim=-1)
        ov = c2w[v, :3, 3].expand_as(dv)
        origins.append(ov.reshape(W, H, 3).permute(1, 0, 2))
        directions.append(dv.reshape(W, H, 3).permute(1, 0, 2))
    return torch.stack(origins), torch.stack(directions)


def undistort(coords, dist, eps=1e-3, max_iterations=10):
    """provider_utils.py:197-234 (radial_and_tangential_undistort) with its residual / Jacobian (:128-194): coords [..., 2],
    dist = [k1, k2, k3, k4, p1, p2] -> undistorted coords.  Pinned by tests/golden/rays_fisheye.npz, which the reference's own function produced."""
    k1, k2, k3, k4, p1, p2 = [torch.tensor(float(v)) for v in dist]
    xd, yd = coords[..., 0], coords[..., 1]
    x, y = xd, yd
    for _ in range(max_iterations):
        r = x * x + y * y
        d = 1.0 + r * (k1 + r * (k2 + r * (k3 + r * k4)))
        fx = d * x + 2 * p1 * x * y + p2 * (r + 2 * x * x) - xd
        fy = d * y + 2 * p2 * x * y + p1 * (r + 2 * y * y) - yd
        d_r = k1 + r * (2.0 * k2 + r * (3.0 * k3 + r * 4.0 * k4))
        d_x = 2.0 * x * d_r
        d_y = 2.0 * y * d_r
        fx_x = d + d_x * x + 2.0 * p1 * y + 6.0 * p2 * x
        fx_y = d_y * x + 2.0 * p1 * x + 2.0 * p2 * y
        fy_x = d_x * y + 2.0 * p2 * y + 2.0 * p1 * x
        fy_y = d + d_y * y + 2.0 * p2 * x + 6.0 * p1 * y
        den = fy_x * fx_y - fx_x * fy_y
        xn = fx * fy_y - fy * fx_y
        yn = fy * fx_x - fx * fy_x
        x = x + torch.where(torch.abs(den) > eps, xn / den, torch.zeros_like(den))
        y = y + torch.where(torch.abs(den) > eps, yn / den, torch.zeros_like(den))
    return torch.stack([x, y], dim=-1)


def generate_rays_fisheye(c2w, fx, fy, cx, cy, H, W, level, dist):
    """provider.py:402-464, OPENCV_FISHEYE branch (:421-433): c2w [V,3,4] -> origins, directions [V, H, W, 3]."""
    tx = torch.linspace(0, W * level - 1, W)
    ty = torch.linspace(0, H * level - 1, H)
    x, y = torch.meshgrid(tx, ty, indexing='ij')
    x = (x + 0.5).reshape(-1)
    y = (y + 0.5).reshape(-1)
    coord = undistort(torch.stack([(x - cx) / fx, -(y - cy) / fy], -1), dist)
    theta = torch.clip(torch.sqrt(torch.sum(coord ** 2, dim=-1)), 0.0, math.pi)
    d = torch.stack([coord[..., 0] * torch.sin(theta) / theta, coord[..., 1] * torch.sin(theta) / theta, -torch.cos(theta)], -1)
    origins, directions = [], []
    for v in range(c2w.shape[0]):
        R = c2w[v, :3, :3]
        dv = torch.nn.functional.normalize(torch.sum(d[:, None, :] * R[None], dim=-1), dim=-1)
        origins.append(c2w[v, :3, 3].expand_as(dv).reshape(W, H, 3).permute(1, 0, 2))
        directions.append(dv.reshape(W, H, 3).permute(1, 0, 2))
    return torch.stack(origins), torch.stack(directions)


def update_extra_state(field, density_grid, bound, cascade, grid_size=128, decay=0.95, density_thresh=10.0, rand=None):
    """renderer.py:1658-1715 with S = grid_size (one chunk): density_grid [cascade, H^3] (Morton order, numpy float32, updated copy returned),
    `rand` = list of the [H^3, 3] `torch.rand_like` draws of :1695 per cascade.  -> (density_grid, mean_density, bitfield).
    `2 * coords / (H - 1)` is written as a multiplication by the float reciprocal: that is what torch computes on the GPU the reference
    runs on (a tensor divided by a host scalar), and the HIP kernel does the same."""
    H = grid_size
    ar = torch.arange(H, dtype=torch.int32)
    xx, yy, zz = torch.meshgrid(ar, ar, ar, indexing='ij')
    coords = torch.cat([xx.reshape(-1, 1), yy.reshape(-1, 1), zz.reshape(-1, 1)], dim=-1)
    indices = torch.from_numpy(co.morton3D(coords.numpy())).long()
    xyzs = (2 * coords.float()) * torch.tensor(1.0 / (H - 1), dtype=torch.float32) - 1
    grid = torch.from_numpy(np.array(density_grid, dtype=np.float32, copy=True))
    tmp = -torch.ones_like(grid)
    for cas in range(cascade):
        b = min(2 ** cas, bound)
        half = b / H
        cas_xyzs = xyzs * (b - half)
        r = rand[cas] if rand is not None else torch.rand_like(cas_xyzs)
        cas_xyzs = cas_xyzs + (r * 2 - 1) * half
        with torch.no_grad():
            sig = field.density(cas_xyzs)['sigma'].reshape(-1).float()
        tmp[cas, indices] = sig
    valid = grid >= 0
    grid[valid] = torch.maximum(grid[valid] * decay, tmp[valid])
    mean_density = torch.mean(grid[valid]).item()
    thresh = min(mean_density, density_thresh)
    return grid.numpy(), mean_density, co.packbits(grid.numpy(), thresh)


# ------------------------------------------------------------------ run_cuda (renderer.py:597-718) and the occupancy grid
def run_cuda_train(field, rays_o, rays_d, aabb, bound, density_bitfield, cascade, grid_size=128, noises=None, dt_gamma=0,
                   max_steps=1024, T_thresh=1e-4, mean_count=-1, force_all_rays=True):
    """Training branch (:617-635).  near_far uses the wrapper default min_near=0.2 as the reference does (:612-613).
    The field's 4-channel output is sliced to rgb before compositing (what run_cuda2 does at :510; run_cuda itself
    passes 4-channel rows to a stride-3 kernel, a reference defect — SURVEY.md §2)."""
    rays_o = rays_o.contiguous().view(-1, 3)
    rays_d = rays_d.contiguous().view(-1, 3)
    nears, fars = co.near_far_from_aabb(rays_o.numpy(), rays_d.numpy(), aabb.numpy(), 0.2)
    counter = np.zeros(2, np.int32)
    xyzs, dirs, deltas, rays = co.march_rays_train(rays_o.numpy(), rays_d.numpy(), bound, density_bitfield, cascade, grid_size, nears,
                                                   fars, counter, mean_count, noises, 128, force_all_rays, dt_gamma, max_steps)
    sigmas, rgbs, _ = field(torch.from_numpy(xyzs), torch.from_numpy(dirs))
    ws, depth, image = CompositeTrainRef.apply(sigmas, rgbs[..., :3].contiguous(), torch.from_numpy(deltas), torch.from_numpy(rays), T_thresh)
    return dict(image=image, depth=depth, weights_sum=ws, mask=torch.from_numpy(nears < fars), rays=rays, xyzs=xyzs, dirs=dirs,
                deltas=deltas, counter=counter, sigmas=sigmas, rgbs=rgbs)


class CompositeTrainRef(torch.autograd.Function):
    """raymarching.py:239-289 on the C oracle."""

    @staticmethod
    def forward(ctx, sigmas, rgbs, deltas, rays, T_thresh):
        ws, depth, image = co.composite_rays_train_forward(sigmas.detach().numpy(), rgbs.detach().numpy(), deltas.numpy(), rays.numpy(), T_thresh)
        ws, depth, image = torch.from_numpy(ws), torch.from_numpy(depth), torch.from_numpy(image)
        ctx.save_for_backward(sigmas.detach(), rgbs.detach(), deltas, rays, ws, image)
        ctx.T = T_thresh
        return ws, depth, image

    @staticmethod
    def backward(ctx, g_ws, g_depth, g_image):
        sigmas, rgbs, deltas, rays, ws, image = ctx.saved_tensors
        gs, gc = co.composite_rays_train_backward(g_ws.contiguous().numpy(), g_image.contiguous().numpy(), sigmas.numpy(), rgbs.numpy(),
                                                  deltas.numpy(), rays.numpy(), ws.numpy(), image.numpy(), ctx.T)
        return torch.from_numpy(gs), torch.from_numpy(gc), None, None, None


def run_cuda_eval(field, rays_o, rays_d, aabb, bound, density_bitfield, cascade, grid_size=128, dt_gamma=0, max_steps=1024, T_thresh=1e-4):
    """Inference branch (:651-688): march <= 8 steps at a time over the alive rays, composite, compact."""
    rays_o = rays_o.contiguous().view(-1, 3).numpy()
    rays_d = rays_d.contiguous().view(-1, 3).numpy()
    N = rays_o.shape[0]
    nears, fars = co.near_far_from_aabb(rays_o, rays_d, aabb.numpy(), 0.2)
    weights_sum, depth, image = np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros((N, 3), np.float32)
    rays_alive = np.arange(N, dtype=np.int32)
    rays_t = nears.copy()
    step, trace = 0, []
    while step < max_steps:
        n_alive = rays_alive.shape[0]
        if n_alive <= 0:
            break
        n_step = max(min(N // n_alive, 8), 1)
        xyzs, dirs, deltas = co.march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, cascade,
                                           grid_size, nears, fars, 128, None, dt_gamma, max_steps)
        with torch.no_grad():
            sigmas, rgbs, _ = field(torch.from_numpy(xyzs), torch.from_numpy(dirs))
        co.composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas.numpy(), rgbs[..., :3].contiguous().numpy(), deltas, weights_sum,
                          depth, image, T_thresh)
        rays_alive = np.ascontiguousarray(rays_alive[rays_alive >= 0])
        trace.append((n_alive, n_step))
        step += n_step
    return dict(image=image, depth=depth, weights_sum=weights_sum, mask=nears < fars, trace=trace)
