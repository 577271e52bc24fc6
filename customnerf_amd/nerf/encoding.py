"""nerf/encoding.py:52-71 (get_encoder) and the 27-d frequency embedder of nerf/base.py:10-77."""
import torch
import torch.nn as nn

from ..gridencoder import GridEncoder


class FreqEmbedder(nn.Module):
    """base.py:10-60: [x, sin(2^k x), cos(2^k x)] for k = 0..multires-1."""

    def __init__(self, multires=4, input_dim=3):
        super().__init__()
        self.input_dim = input_dim
        self.freq_bands = (2. ** torch.linspace(0., multires - 1, multires)).tolist()
        self.out_dim = input_dim * (1 + 2 * multires)

    def forward(self, x):
        out = [x]
        for f in self.freq_bands:
            out.append(torch.sin(x * f))
            out.append(torch.cos(x * f))
        return torch.cat(out, dim=-1)


def get_embedder(multires, input_dim=3):
    """base.py:63-77"""
    if multires < 0:
        return nn.Identity(), input_dim
    e = FreqEmbedder(multires, input_dim)
    return e, e.out_dim


def get_encoder(encoding, input_dim=3, multires=6, degree=4, num_levels=16, level_dim=2, base_resolution=16,
                log2_hashmap_size=19, desired_resolution=2048, align_corners=False, **kwargs):
    """encoding.py:52-71"""
    if encoding == 'None':
        return (lambda x, **kw: x), input_dim
    if encoding in ('hashgrid', 'tiledgrid'):
        encoder = GridEncoder(input_dim=input_dim, num_levels=num_levels, level_dim=level_dim, base_resolution=base_resolution,
                              log2_hashmap_size=log2_hashmap_size, desired_resolution=desired_resolution,
                              gridtype='hash' if encoding == 'hashgrid' else 'tiled', align_corners=align_corners)
        return encoder, encoder.output_dim
    raise NotImplementedError('Unknown encoding mode, choose from [None, hashgrid, tiledgrid]')
