"""Shim for the reference's `import tinycudann as tcnn` (nerf/network_grid.py:10): Network with a flat `params`."""
from customnerf_amd.tcnn import Network, NetworkWithInputEncoding, set_default_dtype  # noqa: F401
