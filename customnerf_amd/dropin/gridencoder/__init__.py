"""Shim for the reference's `from gridencoder import GridEncoder` (nerf/encoding.py:58,62)."""
from customnerf_amd.gridencoder import GridEncoder, grid_encode  # noqa: F401
