"""Shim so that the reference's `import raymarching` (nerf/renderer.py:16) resolves to the MI355X implementation.
Put `customnerf_amd/dropin` on PYTHONPATH ahead of the reference's own `raymarching/` package."""
from customnerf_amd.raymarching import *  # noqa: F401,F403
