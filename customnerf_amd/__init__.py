"""customnerf_amd — MI355X-native (gfx950) hot path of CustomNeRF behind the reference's own call surface.

Sub-packages mirror the reference's import names:
    customnerf_amd.raymarching   <- reference `raymarching`   (raymarching/raymarching.py)
    customnerf_amd.gridencoder   <- reference `gridencoder`   (gridencoder/grid.py)
    customnerf_amd.tcnn          <- `tinycudann` (Network)    (call sites nerf/network_grid.py:18-54, 98-139)
    customnerf_amd.nerf          <- reference `nerf` renderer / field / guidance
All device arithmetic lives in libcustomnerf_hip.so (include/customnerf_hip.h); importing a sub-package raises if
that library is missing.
"""
__version__ = "0.1.0"
