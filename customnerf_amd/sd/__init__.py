"""Score-distillation (SDS) half of the hot path: SD-1.5 VAE encoder (forward + input gradient), UNet eps-prediction,
CFG / SDS gradient and the LGIE editing step, on libcustomnerf_hip.so (include/customnerf_sd.h)."""
from .guidance import StableDiffusion  # noqa: F401
