"""Counterpart of the reference's `nerf` package, restricted to the hot path:
renderer.py (NeRFRenderer.run / run_cuda / render / update_extra_state / weights_sum_i, sample_pdf),
network_grid.py (NeRFNetwork), encoding.py (get_encoder, frequency embedder), provider_utils.py
(trunc_exp, safe_normalize, get_rays / generate_rays)."""
