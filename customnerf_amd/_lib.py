"""ctypes binding of libcustomnerf_hip.so (C-ABI in include/customnerf_hip.h).

The library is the product: there is no CPU or PyTorch fallback.  If it is missing, or an entry point is missing,
importing this module raises.  Every call returns the library's status code through `check()`, which raises
RuntimeError (launch failure) or ValueError (rejected arguments, where the reference raises std::runtime_error).
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcustomnerf_hip.so")
ABI_VERSION = 5

vp, u32, u64, f32, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_float, C.c_int

# name -> argtypes (must list every symbol include/customnerf_hip.h declares; tests/test_abi.py cross-checks)
SIGNATURES = {
    "cnerf_abi_version": [],
    "cnerf_profile_stage_events": [vp, u32],
    "cnerf_near_far_from_aabb": [vp, vp, vp, u32, f32, vp, vp, vp],
    "cnerf_sph_from_ray": [vp, vp, f32, u32, vp, vp],
    "cnerf_morton3D": [vp, u32, vp, vp],
    "cnerf_morton3D_invert": [vp, u32, vp, vp],
    "cnerf_packbits": [vp, u32, f32, vp, vp],
    "cnerf_march_rays_train": [vp, vp, vp, f32, f32, u32, u32, u32, u32, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "cnerf_march_rays_train_count": [vp, vp, vp, f32, f32, u32, u32, u32, u32, vp, vp, vp, vp, vp, vp],
    "cnerf_march_rays_train_write": [vp, vp, vp, f32, f32, u32, u32, u32, u32, u32, vp, vp, vp, vp, vp, vp, vp, vp],
    "cnerf_march_rays_train_count_hits": [vp, vp, vp, f32, f32, u32, u32, u32, u32, vp, vp, vp, vp, vp, vp, vp],
    "cnerf_march_rays_train_write_hits": [vp, vp, f32, f32, u32, u32, u32, u32, u32, vp, vp, vp, vp, vp, vp, vp, vp],
    "cnerf_composite_rays_train_forward": [vp, vp, vp, vp, u32, u32, f32, vp, vp, vp, u32, vp],
    "cnerf_composite_rays_train_backward": [vp, vp, vp, vp, vp, vp, vp, vp, u32, u32, f32, vp, vp, u32, vp],
    "cnerf_march_rays": [u32, u32, vp, vp, vp, vp, f32, f32, u32, u32, u32, vp, vp, vp, vp, vp, vp, vp, vp],
    "cnerf_composite_rays": [u32, u32, f32, vp, vp, vp, vp, vp, vp, vp, vp, u32, vp],
    "cnerf_compact_rays_alive": [vp, u32, vp, vp, vp],
    "cnerf_grid_encode_forward": [vp, vp, vp, vp, u32, u32, u32, u32, u32, f32, u32, vp, u32, i32, u32, i32, vp],
    "cnerf_grid_encode_forward_strided": [vp, vp, vp, vp, u32, u32, u32, u32, u32, f32, u32, vp, u32, i32, u32, i32, u32, vp],
    "cnerf_grid_encode_forward_ordered": [vp, vp, vp, vp, u32, u32, u32, u32, u32, f32, u32, vp, u32, i32, u32, i32, u32, u32, vp],
    "cnerf_grid_encode_backward": [vp, vp, vp, vp, u32, u32, u32, u32, u32, f32, u32, vp, vp, u32, i32, u32, i32, vp, u64, vp],
    "cnerf_grid_encode_backward_prepare": [vp, vp, u32, u32, u32, u32, u32, f32, u32, u32, i32, u32, i32, vp, u64, vp, vp],
    "cnerf_grid_encode_backward_prepared": [vp, vp, vp, vp, u32, u32, u32, u32, u32, f32, u32, u32, i32, u32, i32, vp, u64, vp],
    "cnerf_grid_encode_backward_workspace_bytes": [vp, u32, u32, u32, u32, u32, f32, u32, i32, vp],
    "cnerf_grid_encode_backward_prepare_block": [i32, vp],
    "cnerf_grid_encode_backward_needs_plan": [vp, u32, u32, u32, u32, u32, f32, u32, u32, i32, vp],
    "cnerf_grid_encode_backward_prepare_rows": [vp, vp, u32, u32, u32, u32, f32, u32, u32, i32, u32, i32, u32, u32, vp, u64, vp, vp],
    "cnerf_grid_encode_backward_prepare_finish": [vp, u32, u32, u32, u32, f32, u32, u32, u32, i32, vp, u64, vp, vp],
    "cnerf_grad_total_variation": [vp, vp, vp, vp, f32, u32, u32, u32, u32, f32, u32, u32, i32, vp],
    "cnerf_cast_f32_to_f16": [vp, vp, u64, vp],
    "cnerf_field_forward": [vp, vp, vp, u32, u32, u32, u32, u32, vp, vp, vp, vp, vp, i32, vp],
    "cnerf_field_forward_strided": [vp, vp, vp, u32, u32, u32, u32, u32, vp, vp, vp, vp, vp, i32, u32, vp],
    "cnerf_field_backward": [vp, vp, vp, u32, u32, u32, u32, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, u64, i32, vp],
    "cnerf_field_backward_ex": [vp, vp, vp, u32, u32, u32, u32, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, u64, i32, vp, vp],
    "cnerf_field_backward_workspace_bytes": [u32, u32, u32, u32, i32, vp],
    "cnerf_stream_capture_id": [vp, vp],
    "cnerf_field_weight_image_bytes": [u32, u32, u32, vp],
    "cnerf_field_pack_weights": [u32, u32, u32, vp, vp, vp, vp, u64, vp],
    "cnerf_field_forward_img": [vp, vp, vp, u32, u32, u32, u32, u32, vp, vp, vp, vp, vp, i32, u32, vp, vp],
    "cnerf_field_backward_img": [vp, vp, vp, u32, u32, u32, u32, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, u64, i32, vp, vp, vp],
    "cnerf_mlp_forward": [vp, u32, vp, u32, u32, u32, u32, u32, i32, vp, u32, i32, vp],
    "cnerf_mlp_backward": [vp, u32, vp, vp, u32, u32, u32, u32, u32, u32, i32, vp, u32, vp, vp, u64, i32, vp],
    "cnerf_mlp_backward_workspace_bytes": [u32, u32, u32, u32, u32, i32, vp],
    "cnerf_generate_rays": [vp, u32, u32, u32, f32, f32, f32, f32, f32, i32, vp, vp, vp],
    "cnerf_generate_rays_fisheye": [vp, u32, u32, u32, f32, f32, f32, f32, f32, vp, vp, vp, vp],
    "cnerf_occupancy_points": [vp, u32, f32, f32, vp, vp],
    "cnerf_occupancy_update": [vp, u32, f32, vp, vp, vp],
    "cnerf_occupancy_finalize_pack": [vp, u32, f32, vp, u32, vp, vp, vp],
    "cnerf_sample_pdf": [vp, vp, vp, u32, u32, u32, vp, vp],
    "cnerf_sample_coarse": [vp, vp, vp, vp, vp, vp, u32, u32, vp, vp, vp],
    "cnerf_sample_fine_merge": [vp, vp, vp, vp, vp, vp, vp, vp, u32, u32, u32, vp, vp, vp],
    "cnerf_sample_fine_merge_split": [vp, vp, vp, vp, vp, vp, vp, vp, u32, u32, u32, vp, vp, vp, vp, vp],
    "cnerf_sample_coarse_unit": [vp, vp, vp, vp, vp, vp, u32, u32, vp, vp, vp, f32, vp],
    "cnerf_sample_fine_merge_split_unit": [vp, vp, vp, vp, vp, vp, vp, vp, u32, u32, u32, vp, vp, vp, vp, f32, vp],
    "cnerf_composite_run": [vp, vp, vp, vp, vp, u32, u32, u32, i32, f32, vp, vp, vp],
    "cnerf_composite_run_backward": [vp, vp, vp, vp, vp, vp, u32, u32, u32, i32, f32, i32, i32, vp, vp, vp],
    "cnerf_recon_loss": [vp, vp, vp, u32, f32, f32, vp, vp, vp],
    "cnerf_recon_loss_scaled": [vp, vp, vp, u32, f32, f32, vp, vp, vp, vp],
    "cnerf_sample_coarse_unit_aabb": [vp, vp, vp, f32, vp, u32, u32, vp, vp, vp, vp, vp, f32, vp],
    "cnerf_composite_run_indexed": [vp, vp, vp, vp, vp, u32, u32, u32, i32, f32, vp, vp, vp, vp, vp, vp],
    "cnerf_composite_run_indexed_variants": [vp, vp, vp, vp, vp, u32, u32, u32, i32, f32, vp, vp, vp, vp, vp, u32, vp],
    "cnerf_composite_run_backward_indexed": [vp, vp, vp, vp, vp, vp, u32, u32, u32, i32, f32, i32, i32, vp, vp, vp, vp],
    "cnerf_composite_run_backward_indexed_flush": [vp, vp, vp, vp, vp, vp, u32, u32, u32, i32, f32, i32, i32, vp, vp, vp, i32, vp, vp],
    "cnerf_adam_step": [vp, vp, vp, vp, vp, u64, f32, f32, f32, f32, u32, f32, i32, vp],
    "cnerf_scaler_check": [vp, u64, vp, vp],
    "cnerf_scaler_watch": [vp],
    "cnerf_grid_backward_adam": [vp],
    "cnerf_grid_backward_adam_consumed": [vp],
    "cnerf_adam_step_scaled": [vp, vp, vp, vp, vp, u64, f32, f32, f32, f32, vp, f32, i32, vp],
    "cnerf_dp_pack": [vp, vp, u64, f32, vp],
    "cnerf_dp_reduce": [vp, u32, u64, vp, vp, vp],
    "cnerf_scaler_update": [vp, f32, f32, u32, vp],
    "cnerf_adam_step_scaled_multi": [vp, f32, f32, f32, vp, f32, i32, i32, f32, f32, u32, vp],
    # ---- include/customnerf_sd.h (score-distillation primitives)
    "cnerf_sd_gemm": [vp, vp, u64, vp],
    "cnerf_sd_gemm_workspace_bytes": [vp, vp],
    "cnerf_sd_gemm_serves_ln": [vp, vp],
    "cnerf_sd_groupnorm_forward": [vp, vp, vp, u32, u32, u32, u32, f32, i32, vp, i32, vp, vp],
    "cnerf_sd_groupnorm_backward": [vp, vp, vp, vp, u32, u32, u32, u32, f32, i32, vp, vp, vp, vp],
    "cnerf_sd_groupnorm_backward_ex": [vp, vp, vp, vp, u32, u32, u32, u32, f32, i32, vp, vp, i32, vp, vp, vp],
    "cnerf_sd_layernorm_forward": [vp, vp, vp, u32, u32, f32, vp, vp],
    "cnerf_sd_softmax_forward": [vp, u64, u32, u32, vp],
    "cnerf_sd_softmax_backward": [vp, vp, u64, u32, u32, vp],
    "cnerf_sd_attention": [vp, vp, vp, vp, u32, u32, u32, u32, u32, u32, u64, u32, u64, u32, u64, u32, u64, i32, vp],
    "cnerf_sd_attention_v": [vp, vp, vp, vp, u32, u32, u32, u32, u32, u32, u64, u32, u64, u32, u64, u32, u64, i32, vp],
    "cnerf_sd_geglu": [vp, u64, u32, vp, vp],
    "cnerf_sd_transpose": [vp, vp, u32, u32, u32, u32, u32, u64, u64, vp],
    "cnerf_sd_image_to_vae_input": [vp, u32, u32, u32, u32, u32, vp, vp],
    "cnerf_sd_image_to_vae_input_backward": [vp, u32, u32, u32, u32, u32, vp, vp],
    "cnerf_sd_clip_preprocess": [vp, u32, u32, u32, u32, vp, vp, vp, vp],
    "cnerf_sd_patchify": [vp, u32, u32, u32, vp, vp],
    "cnerf_sd_timestep_embedding": [vp, u32, u32, vp, vp],
    "cnerf_sd_add_noise": [vp, vp, f32, u32, vp, vp],
    "cnerf_sd_sds_grad": [vp, u32, vp, f32, f32, f32, u32, vp, vp],
    "cnerf_edit_ray_images": [vp, u32, u32, vp, vp, vp, vp],
    "cnerf_edit_ray_images_backward": [vp, vp, vp, u32, u32, vp, vp],
    "cnerf_edit_l1_loss": [vp, vp, u32, f32, vp, vp, vp],
    "cnerf_edit_sds_loss": [vp, vp, u32, vp, vp, vp],
    "cnerf_edit_scale_by_scalar": [vp, vp, f32, u32, vp, vp],
    "cnerf_sd_sample_latents": [vp, vp, u32, u32, f32, vp, vp],
    "cnerf_sd_sample_latents_backward": [vp, vp, vp, u32, u32, f32, vp, vp],
    "cnerf_set_floats": [vp, vp, u32, vp],
    "cnerf_sd_add": [vp, vp, u64, vp, vp],
    "cnerf_sd_silu": [vp, u64, vp, vp],
    "cnerf_sd_concat": [vp, vp, u64, u32, u32, vp, vp],
    "cnerf_sd_concat_gn": [vp, vp, u64, u32, u32, vp, vp, u32, u32, vp],
}


class SdGemmDesc(C.Structure):
    """struct CnerfSdGemm of include/customnerf_sd.h (field order and types must match)."""
    _fields_ = [("A", vp), ("B", vp), ("C", vp), ("C32", vp), ("bias", vp), ("bias_rows", vp), ("residual", vp),
                ("M", u32), ("N", u32), ("K", u32), ("lda", u32), ("ldb", u32), ("ldc", u32), ("ldr", u32),
                ("rows_per_bias_row", u32), ("ld_bias_rows", u32), ("alpha", f32), ("act", C.c_int32), ("batch_outer", u32), ("batch_inner", u32),
                ("sa_o", u64), ("sa_i", u64), ("sb_o", u64), ("sb_i", u64), ("sc_o", u64), ("sc_i", u64),
                ("mode", C.c_int32), ("Cin", u32), ("H_in", u32), ("W_in", u32), ("H_out", u32), ("W_out", u32),
                ("KH", u32), ("KW", u32), ("stride", u32), ("pad_t", u32), ("pad_l", u32), ("ups", u32), ("tstride", u32),
                ("gn_sums", vp), ("gn_groups", u32), ("gn_rows", u32),
                ("ln_out", vp), ("ln_gamma", vp), ("ln_beta", vp), ("ln_eps", f32)]

ADAM_MAX_JOBS = 8


class AdamJobs(C.Structure):
    """struct CnerfAdamJobs of include/customnerf_hip.h"""
    _fields_ = [("p", vp * ADAM_MAX_JOBS), ("g", vp * ADAM_MAX_JOBS), ("m", vp * ADAM_MAX_JOBS), ("v", vp * ADAM_MAX_JOBS), ("p_half", vp * ADAM_MAX_JOBS),
                ("n", u64 * ADAM_MAX_JOBS), ("lr", f32 * ADAM_MAX_JOBS), ("n_jobs", u32)]


class GridAdam(C.Structure):
    """struct CnerfGridAdam of include/customnerf_hip.h"""
    _fields_ = [("p", vp), ("g", vp), ("m", vp), ("v", vp), ("p_half", vp), ("n", u64), ("lr", f32), ("beta1", f32), ("beta2", f32), ("eps", f32),
                ("scaler_state", vp), ("extra_inv", f32), ("zero_grad", C.c_int)]


F32, F16 = 0, 1


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is the product and has no fallback. "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C customnerf_amd/csrc`.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing: fail loudly
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.cnerf_target_arch.restype = C.c_char_p
    if lib.cnerf_abi_version() != ABI_VERSION:
        raise ImportError(f"ABI mismatch: library {lib.cnerf_abi_version()} vs binding {ABI_VERSION}")
    return lib


lib = _load()


def check(rc, what=""):
    if rc == 0:
        return
    if rc < 0:
        raise ValueError(f"customnerf_hip: {what} rejected its arguments (code {rc})")
    raise RuntimeError(f"customnerf_hip: {what} failed with hipError {rc}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("customnerf_amd: tensors must live on the GPU (HIP device memory); there is no CPU path")


def dtype_id(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.float16:
        return F16
    raise ValueError(f"unsupported dtype {t.dtype}")


def scratch_key(device):
    """Key of a per-device scratch buffer that must not be shared between concurrently running streams (the micro-batch pipeline of
    trainer.ReconTrainer runs two half-batches on two streams): (device, stream handle)."""
    import torch
    return (device, torch.cuda.current_stream(device).cuda_stream)


_SCRATCH_GENERATION = [0]


def scratch_reallocated():
    """Call whenever a cached grow-on-demand scratch buffer (scatter workspaces, field / MLP partial-gradient rows, the march's probe list)
    is replaced: captured hipGraphs hold the OLD buffer's address, so their owners (trainer.ReconTrainer.train_step_graphed) compare
    scratch_generation() with the value at capture and drop every cached graph when it moved."""
    _SCRATCH_GENERATION[0] += 1


def scratch_generation():
    return _SCRATCH_GENERATION[0]


_GRAD_CHAIN = {}


def grad_chain_wait(device):
    """Kernels that read-modify-write the trainers' shared persistent .grad buffers (grid scatter, field partial reduction) must not
    overlap each other across streams: wait for the previous one ...  On ONE stream the launch order already is that order, and nothing is
    enqueued (round 6: an event record per producer was a 5.8 us bubble on the queue, two per step — scratch/recon_gaps.sh); a producer on
    another stream waits for the tail of the previous producer's stream (a later point than the producer itself: conservative)."""
    import torch
    cur = torch.cuda.current_stream(device)
    last = _GRAD_CHAIN.get(device)
    if last is not None and last.cuda_stream != cur.cuda_stream:
        cur.wait_stream(last)


def grad_chain_record(device):
    """... and remember which stream the next one has to order itself after."""
    import torch
    _GRAD_CHAIN[device] = torch.cuda.current_stream(device)
