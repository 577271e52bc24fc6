/*
 * customnerf_hip.h — C-ABI of libcustomnerf_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for CustomNeRF's volumetric-rendering hot path.  Each entry point replaces one function of
 * the reference's two native torch extensions (pybind modules `_raymarching`, `_gridencoder`) or one
 * third-party call on the path (tinycudann FullyFusedMLP).  Reference interfaces are cited per function as
 * file:line under /root/reference.
 *
 * Conventions (all entry points):
 *   - plain pointers + sizes, no torch types.  Pointers are DEVICE pointers unless the name ends in `_host`.
 *   - caller owns every buffer; nothing is allocated, nothing synchronises with the host; work is enqueued on
 *     `stream` (a hipStream_t passed as void*; NULL = the null stream).
 *   - returns 0 on success, a positive hipError_t if a launch failed, or a negative CNERF_E* code for rejected
 *     arguments (the reference raises std::runtime_error / TORCH_CHECK there: gridencoder.cu:380,397,448-464).
 *   - float32 data unless stated; `dtype` 0 = float32, 1 = float16 (IEEE binary16).
 */
#ifndef CUSTOMNERF_HIP_H
#define CUSTOMNERF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CNERF_OK 0
#define CNERF_EINVAL (-1)      /* unsupported D / C / dtype / size */
#define CNERF_ENULL (-2)       /* required pointer is NULL */

#define CNERF_F32 0
#define CNERF_F16 1

/* ABI version of this header; cnerf_abi_version() of the loaded library must match.
 * 2: the GroupNorm / GEMM-epilogue statistics buffers of customnerf_sd.h are int64[B][G][2] fixed point (were float[B][G][2]).
 * 4: struct CnerfSdGemm grew the ln_* fields (LayerNorm of the output rows in the split-K tail); new entry points
 *    cnerf_grid_encode_forward_ordered, cnerf_sd_concat_gn, cnerf_sd_gemm_serves_ln, cnerf_profile_stage_events. */
#define CNERF_ABI_VERSION 5
int cnerf_abi_version(void);
/* name of the code object's target ("gfx950") */
const char *cnerf_target_arch(void);
/* Measurement aid (bench.py `variants.small_batch`; no reference counterpart): hipEvent_t handles that the multi-kernel entry points record on
 * their launch stream between their kernels, or NULL entries / n = 0 to switch it off (the default).  Slots:
 *   0 / 1 before / after the scatter's emit kernel, 2 after its accumulate kernel, 3 after its split-bin reduction (cnerf_grid_encode_backward*);
 *   4 / 5 before / after the field backward's main kernel, 6 after its partial-gradient reduction (cnerf_field_backward*).
 * The handles are borrowed: the caller keeps them alive until it clears the slots. */
#define CNERF_STAGE_EVENTS 8
int cnerf_profile_stage_events(void *const *events, uint32_t n);

/* ------------------------------------------------------------------------------------------------
 * _raymarching  (reference: raymarching/src/raymarching.h:7-21, bindings.cpp:5-20)
 * ---------------------------------------------------------------------------------------------- */

/* near_far_from_aabb — raymarching.h:7, kernel raymarching.cu:91-145.
 * rays_o, rays_d [N,3]; aabb [6]; nears, fars [N] (miss: both = FLT_MAX). */
int cnerf_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb, uint32_t N,
                             float min_near, float *nears, float *fars, void *stream);

/* sph_from_ray — raymarching.h:8, kernel raymarching.cu:162-198.  coords [N,2]. */
int cnerf_sph_from_ray(const float *rays_o, const float *rays_d, float radius, uint32_t N, float *coords, void *stream);

/* morton3D / morton3D_invert — raymarching.h:9-10, kernels raymarching.cu:214-226, 237-254. */
int cnerf_morton3D(const int32_t *coords, uint32_t N, int32_t *indices, void *stream);
int cnerf_morton3D_invert(const int32_t *indices, uint32_t N, int32_t *coords, void *stream);

/* packbits — raymarching.h:11, kernel raymarching.cu:267-289.  grid [N*8] floats -> bitfield [N] bytes. */
int cnerf_packbits(const float *grid, uint32_t N, float density_thresh, uint8_t *bitfield, void *stream);

/* Occupancy-grid refresh — NeRFRenderer.update_extra_state (nerf/renderer.py:1658-1715) without its Python loops, temporary grid and
 * host synchronisations.  Per cascade: cnerf_occupancy_points -> the caller's density query -> cnerf_occupancy_update; then once
 * cnerf_occupancy_finalize_pack.
 *   points:   rand [H^3, 3] U[0,1) (the `torch.rand_like` draw of :1695) -> xyzs [H^3, 3], cell i = (x, y, z) in meshgrid('ij') order
 *             (:1679-1684): xyzs = (2 c / (H-1) - 1) * (cas_bound - half_grid) + (2 rand - 1) * half_grid  (:1690-1695);
 *   update:   sigmas [H^3] in the same order; density_grid_cascade [H^3] (Morton order, :1688 + :1697) gets max(old * decay, sigma) where
 *             old >= 0 (:1707-1709); partials [ceil(H^3 / 256)][2] doubles receive (sum, count) of the valid cells of this cascade;
 *   finalize: partials of ALL cascades [n_partials][2] -> state[0] = mean density of the valid cells (:1710), state[1] = min(mean,
 *             density_thresh) (:1712); bitfield [n_bytes] = packbits(density_grid [n_bytes * 8], state[1]) (:1713, raymarching.cu:267-289). */
int cnerf_occupancy_points(const float *rand, uint32_t H, float cas_bound, float half_grid, float *xyzs, void *stream);
int cnerf_occupancy_update(const float *sigmas, uint32_t H, float decay, float *density_grid_cascade, double *partials, void *stream);
int cnerf_occupancy_finalize_pack(const double *partials, uint32_t n_partials, float density_thresh, const float *density_grid,
                                  uint32_t n_bytes, float *state, uint8_t *bitfield, void *stream);

/* march_rays_train — raymarching.h:13, kernel raymarching.cu:311-480.
 * Same arguments as the reference binding.  Unlike the reference (atomics-ordered, nondeterministic slots) the
 * sample segments are laid out in RAY ORDER by an exclusive scan: rays[n] = (n, offset_n, num_steps_n).
 * counter[0] += total samples, counter[1] += N.  Rays with offset+num_steps > M are dropped (as :416); a call that overflows its
 * budget (counter[0] + total > M) starts the scan at ray floor(noises[0] * N) and wraps, so that the dropped rays move with the
 * per-call jitter draw instead of always being the highest-numbered ones (the reference drops in atomics arrival order).
 * xyzs/dirs [M,3], deltas [M,2], rays [N,3] int32, counter [2] int32, noises [N]. */
int cnerf_march_rays_train(const float *rays_o, const float *rays_d, const uint8_t *grid, float bound, float dt_gamma,
                           uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float *nears,
                           const float *fars, float *xyzs, float *dirs, float *deltas, int32_t *rays, int32_t *counter,
                           const float *noises, void *stream);
/* The same operation split in its two passes so a caller can size xyzs/dirs/deltas from counter[0] instead of
 * pre-allocating N*max_steps samples (raymarching.py:197,206-208): _count fills rays[] and counter[];
 * _write (given the rays[] produced by _count) writes the samples. */
int cnerf_march_rays_train_count(const float *rays_o, const float *rays_d, const uint8_t *grid, float bound, float dt_gamma,
                                 uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, const float *nears,
                                 const float *fars, int32_t *rays, int32_t *counter, const float *noises, void *stream);
int cnerf_march_rays_train_write(const float *rays_o, const float *rays_d, const uint8_t *grid, float bound, float dt_gamma,
                                 uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float *nears,
                                 const float *fars, float *xyzs, float *dirs, float *deltas, const int32_t *rays,
                                 const float *noises, void *stream);
/* The two passes without the second march: _count_hits additionally records (t, dt) of every occupied probe of ray n at
 * hits[n][step] (hits float32 [N, max_steps, 2], 8-byte aligned, caller-owned scratch; only the first num_steps_n entries of a row are
 * written); _write_hits turns the list into xyzs / dirs / deltas — one wave per ray, no occupancy lookups — with the arithmetic of
 * _write, bit for bit. */
int cnerf_march_rays_train_count_hits(const float *rays_o, const float *rays_d, const uint8_t *grid, float bound, float dt_gamma,
                                      uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, const float *nears,
                                      const float *fars, int32_t *rays, int32_t *counter, const float *noises, float *hits,
                                      void *stream);
int cnerf_march_rays_train_write_hits(const float *rays_o, const float *rays_d, float bound, float dt_gamma, uint32_t max_steps,
                                      uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float *nears, const float *noises,
                                      const float *hits, const int32_t *rays, float *xyzs, float *dirs, float *deltas,
                                      void *stream);

/* composite_rays_train_forward / _backward — raymarching.h:14-15 (the `_sdf` twins :16-17 are byte-identical
 * duplicates in the reference and map to the same entry points); kernels raymarching.cu:500-577, 691-772.
 * rgbs has `rgb_stride` floats per sample (3 in the reference binding; 4 lets a caller pass the field's
 * rgb+confidence rows without a slice/copy — renderer.py:630-635 passes such a tensor). */
int cnerf_composite_rays_train_forward(const float *sigmas, const float *rgbs, const float *deltas, const int32_t *rays,
                                       uint32_t M, uint32_t N, float T_thresh, float *weights_sum, float *depth,
                                       float *image, uint32_t rgb_stride, void *stream);
int cnerf_composite_rays_train_backward(const float *grad_weights_sum, const float *grad_image, const float *sigmas,
                                        const float *rgbs, const float *deltas, const int32_t *rays,
                                        const float *weights_sum, const float *image, uint32_t M, uint32_t N,
                                        float T_thresh, float *grad_sigmas, float *grad_rgbs, uint32_t rgb_stride,
                                        void *stream);

/* march_rays / composite_rays (inference) — raymarching.h:19-20, kernels raymarching.cu:884-989, 1002-1089. */
int cnerf_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t,
                     const float *rays_o, const float *rays_d, float bound, float dt_gamma, uint32_t max_steps,
                     uint32_t C, uint32_t H, const uint8_t *grid, const float *nears, const float *fars, float *xyzs,
                     float *dirs, float *deltas, const float *noises, void *stream);
int cnerf_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *rays_alive, float *rays_t,
                         const float *sigmas, const float *rgbs, const float *deltas, float *weights_sum, float *depth,
                         float *image, uint32_t rgb_stride, void *stream);
/* Order-preserving device-side compaction replacing `rays_alive = rays_alive[rays_alive >= 0]`
 * (renderer.py:685): out[0..count) = the non-negative entries of in[0..n) in order, *count = their number.
 * Wave ballot + prefix scan; `count` is a device int32. */
int cnerf_compact_rays_alive(const int32_t *rays_alive_in, uint32_t n, int32_t *rays_alive_out, int32_t *count, void *stream);

/* ------------------------------------------------------------------------------------------------
 * _gridencoder  (reference: gridencoder/src/gridencoder.h:12-15, bindings.cpp:5-7)
 * ---------------------------------------------------------------------------------------------- */

/* grid_encode_forward — gridencoder.h:12, kernel gridencoder.cu:87-244.
 * inputs [B,D] in [0,1]; embeddings [offsets[L], C] (dtype); outputs [L,B,C] (dtype); dy_dx [B, L*D*C] (dtype) or NULL.
 * offsets_host: HOST int32 [L+1] (the reference passes a device tensor; the level geometry — scale, resolution,
 * table size — is derived from it on the host and travels as kernel arguments).
 * S = log2(per_level_scale), H = base resolution.  gridtype 0 hash / 1 tiled; interp 0 linear / 1 smoothstep.
 * D in {2,3,4,5}, C in {1,2,4,8}, L <= 32 — else CNERF_EINVAL (gridencoder.cu:380,397). */
int cnerf_grid_encode_forward(const float *inputs, const void *embeddings, const int32_t *offsets_host, void *outputs,
                              uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level, float S, uint32_t H,
                              void *dy_dx, uint32_t gridtype, int align_corners, uint32_t interp, int dtype, void *stream);
/* Same, writing into a larger output buffer: outputs [L, out_level_stride, C] with out_level_stride >= B rows per level (0 = B);
 * the caller offsets `outputs` to the first row it wants written.  Lets several gathers (the coarse and the importance samples of
 * NeRFRenderer.run, renderer.py:327-363) fill one feature buffer, so that the coarse samples are not gathered a second time. */
int cnerf_grid_encode_forward_strided(const float *inputs, const void *embeddings, const int32_t *offsets_host, void *outputs,
                                      uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level, float S, uint32_t H,
                                      void *dy_dx, uint32_t gridtype, int align_corners, uint32_t interp, int dtype,
                                      uint32_t out_level_stride, void *stream);
/* Same with the traversal of the sample list chosen by the caller (results are bit-identical either way; only the specialised
 * fp16 / D = 3 / C = 2 / linear kernel has two forms, every other configuration ignores it):
 *   CNERF_GRID_LEVEL_MAJOR  0  one level at a time per XCD (its table stays in that XCD's L2): right for samples spread over the volume
 *                              — the stratified samples of NeRFRenderer.run (renderer.py:317-325) — and the default of the entry points above;
 *   CNERF_GRID_SAMPLE_MAJOR 1  all levels of a tile of consecutive samples per workgroup: right when neighbours of the list are neighbours
 *                              in space — the importance samples (renderer.py:340-352) of a field that has a surface. */
#define CNERF_GRID_LEVEL_MAJOR 0
#define CNERF_GRID_SAMPLE_MAJOR 1
int cnerf_grid_encode_forward_ordered(const float *inputs, const void *embeddings, const int32_t *offsets_host, void *outputs,
                                      uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level, float S, uint32_t H,
                                      void *dy_dx, uint32_t gridtype, int align_corners, uint32_t interp, int dtype,
                                      uint32_t out_level_stride, uint32_t traversal, void *stream);

/* grid_encode_backward — gridencoder.h:13, kernels gridencoder.cu:247-339 (+ :342-368 when dy_dx != NULL).
 * grad [L,B,C] (dtype).  grad_embeddings is ALWAYS float32 [offsets[L], C], pre-zeroed by the caller
 * (grid.py:83) and accumulated with float32 atomics (the reference uses __half2 atomics for fp16 tables).
 * grad_inputs float32 [B,D] (written, not accumulated) when dy_dx != NULL.
 * workspace (optional, 256-byte aligned device scratch of at least cnerf_grid_encode_backward_workspace_bytes()):
 * when given, large D=3/C=2 scatters run the atomic-free binned path (records partitioned by 4096-entry table
 * chunk, summed in LDS — exact 64-bit fixed point for fp16 tables, hence bit-reproducible — and added to grad_embeddings
 * with plain coalesced read-modify-writes); with NULL, or for other shapes / small B, the scatter uses global float
 * atomics.  Both produce the same sums up to float reassociation (fp16: up to the rounding of each w*g product). */
int cnerf_grid_encode_backward(const void *grad, const float *inputs, const int32_t *offsets_host, float *grad_embeddings,
                               uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level, float S, uint32_t H,
                               const void *dy_dx, float *grad_inputs, uint32_t gridtype, int align_corners,
                               uint32_t interp, int dtype, void *workspace, uint64_t workspace_bytes, void *stream);
/* Two-phase form of the binned backward.  _prepare runs the part that needs only the sample coordinates (per-chunk histogram and
 * scans) into `workspace` — issue it early, on a second stream, and it overlaps the forward pass and the field backward; *prepared
 * tells whether the binned path applies to this shape (0: nothing was launched, use cnerf_grid_encode_backward).  _prepared then runs
 * the gradient-dependent part (record emit + LDS accumulation) and must see the same inputs, shape and workspace, after _prepare's
 * work has completed (event / stream order is the caller's). */
int cnerf_grid_encode_backward_prepare(const float *inputs, const int32_t *offsets_host, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                       uint32_t max_level, float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp,
                                       int dtype, void *workspace, uint64_t workspace_bytes, int *prepared, void *stream);
/* The plan in pieces (float16 records): the histogram of rows [row0, row0 + rows) of the B-sample list can be taken as soon as THOSE coordinates
 * exist — run()'s coarse block at the very beginning of the forward, its fine block after the importance sampling — so that all of it is done
 * before the field backward starts (a histogram that is still running then slows that kernel from 407 to 491 us: DESIGN.md §4 round 3).
 * row0 must be a multiple of *block_points (cnerf_grid_encode_backward_prepare_block) and the range must end on a block border or at B;
 * ..._finish runs the scans once every row has been counted.  *prepared = 0: nothing launched (the shape takes the atomic kernel / float32 records).
 * Round 6: the histogram-driven float16 form these pieces belonged to is gone (the histogram-free form took over its shapes); the three entry points
 * stay in the ABI and report *block_points = 0 / *prepared = 0 for every shape. */
int cnerf_grid_encode_backward_prepare_block(int dtype, uint32_t *block_points);
/* *needs_plan = 1 when cnerf_grid_encode_backward of this shape profits from a plan prepared ahead of time (the forms that need the exact record
 * counts before the emit: float32 records; float16 records on hashed levels smaller than one 4096-entry bin or of more than 512 bins); 0 when there
 * is nothing to prepare — the atomic kernel, or the scatter that counts inside its emit kernel (round 5; round 6: up to 512 bins per level, hash and
 * tiled grids — the benchmark table and the reference field's own T = 2^21 table).
 * The _prepare* entry points report *prepared = 0 for such shapes; this query lets a caller skip them (and their workspace) altogether. */
int cnerf_grid_encode_backward_needs_plan(const int32_t *offsets_host, uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level, float S,
                                          uint32_t H, uint32_t gridtype, int dtype, int *needs_plan);
int cnerf_grid_encode_backward_prepare_rows(const float *inputs, const int32_t *offsets_host, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                            float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp, int dtype, uint32_t row0,
                                            uint32_t rows, void *workspace, uint64_t workspace_bytes, int *prepared, void *stream);
int cnerf_grid_encode_backward_prepare_finish(const int32_t *offsets_host, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                              uint32_t gridtype, uint32_t interp, int dtype, void *workspace, uint64_t workspace_bytes,
                                              int *prepared, void *stream);
int cnerf_grid_encode_backward_prepared(const void *grad, const float *inputs, const int32_t *offsets_host, float *grad_embeddings,
                                        uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level, float S, uint32_t H,
                                        uint32_t gridtype, int align_corners, uint32_t interp, int dtype, void *workspace,
                                        uint64_t workspace_bytes, void *stream);
/* *bytes = scratch size the binned scatter wants for this problem (0: the atomic path will be used). */
int cnerf_grid_encode_backward_workspace_bytes(const int32_t *offsets_host, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                               uint32_t max_level, float S, uint32_t H, int dtype, uint64_t *bytes);

/* grad_total_variation — gridencoder.h:15, kernel gridencoder.cu:505-609.  float32 only (grid.py:171). */
int cnerf_grad_total_variation(const float *inputs, const float *embeddings, float *grad, const int32_t *offsets_host,
                               float weight, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                               uint32_t gridtype, int align_corners, void *stream);

/* float32 -> float16 shadow copy of a table (what `embeddings.to(torch.half)` does under autocast, grid.py:45-46). */
int cnerf_cast_f32_to_f16(const float *src, void *dst, uint64_t n, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Fused field evaluation: NeRFNetwork.forward / .density (nerf/network_grid.py:159-193) in one launch on the matrix
 * cores — the tinycudann FullyFusedMLP x3 of the reference (network_grid.py:98-139) plus its glue:
 *   fea   = MLP_net(enc)                                 enc_dim -> 64 x n_hidden_geo -> 64   (no output activation)
 *   sigma = exp(MLP_den(fea)[0] + 5 exp(-|x|^2 / 0.08))  64 -> 64 -> 1   (trunc_exp fwd provider_utils.py:20-22; blob :150-156)
 *   rgbc  = sigmoid(MLP_rgb([freq(d) (27), fea (64)]))   91 -> 64 -> n_rgb_out (3, or 4 = rgb + confidence; base.py:42-60)
 * MLPs are bias-free, ReLU hidden; params_* are the float32 flat vectors of tcnn.Network (row-major [out,in] matrices,
 * in padded to x16, out padded to x16: 64*pad16(enc_dim) + 4096*n_hidden_geo, 5120, 7168 floats).
 * enc is the grid encoder output in ITS kernel layout [L, P, 2] (dtype), so no permute copy is needed (grid.py:49,63).
 * xyz [P,3]; dirs [ceil(P/dir_group), 3]: one direction per dir_group consecutive samples (1 = per sample).
 * sigma float32 [P]; rgbc float32 [P,4] 16-byte aligned (NULL => density only; channel 3 is 0 when n_rgb_out == 3).
 * dtype CNERF_F16: fp16 weights/activations, fp32 accumulate (v_mfma_f32_32x32x16_f16), outputs rounded to fp16 values
 * — tcnn's numerics; CNERF_F32: exact float32 (v_mfma_f32_32x32x2_f32).
 * ---------------------------------------------------------------------------------------------- */
int cnerf_field_forward(const void *enc, const float *xyz, const float *dirs, uint32_t dir_group, uint32_t P, uint32_t enc_dim,
                        uint32_t n_hidden_geo, uint32_t n_rgb_out, const float *params_net, const float *params_den,
                        const float *params_rgb, float *sigma, float *rgbc, int dtype, void *stream);
/* Same, reading the first P samples of an enc buffer that holds enc_level_stride >= P samples per level (0 = P). */
int cnerf_field_forward_strided(const void *enc, const float *xyz, const float *dirs, uint32_t dir_group, uint32_t P, uint32_t enc_dim,
                                uint32_t n_hidden_geo, uint32_t n_rgb_out, const float *params_net, const float *params_den,
                                const float *params_rgb, float *sigma, float *rgbc, int dtype, uint32_t enc_level_stride,
                                void *stream);

/* Backward of cnerf_field_forward (activations are recomputed, nothing is saved by the forward).
 * grad_sigma [P] and grad_rgbc [P,4] (16-byte aligned) in; grad_enc [L,P,2] (dtype) out = d(loss)/d(grid features) in the
 * encoder's kernel layout (feeds cnerf_grid_encode_backward directly); grad_params_* float32, ACCUMULATED (caller
 * pre-zeroes).  trunc_exp backward clamps the exponent to [-15,15] (provider_utils.py:26-29); positions and directions
 * receive no gradient on this path.  workspace: 16-byte aligned scratch of cnerf_field_backward_workspace_bytes(). */
int cnerf_field_backward(const void *enc, const float *xyz, const float *dirs, uint32_t dir_group, uint32_t P, uint32_t enc_dim,
                         uint32_t n_hidden_geo, uint32_t n_rgb_out, const float *params_net, const float *params_den,
                         const float *params_rgb, const float *grad_sigma, const float *grad_rgbc, void *grad_enc,
                         float *grad_params_net, float *grad_params_den, float *grad_params_rgb, void *workspace,
                         uint64_t workspace_bytes, int dtype, void *stream);
/* cnerf_field_backward with early termination: tile_live (optional) = the uint8 [ceil(P / 32)] flags of
 * cnerf_composite_run_backward_indexed_flush — a zero byte promises that grad_sigma / grad_rgbc of rows 32 k .. 32 k + 31 are exactly zero; the
 * kernel then skips the tile (its rows of grad_enc are written as zeros, the weight gradients are bit-identical: skipped tiles add exact zeros). */
int cnerf_field_backward_ex(const void *enc, const float *xyz, const float *dirs, uint32_t dir_group, uint32_t P, uint32_t enc_dim,
                            uint32_t n_hidden_geo, uint32_t n_rgb_out, const float *params_net, const float *params_den,
                            const float *params_rgb, const float *grad_sigma, const float *grad_rgbc, void *grad_enc,
                            float *grad_params_net, float *grad_params_den, float *grad_params_rgb, void *workspace,
                            uint64_t workspace_bytes, int dtype, const uint8_t *tile_live, void *stream);
int cnerf_field_backward_workspace_bytes(uint32_t P, uint32_t enc_dim, uint32_t n_hidden_geo, uint32_t n_rgb_out, int dtype,
                                         uint64_t *bytes);
/* Packed weights (ABI 5).  Every launch above re-derives its fp16 MFMA fragment image of the three MLPs from the float32 parameters (14 us per
 * launch, three launches per training step).  cnerf_field_pack_weights writes that image once — `image`: 16-byte aligned device buffer of
 * cnerf_field_weight_image_bytes() bytes; the caller repacks after every change of the parameters — and the _img variants copy it into LDS
 * instead (CNERF_F16 only: with weight_image == NULL or CNERF_F32 they ARE cnerf_field_forward_strided / cnerf_field_backward_ex; the
 * narrow-encoding backward — enc_dim <= 16 — accepts an image and ignores it).  Results are bit-identical: the image holds the very halves
 * the kernels would have staged.  The reference has no counterpart (tcnn keeps fp16 parameters: network_grid.py:98-139). */
/* The table's optimiser step inside the backward scatter (ABI 5).  cnerf_grid_encode_backward* is where a table entry's gradient becomes final, and
 * cnerf_adam_step_scaled would read it back — with the parameter and both moments — a few launches later (62 us at the HBM roofline for the benchmark
 * table).  cnerf_grid_backward_adam(cfg) ARMS that step for the NEXT backward pass whose grad_embeddings == cfg->g (one shot; NULL disarms): if that
 * pass runs the histogram-free binned scatter (fp16, 3-D, C = 2) over ALL levels of the table (cfg->n floats), every entry leaves the pass updated
 * as cnerf_adam_step_scaled(p, g, m, v, p_half, n, lr, beta1, beta2, eps, scaler_state, extra_inv, zero_grad) would have left it — bit for bit, the
 * skip on found_inf included — and cnerf_grid_backward_adam_consumed() reports 1 (reads and clears); otherwise nothing is applied, it reports 0 and the
 * caller steps the table as usual.  The caller guarantees that this backward pass is the ONLY contribution to g in this optimiser step and that the
 * scaler's found_inf is final when the scatter runs (cnerf_scaler_watch covers the field's own producers).  Host-side switches, no launch.
 * No counterpart in the reference (torch.optim.Adam after loss.backward(): main.py:182, utils_init_nerf.py:608-616). */
typedef struct CnerfGridAdam {
    float *p, *g, *m, *v;            /* parameter, its gradient table, Adam moments: n floats each */
    void *p_half;                    /* fp16 shadow of p refreshed in the same pass, or NULL */
    uint64_t n;
    float lr, beta1, beta2, eps;
    const float *scaler_state;       /* float32[4] {scale, growth_tracker, found_inf, good_steps} (cnerf_scaler_*) */
    float extra_inv;                 /* extra factor on the un-scaling (1 / world_size) */
    int zero_grad;                   /* clear g after the update */
} CnerfGridAdam;
int cnerf_grid_backward_adam(const CnerfGridAdam *cfg);
int cnerf_grid_backward_adam_consumed(int *yes);

/* *id = the capture sequence id of `stream` while it is capturing into a hipGraph, 0 otherwise (host-side query, no launch): what a cache of
 * derived device data — e.g. the packed weights below — needs to know that a refresh it issues now is RECORDED, not executed. */
int cnerf_stream_capture_id(void *stream, uint64_t *id);
int cnerf_field_weight_image_bytes(uint32_t enc_dim, uint32_t n_hidden_geo, uint32_t n_rgb_out, uint64_t *bytes);
int cnerf_field_pack_weights(uint32_t enc_dim, uint32_t n_hidden_geo, uint32_t n_rgb_out, const float *params_net, const float *params_den,
                             const float *params_rgb, void *image, uint64_t image_bytes, void *stream);
int cnerf_field_forward_img(const void *enc, const float *xyz, const float *dirs, uint32_t dir_group, uint32_t P, uint32_t enc_dim,
                            uint32_t n_hidden_geo, uint32_t n_rgb_out, const float *params_net, const float *params_den,
                            const float *params_rgb, float *sigma, float *rgbc, int dtype, uint32_t enc_level_stride,
                            const void *weight_image, void *stream);
int cnerf_field_backward_img(const void *enc, const float *xyz, const float *dirs, uint32_t dir_group, uint32_t P, uint32_t enc_dim,
                             uint32_t n_hidden_geo, uint32_t n_rgb_out, const float *params_net, const float *params_den,
                             const float *params_rgb, const float *grad_sigma, const float *grad_rgbc, void *grad_enc,
                             float *grad_params_net, float *grad_params_den, float *grad_params_rgb, void *workspace,
                             uint64_t workspace_bytes, int dtype, const uint8_t *tile_live, const void *weight_image, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Generic fully fused MLP = tinycudann.Network(n_in, n_out, {FullyFusedMLP, ReLU, 64 neurons, 1|2 hidden layers})
 * (reference call sites: nerf/network_grid.py:18-54 RGB_network; :98-139 when the fused field is not used).
 * x [P, ldx] and y [P, ldy] row-major in `dtype` (leading dimensions in elements: strided views are accepted);
 * params float32 flat: row-major [64, pad16(n_in)], ([64,64]), [pad16(n_out), 64]; bias-free; output_activation 0 none | 1 sigmoid.
 * n_in <= 128, n_out <= 64, n_neurons == 64.  Numerics as cnerf_field_forward (CNERF_F16 = tcnn's, CNERF_F32 exact).
 * ---------------------------------------------------------------------------------------------- */
int cnerf_mlp_forward(const void *x, uint32_t ldx, const float *params, uint32_t P, uint32_t n_in, uint32_t n_out,
                      uint32_t n_neurons, uint32_t n_hidden_layers, int output_activation, void *y, uint32_t ldy, int dtype,
                      void *stream);
/* Backward (forward recomputed): grad_y [P, ldgy] (dtype) in; grad_x [P, ldgx] (dtype) out, NULL to skip;
 * grad_params float32, ACCUMULATED (caller pre-zeroes).  workspace: 16-byte aligned, cnerf_mlp_backward_workspace_bytes(). */
int cnerf_mlp_backward(const void *x, uint32_t ldx, const float *params, const void *grad_y, uint32_t ldgy, uint32_t P,
                       uint32_t n_in, uint32_t n_out, uint32_t n_neurons, uint32_t n_hidden_layers, int output_activation,
                       void *grad_x, uint32_t ldgx, float *grad_params, void *workspace, uint64_t workspace_bytes, int dtype,
                       void *stream);
int cnerf_mlp_backward_workspace_bytes(uint32_t P, uint32_t n_in, uint32_t n_out, uint32_t n_neurons,
                                       uint32_t n_hidden_layers, int dtype, uint64_t *bytes);

/* ------------------------------------------------------------------------------------------------
 * Ray generation (reference: nerf/provider.py:402-464 pinhole branch; nerf/provider_utils.py:239-302 get_rays)
 * c2w [V,3,4] row-major; outputs origins, directions [V, H, W, 3].
 * convention 0 = nerfstudio/OpenGL (provider.py: dir = normalize(R [ (x+.5-cx)/fx, -(y+.5-cy)/fy, -1 ]),
 *                 x = linspace(0, W*level-1, W), y likewise);
 * convention 1 = torch-ngp get_rays (dir = R normalize([ (i+.5-cx)/fx, (j+.5-cy)/fy, 1 ])), level ignored.
 * ---------------------------------------------------------------------------------------------- */
int cnerf_generate_rays(const float *c2w, uint32_t V, uint32_t H, uint32_t W, float fx, float fy, float cx, float cy,
                        float level, int convention, float *origins, float *directions, void *stream);
/* The OPENCV_FISHEYE branch of NerfstudioData._generate_rays (nerf/provider.py:421-433): the nerfstudio pixel grid of convention 0, each
 * normalised coordinate un-distorted by radial_and_tangential_undistort (nerf/provider_utils.py:197-234: ten Newton steps on the residual /
 * Jacobian of :128-194, eps 1e-3), theta = clip(|coord|, 0, pi), dir = normalize(R [x sin(theta)/theta, y sin(theta)/theta, -cos(theta)]).
 * distortion_host: [k1, k2, k3, k4, p1, p2] (host array, as the reference's `distortion_params`, provider.py:359). */
int cnerf_generate_rays_fisheye(const float *c2w, uint32_t V, uint32_t H, uint32_t W, float fx, float fy, float cx, float cy,
                                float level, const float *distortion_host, float *origins, float *directions, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Pure-PyTorch renderer `NeRFRenderer.run` (nerf/renderer.py:278-405) as fused kernels.
 * ---------------------------------------------------------------------------------------------- */
/* sample_pdf — nerf/renderer.py:21-55 for arbitrary bins / weights (run() itself uses the fused cnerf_sample_fine_merge): bins [B, n_bins],
 * weights [B, n_bins - 1] -> samples [B, n_samples]; pdf = (w + 1e-5) / sum, cdf = [0, cumsum(pdf)], searchsorted(right=True), lerp with
 * denom < 1e-5 -> 1.  u [B, n_samples] replays the torch.rand draw of :37; NULL = det (the midpoint linspace of :33-35).  n_bins <= 256. */
int cnerf_sample_pdf(const float *bins, const float *weights, const float *u, uint32_t B, uint32_t n_bins, uint32_t n_samples,
                     float *samples, void *stream);
/* Stratified sampling: z = near + (far-near) * linspace(0,1,T)[i] + (noise-0.5)*sample_dist (noise NULL = no
 * perturbation), xyz = clip(o + d z, aabb)  (renderer.py:310-322).  z_vals [N,T], xyzs [N,T,3]. */
int cnerf_sample_coarse(const float *rays_o, const float *rays_d, const float *nears, const float *fars, const float *aabb,
                        const float *noise, uint32_t N, uint32_t T, float *z_vals, float *xyzs, void *stream);
/* Importance resampling + merge (renderer.py:334-363 + sample_pdf :21-55): from coarse z [N,T] and sigma [N,T]
 * computes weights, inverts the CDF at u [N,t] (u NULL = deterministic midpoints, sample_pdf:34), and writes the
 * merged, sorted z_all [N,T+t] and xyz_all [N,T+t,3].  T, t <= 128. */
int cnerf_sample_fine_merge(const float *rays_o, const float *rays_d, const float *nears, const float *fars,
                            const float *aabb, const float *z_vals, const float *sigmas, const float *u, uint32_t N,
                            uint32_t T, uint32_t t, float *z_all, float *xyz_all, void *stream);
/* Split form of the same step: the new samples are NOT merged into the coarse ones in memory.  z_all [N,T+t] is the sorted merge as
 * above; xyz_fine [N,t,3] receives the new samples in their own block; src_index [N,T+t] (uint32) maps every sorted position to its
 * row in the sample list [coarse block N*T rows | fine block N*t rows]: coarse i of ray n -> n*T + i, fine m (draw order) -> N*T + n*t + m.
 * The field is then evaluated on that list (the coarse block's grid features are already there from the density pass) and the
 * compositing entries below read through src_index.  xyz_all may be NULL here (or non-NULL to get both forms). */
int cnerf_sample_fine_merge_split(const float *rays_o, const float *rays_d, const float *nears, const float *fars,
                                  const float *aabb, const float *z_vals, const float *sigmas, const float *u, uint32_t N,
                                  uint32_t T, uint32_t t, float *z_all, float *xyz_all, float *xyz_fine, uint32_t *src_index,
                                  void *stream);
/* The same two samplers also writing the grid's [0,1] coordinates of their points, unit = (xyz + bound) / (2 bound) — the map
 * GridEncoder.forward applies first (gridencoder/grid.py:156) — so that the gather needs no elementwise pass in between.
 * unit [N,T,3] / unit_fine [N,t,3] float32, same arithmetic as the torch expression on a GPU (float add, then a multiplication by the
 * float reciprocal of 2*bound: that is how torch divides a tensor by a host scalar). */
int cnerf_sample_coarse_unit(const float *rays_o, const float *rays_d, const float *nears, const float *fars, const float *aabb,
                             const float *noise, uint32_t N, uint32_t T, float *z_vals, float *xyzs, float *unit, float bound,
                             void *stream);
/* cnerf_sample_coarse_unit with cnerf_near_far_from_aabb folded in (renderer.py:297 then :309-322: one launch instead of two): nears / fars
 * [N] are OUTPUTS here, computed per ray against `aabb` with the arithmetic of raymarching.cu:91-145 (bit-identical to the separate call). */
int cnerf_sample_coarse_unit_aabb(const float *rays_o, const float *rays_d, const float *aabb, float min_near, const float *noise, uint32_t N,
                                  uint32_t T, float *nears, float *fars, float *z_vals, float *xyzs, float *unit, float bound, void *stream);
int cnerf_sample_fine_merge_split_unit(const float *rays_o, const float *rays_d, const float *nears, const float *fars,
                                       const float *aabb, const float *z_vals, const float *sigmas, const float *u, uint32_t N,
                                       uint32_t T, uint32_t t, float *z_all, float *xyz_fine, uint32_t *src_index,
                                       float *unit_fine, float bound, void *stream);
/* weights_sum_i x3 (renderer.py:384-402, 407-474) in one pass over [N,S] samples: all / fg (sigma*edit_mask) /
 * bg (sigma*(1-edit_mask)) composites.  soft_mask: edit = sigmoid((conf-thr)*100) else conf>0.5.
 * rgbc [N,S,4] (rgb + confidence), sigmas [N,S], z_vals [N,S].
 * out_ray [3][N][6]  = per composite (all,fg,bg): image rgb, depth, weights_sum, render_mask;
 * out_weights [3][N][S] (NULL to skip). */
int cnerf_composite_run(const float *sigmas, const float *rgbc, const float *z_vals, const float *nears, const float *fars,
                        uint32_t N, uint32_t S, uint32_t num_steps, int soft_mask, float conf_thr, float *out_ray,
                        float *out_weights, void *stream);
/* Backward of cnerf_composite_run: grad_out_ray [3][N][6] -> grad_sigmas [N,S], grad_rgbc [N,S,4]
 * (depth and weights_sum gradients are honoured; detach_* flags follow renderer.py:409-418, 460-463). */
int cnerf_composite_run_backward(const float *grad_out_ray, const float *sigmas, const float *rgbc, const float *z_vals,
                                 const float *nears, const float *fars, uint32_t N, uint32_t S, uint32_t num_steps,
                                 int soft_mask, float conf_thr, int detach_bg, int detach_mask_from_field,
                                 float *grad_sigmas, float *grad_rgbc, void *stream);
/* Indexed forms: sample n,i (sorted position) lives at row src_index[n*S+i] of sigmas / rgbc / grad_* (NULL = row n*S+i, i.e. the
 * plain forms).  sigma_sorted [N,S] / rgbc_sorted [N,S,4] (may be NULL) receive the per-sample inputs in sorted order — the
 * `sigma` / `rgbs` entries of run()'s result dict (renderer.py:396-398). */
int cnerf_composite_run_indexed(const float *sigmas, const float *rgbc, const float *z_vals, const float *nears, const float *fars,
                                uint32_t N, uint32_t S, uint32_t num_steps, int soft_mask, float conf_thr,
                                const uint32_t *src_index, float *out_ray, float *out_weights, float *sigma_sorted,
                                float *rgbc_sorted, void *stream);
/* cnerf_composite_run_indexed computing only the variants whose bit is set in variant_mask (bit 0 all, 1 edit region, 2 background; 7 = all three):
 * the rows of out_ray (and of out_weights, when given) that belong to the others are written as zeros.  The reconstruction stage reads the first
 * composite only (utils_init_nerf.py:214-260 `train_step`); the editing stage all three. */
int cnerf_composite_run_indexed_variants(const float *sigmas, const float *rgbc, const float *z_vals, const float *nears, const float *fars,
                                         uint32_t N, uint32_t S, uint32_t num_steps, int soft_mask, float conf_thr, const uint32_t *src_index,
                                         float *out_ray, float *out_weights, float *sigma_sorted, float *rgbc_sorted, uint32_t variant_mask,
                                         void *stream);
int cnerf_composite_run_backward_indexed(const float *grad_out_ray, const float *sigmas, const float *rgbc, const float *z_vals,
                                         const float *nears, const float *fars, uint32_t N, uint32_t S, uint32_t num_steps,
                                         int soft_mask, float conf_thr, int detach_bg, int detach_mask_from_field,
                                         const uint32_t *src_index, float *grad_sigmas, float *grad_rgbc, void *stream);
/* The same with early termination for a half-precision consumer (round 5; the reference's own early-out is `T < T_thresh` in
 * raymarching.cu:691-772, its run() path has none).  flush_half_zero: a sample whose gradients AS THE FUSED FIELD BACKWARD CONSUMES THEM —
 * half(g_sigma * exp'(raw)), half(g_c * sigmoid'(c)) — are all zero (|g_sigma| sigma < 2^-26 and |g_c| c (1 - c) < 2^-26: behind an opaque
 * surface, in empty space) gets exact zeros in grad_sigmas / grad_rgbc: every parameter gradient computed downstream is bit-identical, and the
 * field backward / grid scatter can skip the row.  tile_live (optional, uint8 [N * S / 32]; needs src_index and num_steps, S - num_steps
 * multiples of 32): byte k = 1 when rows 32 k .. 32 k + 31 of the sample list hold at least one live row (cnerf_field_backward_ex). */
int cnerf_composite_run_backward_indexed_flush(const float *grad_out_ray, const float *sigmas, const float *rgbc, const float *z_vals,
                                               const float *nears, const float *fars, uint32_t N, uint32_t S, uint32_t num_steps,
                                               int soft_mask, float conf_thr, int detach_bg, int detach_mask_from_field,
                                               const uint32_t *src_index, float *grad_sigmas, float *grad_rgbc, int flush_half_zero,
                                               uint8_t *tile_live, void *stream);

/* Loss of the reconstruction step (Trainer_Nerf.train_step_pretrain, utils_init_nerf.py:220-234) with its gradient, one launch:
 *   loss = w_rgb * mean((image - rgb_gt)^2) + w_conf * mean((render_mask - mask_gt)^2)
 * on the `all` composite of out_ray [3][N][6] (cnerf_composite_run); rgb_gt [N,3], mask_gt [N] (NULL: no mask term).
 * loss float32[65]: loss[0] = the loss, loss[1..64] = scratch (per-workgroup partial sums, added in a fixed order);
 * grad_out_ray [3][N][6] = d(loss)/d(out_ray), ready for cnerf_composite_run_backward. */
int cnerf_recon_loss(const float *out_ray, const float *rgb_gt, const float *mask_gt, uint32_t N, float w_rgb, float w_conf,
                     float *loss, float *grad_out_ray, void *stream);
/* the same with grad_out_ray multiplied by the device scalar grad_scale[0] — the seed the backward pass starts from under a loss scaler
 * (GradScaler.scale(loss).backward()): the element-wise multiply of the autograd node disappears.  grad_scale NULL = plain. */
int cnerf_recon_loss_scaled(const float *out_ray, const float *rgb_gt, const float *mask_gt, uint32_t N, float w_rgb, float w_conf,
                            const float *grad_scale, float *loss, float *grad_out_ray, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Optimiser step used by the reference's recipe (main.py:182: Adam betas (0.9,0.99) eps 1e-15, no weight decay),
 * fused with gradient un-scaling, the fp16 shadow-table refresh and gradient zeroing.
 * p, m, v float32 [n]; g float32 [n] (zeroed after use if zero_grad); p_half (may be NULL) float16 [n].
 * step = 1-based step index (bias correction on the host); g is multiplied by grad_scale_inv first.
 * ---------------------------------------------------------------------------------------------- */
int cnerf_adam_step(float *p, float *g, float *m, float *v, void *p_half, uint64_t n, float lr, float beta1, float beta2,
                    float eps, uint32_t step, float grad_scale_inv, int zero_grad, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Dynamic loss scaling with torch.cuda.amp.GradScaler's semantics (the reference trainer's fp16 recipe:
 * `self.scaler = GradScaler(enabled=fp16)`, scaler.scale(loss).backward(); scaler.step(optimizer); scaler.update()),
 * kept on the device so that a training step has no host read:
 *   state = float32[4] {scale, growth_tracker, found_inf, good_steps}  (initialise to {65536, 0, 0, 0}).
 * cnerf_scaler_check     : found_inf |= any non-finite value in g[0..n)  (call on the all-reduced gradients).
 * cnerf_adam_step_scaled : cnerf_adam_step with g multiplied by extra_inv / scale, the 1-based step = good_steps + 1 (bias
 *                          correction in the kernel) and the whole update skipped when found_inf is set (gradients are still
 *                          zeroed if zero_grad) — what scaler.step() does.
 * cnerf_scaler_update    : found_inf ? {scale *= backoff, tracker = 0} : {good_steps += 1, tracker += 1, tracker == interval ?
 *                          {scale *= growth, tracker = 0}}; found_inf = 0.  Call once per step after every adam_step_scaled.
 * ---------------------------------------------------------------------------------------------- */
int cnerf_scaler_check(const float *g, uint64_t n, float *state, void *stream);
/* Round 6: while a scaler state is WATCHED (state != NULL; NULL switches it off — the default), the two gradient-producing entry points of the
 * field raise its found_inf themselves on their launch stream: cnerf_field_backward* when a parameter gradient it writes is not finite,
 * cnerf_grid_encode_backward* when an incoming feature gradient is not finite (the value it then also poisons the table gradient with).  A caller
 * whose gradients ALL come from these entry points may skip cnerf_scaler_check (a pass over every gradient: 9 us for the benchmark table, 30 us
 * for the reference field's).  One watched state per process; host-side switch, no launch. */
int cnerf_scaler_watch(float *state);
int cnerf_adam_step_scaled(float *p, float *g, float *m, float *v, void *p_half, uint64_t n, float lr, float beta1, float beta2,
                           float eps, const float *state, float extra_inv, int zero_grad, void *stream);
int cnerf_scaler_update(float *state, float growth_factor, float backoff_factor, uint32_t growth_interval, void *stream);
/* cnerf_adam_step_scaled for up to CNERF_ADAM_MAX_JOBS SMALL tensors in one single-workgroup launch (the reference's three tinycudann
 * parameter vectors: main.py:182 gives each network its own Adam group) and, with update_scaler != 0, cnerf_scaler_update folded into
 * its tail — call it as the LAST adam step of the iteration.  Plain pointers, no alignment demands. */
#define CNERF_ADAM_MAX_JOBS 8
typedef struct CnerfAdamJobs {
    float *p[CNERF_ADAM_MAX_JOBS], *g[CNERF_ADAM_MAX_JOBS], *m[CNERF_ADAM_MAX_JOBS], *v[CNERF_ADAM_MAX_JOBS];
    void *p_half[CNERF_ADAM_MAX_JOBS];            /* optional fp16 shadows (NULL) */
    uint64_t n[CNERF_ADAM_MAX_JOBS];
    float lr[CNERF_ADAM_MAX_JOBS];
    uint32_t n_jobs;
} CnerfAdamJobs;
int cnerf_adam_step_scaled_multi(const CnerfAdamJobs *jobs, float beta1, float beta2, float eps, float *state, float extra_inv, int zero_grad,
                                 int update_scaler, float growth_factor, float backoff_factor, uint32_t growth_interval, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Data-parallel gradient exchange (customnerf_amd/dp.py; the reference has no working multi-GPU path: its DDP scaffolding
 * is dead code, nerf/utils_init_nerf.py:76-78, 709-726 — SURVEY.md §8e).  The collectives themselves are RCCL's; these are
 * the two passes either side of the all-to-all.
 * cnerf_dp_pack   : payload[i] = half(grad[i] * scale), grad[i] = 0   (grad 16-byte aligned, payload 8-byte aligned)
 * cnerf_dp_reduce : out[i] = sum_r float(recv[r * shard + i]), r < world, accumulated in float32; shard % 64 == 0;
 *                   scaler_state (nullable): found_inf |= any non-finite sum — cnerf_scaler_check folded in.
 * ---------------------------------------------------------------------------------------------- */
int cnerf_dp_pack(float *grad, void *payload_half, uint64_t n, float scale, void *stream);
int cnerf_dp_reduce(const void *recv_half, uint32_t world, uint64_t shard, float *out, float *scaler_state, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CUSTOMNERF_HIP_H */
