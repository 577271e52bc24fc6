/*
 * customnerf_sd.h — C-ABI of the score-distillation (SDS) half of the hot path in libcustomnerf_hip.so (MI355X / gfx950).
 *
 * The reference reaches this arithmetic through `diffusers` (third-party, un-vendored, unpinned: requirements.txt:24):
 *   nerf/sd.py:97-105   StableDiffusion.encode_imgs   -> AutoencoderKL.encode     (VAE encoder, forward AND backward)
 *   nerf/sd.py:115-155  StableDiffusion.train_step    -> UNet2DConditionModel     (eps-prediction, forward only)
 * Both networks are sequences of five primitive shapes — (implicit-)GEMM, GroupNorm, LayerNorm, row softmax, element-wise —
 * and that is the boundary drawn here: one entry point per primitive, activations in NHWC / [tokens, channels] float16,
 * float32 accumulation on the matrix cores (v_mfma_f32_32x32x16_f16).  The network graphs (which primitive, which weights,
 * in which order) live above the boundary in customnerf_amd/sd/ and mirror diffusers' module tree key for key.
 *
 * Conventions: as customnerf_hip.h (device pointers, caller-owned buffers, no allocation, no host sync, `stream`, int
 * status).  `half` below means IEEE binary16 stored as uint16_t-sized elements.
 */
#ifndef CUSTOMNERF_SD_H
#define CUSTOMNERF_SD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------
 * cnerf_sd_gemm:  C[z][m][n] = epilogue( alpha * sum_k A(z,m,k) * B[z][n][k] )
 * B is always "weights-like": row n holds its K values contiguously (ldb elements between rows).
 * A is either a dense row-major matrix (mode 0: row m at A + m*lda, K contiguous) or an implicit im2col view of an NHWC
 * activation (mode 1: m = (image, oh, ow), k = (kh, kw, c)):
 *      num_h = oh*stride + kh - pad_t ;  num_h must be >= 0, and (tstride == 2) even, then num_h /= tstride ;
 *      ih = num_h / ups  (ups = 2: the conv reads a nearest-neighbour 2x upsampled input without materialising it) ;
 *      valid iff ih < H_in (same for w); invalid taps read 0.   A(m,k) = in[image][ih][iw][c], c < Cin, Cin % 8 == 0.
 *   forward conv (torch.nn.Conv2d):            stride s, pad p, tstride 1
 *   input-gradient of a conv (frozen weights):  stride 1, pad KH-1-p, tstride s, B = weights flipped and transposed
 * epilogue, in this order:  (+ bias[n])  (+ bias_rows[(m / rows_per_bias_row)][n])  (activation: 0 none, 1 SiLU, 2 GELU(erf), 3 quick-GELU x*sigmoid(1.702x),
 *                           4 GEGLU on interleaved column pairs: C[m][j] = y[2j] * gelu(y[2j+1]), C is N/2 wide, no residual)
 *                           (+ residual[z][m][n])  -> C (half), and/or C32 (float, optional).
 * batching: z = zo * batch_inner + zi, operand bases advance by s?_o * zo + s?_i * zi elements (attention heads are a strided
 * view of [batch, tokens, heads*dim]); batch = 1 and zero strides for plain GEMMs / convs.
 * workspace: split-K partial sums (float).  cnerf_sd_gemm_workspace_bytes() gives the size the library's own split heuristic
 * needs for this problem; a NULL / too small workspace simply disables split-K.
 * Requirements: K % 8 == 0, lda/ldb % 8 == 0, all bases 16-byte aligned.  M, N arbitrary.
 * ---------------------------------------------------------------------------------------------- */
typedef struct CnerfSdGemm {
    const void *A;            /* half */
    const void *B;            /* half */
    void *C;                  /* half, may be NULL when C32 is set */
    float *C32;               /* float, optional */
    const float *bias;        /* [N] or NULL */
    const float *bias_rows;   /* [ceil(M / rows_per_bias_row)][ld_bias_rows] or NULL */
    const void *residual;     /* half [z][M][ldr] or NULL */
    uint32_t M, N, K;
    uint32_t lda, ldb, ldc, ldr;
    uint32_t rows_per_bias_row;
    uint32_t ld_bias_rows;    /* floats between consecutive rows of bias_rows (0 = N) */
    float alpha;
    int32_t act;
    uint32_t batch_outer, batch_inner;
    uint64_t sa_o, sa_i, sb_o, sb_i, sc_o, sc_i;    /* residual uses the C strides */
    int32_t mode;             /* 0 dense, 1 implicit conv */
    uint32_t Cin, H_in, W_in, H_out, W_out, KH, KW, stride, pad_t, pad_l, ups, tstride;
    /* optional GroupNorm statistics of the OUTPUT (the norm that consumes C next): gn_sums [M / gn_rows][gn_groups][2] int64 fixed point
     * (CNERF_SD_GN_FRAC_BITS fractional bits), pre-zeroed, receives sum and sum of squares of the half-rounded outputs per (image,
     * channel group); N % gn_groups == 0, gn_rows >= 64.
     * Served on every schedule (round 6: the split-K tail kernel accumulates them; a request of more than 512 (image, group) pairs keeps
     * the problem off the split-K path). */
    int64_t *gn_sums;
    uint32_t gn_groups, gn_rows;
    /* optional LayerNorm of the OUTPUT rows (ABI 4; the transformer blocks normalise the result of proj_in / attn.to_out right away):
     * ln_out [M, N] half (row pitch N) receives (row - mean) * rstd * ln_gamma + ln_beta of the half-rounded row of C, biased variance, ln_eps —
     * exactly cnerf_sd_layernorm_forward on C.  Served only by the split-K tail kernel (cnerf_sd_gemm_serves_ln tells): a one-launch GEMM
     * leaves ln_out untouched and the caller runs cnerf_sd_layernorm_forward. */
    void *ln_out;
    const float *ln_gamma, *ln_beta;
    float ln_eps;
} CnerfSdGemm;

int cnerf_sd_gemm(const CnerfSdGemm *desc, void *workspace, uint64_t workspace_bytes, void *stream);
int cnerf_sd_gemm_workspace_bytes(const CnerfSdGemm *desc, uint64_t *bytes);
/* *yes = 1 when cnerf_sd_gemm (given the workspace cnerf_sd_gemm_workspace_bytes asks for) will write desc->ln_out: split-K schedule, dense half
 * output with ldc == N, N % 8 == 0, N <= 2048, no GEGLU, no GroupNorm-statistics request. */
int cnerf_sd_gemm_serves_ln(const CnerfSdGemm *desc, int *yes);

/* ------------------------------------------------------------------------------------------------
 * GroupNorm over NHWC half activations x [B, HW, C] (torch.nn.GroupNorm(G, C, eps) semantics, biased variance, statistics in
 * float32), optionally followed by SiLU (the `norm -> nonlinearity` pair of every diffusers ResnetBlock2D).
 *   stats:   sums [B][G][2] int64 (sum, sum of squares) in FIXED POINT with CNERF_SD_GN_FRAC_BITS fractional bits: every partial sum is
 *            rounded to that grid once and added with integer atomics, so the statistics — and everything computed from them — do
 *            not depend on the order in which workgroups arrive (bit-reproducible).  Filled by the call; zero_sums != 0: the call
 *            zeroes them first,
 *            zero_sums == 0: the caller passes zeros (one fill for all the norms of a network instead of one launch each);
 *            zero_sums == 2: `sums` already hold the statistics (filled by the producing cnerf_sd_gemm's gn_sums): no statistics pass
 *   forward: y = act((x - mean) * rstd * gamma[c] + beta[c])
 *   backward (frozen gamma/beta): dx from dy, recomputing the forward; `sums` are the forward's; scratch [B][G][2] int64 (same format).
 * ---------------------------------------------------------------------------------------------- */
#define CNERF_SD_GN_FRAC_BITS 20
int cnerf_sd_groupnorm_forward(const void *x, const float *gamma, const float *beta, uint32_t B, uint32_t HW, uint32_t C,
                               uint32_t G, float eps, int silu, int64_t *sums, int zero_sums, void *y, void *stream);
int cnerf_sd_groupnorm_backward(const void *x, const void *dy, const float *gamma, const float *beta, uint32_t B, uint32_t HW,
                                uint32_t C, uint32_t G, float eps, int silu, const int64_t *sums, int64_t *scratch, void *dx,
                                void *stream);
/* The same with two savings for residual blocks (round 4): scratch_is_zero != 0 — the caller hands over zeros (one fill for all the norms of a
 * backward pass, as zero_sums == 0 does for the forward); residual (nullable, x's shape) — a second gradient arriving at x (the block's skip
 * connection) is added before the one rounding to half: dx = groupnorm_backward(dy) + residual, no separate add launch. */
int cnerf_sd_groupnorm_backward_ex(const void *x, const void *dy, const float *gamma, const float *beta, uint32_t B, uint32_t HW,
                                   uint32_t C, uint32_t G, float eps, int silu, const int64_t *sums, int64_t *scratch,
                                   int scratch_is_zero, const void *residual, void *dx, void *stream);

/* LayerNorm over the last dimension of x [rows, C] (half), float32 statistics, eps inside the sqrt. */
int cnerf_sd_layernorm_forward(const void *x, const float *gamma, const float *beta, uint32_t rows, uint32_t C, float eps,
                               void *y, void *stream);

/* Row softmax of S [rows, ld] (half) in place over the first `cols` columns, float32 arithmetic; columns cols..ld-1 are set
 * to zero (so the matrix can be used directly as a K-padded GEMM operand).
 * backward: dS = P * (dP - rowsum(dP * P)), written over dP. */
int cnerf_sd_softmax_forward(void *S, uint64_t rows, uint32_t cols, uint32_t ld, void *stream);
int cnerf_sd_softmax_backward(const void *P, void *dP, uint64_t rows, uint32_t cols, uint32_t ld, void *stream);

/* Fused attention forward (no materialised scores): out[b][q][h*d + :] = softmax(q_h k_h^T / sqrt(d)) v_h for q [B][Tq][.] (row stride
 * ldq, batch stride sq, head h at column h*d), k likewise, vT [B][H*d][ldv] = V transposed (cnerf_sd_transpose; columns Tk..ldv-1
 * must be finite, ldv >= Tk rounded up to 32), out row stride ldo.  d % 8 == 0, d <= 160 (the UNet's 40 / 80 / 160, CLIP's 64).
 * causal != 0: query i attends to keys <= i (CLIP text encoder). */
int cnerf_sd_attention(const void *q, const void *k, const void *vT, void *out, uint32_t B, uint32_t H, uint32_t Tq, uint32_t Tk,
                       uint32_t d, uint32_t ldq, uint64_t sq, uint32_t ldk, uint64_t sk, uint32_t ldv, uint64_t sv, uint32_t ldo,
                       uint64_t so, int causal, void *stream);
/* The same with V as the value projection leaves it — v [B][Tk][.] (row pitch ldv, batch stride sv, head h at column h*d; 16-byte
 * aligned rows, ldv % 8 == 0): the kernel transposes its key tiles in LDS (ds_read_b64_tr_b16), no cnerf_sd_transpose launch. */
int cnerf_sd_attention_v(const void *q, const void *k, const void *v, void *out, uint32_t B, uint32_t H, uint32_t Tq, uint32_t Tk,
                         uint32_t d, uint32_t ldq, uint64_t sq, uint32_t ldk, uint64_t sk, uint32_t ldv, uint64_t sv, uint32_t ldo,
                         uint64_t so, int causal, void *stream);

/* GEGLU (diffusers GEGLU): y[r][c] = x[r][c] * gelu_erf(x[r][c + C]) for x [rows, 2C] -> y [rows, C] (half). */
int cnerf_sd_geglu(const void *x, uint64_t rows, uint32_t C, void *y, void *stream);

/* Batched 2-D transpose of half matrices: dst[z][c][r] = src[z][r][c]; src row stride lds, dst row stride ldd; z advances the
 * bases by ss / sd elements.  Pad columns of dst (r in rows..ldd-1) are zero-filled. */
int cnerf_sd_transpose(const void *src, void *dst, uint32_t rows, uint32_t cols, uint32_t lds, uint32_t ldd, uint32_t batch,
                       uint64_t ss, uint64_t sd, void *stream);

/* Image front-end of encode_imgs (utils_init_nerf.py:303 + sd.py:100): bilinear resize (align_corners=False) of img
 * [B,3,Hi,Wi] float32 NCHW in [0,1] to Ho x Wo, then 2x-1, written as NHWC half with channels padded to 8 (zeros).
 * backward: d(img) float32 [B,3,Hi,Wi] from d(out) [B,Ho,Wo,8] half (overwrites d_img). */
int cnerf_sd_image_to_vae_input(const float *img, uint32_t B, uint32_t Hi, uint32_t Wi, uint32_t Ho, uint32_t Wo, void *out,
                                void *stream);
int cnerf_sd_image_to_vae_input_backward(const void *d_out, uint32_t B, uint32_t Hi, uint32_t Wi, uint32_t Ho, uint32_t Wo,
                                         float *d_img, void *stream);

/* CLIP image front-end of the view classifier (nerf/clip.py:13-17 `transformCLIP`, used at utils_init_nerf.py:256):
 * torchvision Resize(S, BICUBIC, antialias=None) -- smaller edge to S, aspect kept, torch's bicubic (A = -0.75, align_corners=False,
 * no antialias filter) -- then CenterCrop(S) and Normalize(mean, std).  img [B,3,Hi,Wi] float32 -> out [B,3,S,S] float32.
 * mean3 / std3 are HOST pointers to 3 floats.  cnerf_sd_patchify: x [B,3,S,S] float32 -> patch rows [B,(S/P)^2, P*P*3] half in
 * (kh, kw, c) order, the A operand of the ViT patch-embedding convolution (`visual.conv1`, kernel = stride = P) as a dense GEMM. */
int cnerf_sd_clip_preprocess(const float *img, uint32_t B, uint32_t Hi, uint32_t Wi, uint32_t S, const float *mean3, const float *std3,
                             float *out, void *stream);
int cnerf_sd_patchify(const float *x, uint32_t B, uint32_t S, uint32_t P, void *out, void *stream);

/* Diffusers get_timestep_embedding(t, dim, flip_sin_to_cos=True, downscale_freq_shift=0): out[b][0:dim/2] = cos(t_b w_i),
 * out[b][dim/2:] = sin(t_b w_i), w_i = exp(-ln(10000) i / (dim/2)); half output [B, dim]. */
int cnerf_sd_timestep_embedding(const float *t, uint32_t B, uint32_t dim, void *out, void *stream);

/* Score-distillation tail of train_step (sd.py:133-148) on latents [n] float32:
 *   noisy[i]     = sqrt(ab) * latents[i] + sqrt(1 - ab) * noise[i]                      (scheduler.add_noise), half, twice (CFG pair)
 *   grad[i]      = nan_to_num( (1 - ab) * ((e_text + g (e_text - e_uncond)) - noise[i]) * lambda_sd )
 * cnerf_sd_add_noise writes the UNet input [2][n/4 pixels][8] (NHWC half, 4 latent channels padded to 8);
 * cnerf_sd_sds_grad reads the UNet output eps [2][pixels][ld_eps] half and writes grad float32 [n] in the latents' NCHW order. */
int cnerf_sd_add_noise(const float *latents, const float *noise, float alpha_bar, uint32_t pixels, void *unet_in, void *stream);
int cnerf_sd_sds_grad(const void *eps, uint32_t ld_eps, const float *noise, float alpha_bar, float guidance, float lambda_sd,
                      uint32_t pixels, float *grad, void *stream);

/* Generic element-wise helpers on half buffers (n elements): y = a + b ; y = silu(x). */
int cnerf_sd_add(const void *a, const void *b, uint64_t n, void *y, void *stream);
int cnerf_sd_silu(const void *x, uint64_t n, void *y, void *stream);
/* channel concat of NHWC half tensors: y[r][0:C1] = a[r], y[r][C1:C1+C2] = b[r] */
int cnerf_sd_concat(const void *a, const void *b, uint64_t rows, uint32_t C1, uint32_t C2, void *y, void *stream);
/* Same, also accumulating the GroupNorm statistics of y into gn_sums [rows / gn_rows][gn_groups][2] (int64 fixed point, pre-zeroed: the
 * gn_sums contract of CnerfSdGemm) — the UNet's up blocks normalise the concat first thing.  gn_sums NULL: plain concat.
 * (C1 + C2) / gn_groups >= 8, rows % gn_rows == 0, (rows / gn_rows) * gn_groups <= 512 — else CNERF_EINVAL. */
int cnerf_sd_concat_gn(const void *a, const void *b, uint64_t rows, uint32_t C1, uint32_t C2, void *y, int64_t *gn_sums, uint32_t gn_groups,
                       uint32_t gn_rows, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Glue of one editing step as single launches (round 5; each replaces a chain of framework element-wise ops).
 *
 * cnerf_edit_ray_images: out_ray [3][B*HW][6] float32 (cnerf_composite_run's all / fg / bg outputs; channels rgb, depth, weights_sum,
 *   mask) -> the three NCHW images [B][3][HW] that train_step_editing builds with reshape/permute/contiguous (utils_init_nerf.py:361-366).
 *   _backward: d(out_ray), written in full, from the three image gradients (any of them NULL = zeros; channels 3..5 get zeros).
 * cnerf_edit_l1_loss: loss[0] = scale * mean |a - b| (keep_bg * F.l1_loss, utils_init_nerf.py:389-391), dsign[i] = (scale / n) sgn(a_i - b_i)
 *   (= d loss / d a; d loss / d b = -dsign).  One workgroup, fixed order of additions.
 * cnerf_edit_sds_loss: sd.py:150-152 — d = latents - (latents - grad), loss[0] = 0.5 sum d^2, diff2 = 2 d (d loss / d latents = diff2 * 0.5,
 *   formed in that order like torch's mse backward: overflow to infinity included).
 * cnerf_edit_scale_by_scalar: dst = src * (scalar[0] * mult), scalar a DEVICE float (the upstream gradient of a loss).
 * cnerf_sd_sample_latents: encode_imgs' posterior sample (sd.py:102-104, diffusers DiagonalGaussianDistribution): moments [B][hw][8] half
 *   NHWC (mean | logvar) -> latents [B][4][hw] float32 = (mean + exp(0.5 clamp(logvar, -30, 20)) noise) * scaling_factor;
 *   _backward: d(moments) [B][hw][8] half from d(latents).
 * cnerf_set_floats: n <= 16 floats from HOST memory into a device buffer by kernel argument (the UNet graph's timestep input). */
int cnerf_edit_ray_images(const float *out_ray, uint32_t B, uint32_t HW, float *img_all, float *img_fg, float *img_bg, void *stream);
int cnerf_edit_ray_images_backward(const float *d_all, const float *d_fg, const float *d_bg, uint32_t B, uint32_t HW, float *d_out_ray,
                                   void *stream);
int cnerf_edit_l1_loss(const float *a, const float *b, uint32_t n, float scale, float *loss, float *dsign, void *stream);
int cnerf_edit_sds_loss(const float *latents, const float *grad, uint32_t n, float *loss, float *diff2, void *stream);
int cnerf_edit_scale_by_scalar(const float *src, const float *scalar, float mult, uint32_t n, float *dst, void *stream);
int cnerf_sd_sample_latents(const void *moments, const float *noise, uint32_t B, uint32_t hw, float scaling_factor, float *latents,
                            void *stream);
int cnerf_sd_sample_latents_backward(const void *moments, const float *noise, const float *d_latents, uint32_t B, uint32_t hw,
                                     float scaling_factor, void *d_moments, void *stream);
int cnerf_set_floats(float *dst, const float *host_values, uint32_t n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CUSTOMNERF_SD_H */
