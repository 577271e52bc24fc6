// Fused field forward: NeRFNetwork.forward / .density (nerf/network_grid.py:159-193) in one persistent launch.
// See field_common.h for the MFMA data flow.  One wave = one 32-sample tile at a time; the weight fragments of all seven
// layers live in LDS (48 KiB fp16 / 96 KiB fp32) and are staged once per workgroup.
#include "field_common.h"

// ------------------------------------------------------------------------------------------------ weight staging
// Copy one layer's [rows, in_stride] float32 matrix into LDS in A-fragment order:
//   dst[((t * S + s) * 64 + lane) * J + j] = W[32 t + (lane & 31)][col(s, lane >> 5, j)]      (0 where out of range)
// KIND 0: natural column order (grid features).  KIND 1: C-register order (hidden activations).
// KIND 2: rgb layer 0 = [C-ordered fea at column 27.., then natural dir features at column 0..26].
template <bool H, int KIND>
__device__ __forceinline__ void fld_stage_layer(typename Prec<H>::elem_t *dst, const float *__restrict__ W, uint32_t rows, uint32_t in_stride,
                                                uint32_t T, uint32_t S, uint32_t n_valid_cols, uint32_t i0 = threadIdx.x, uint32_t istride = FLD_THREADS) {
    using P = Prec<H>;
    const uint32_t total = T * S * 64 * P::J;
    for (uint32_t i = i0; i < total; i += istride) {
        const uint32_t j = i % P::J, lane = (i / P::J) % 64, ts = i / (P::J * 64);
        const uint32_t s = ts % S, t = ts / S;
        const uint32_t row = 32 * t + (lane & 31), hi = lane >> 5;
        int col;
        if (KIND == 0) col = fld_col_natural<H>(s, hi, j);
        else if (KIND == 1) col = fld_col_clayout<H>(s, hi, j);
        else {
            const uint32_t s_fea = FLD_HID / P::KS;
            if (s < s_fea) col = FLD_NDIR + fld_col_clayout<H>(s, hi, j);
            else {
                col = fld_col_natural<H>(s - s_fea, hi, j);
                if (col >= FLD_NDIR) col = -1;
            }
        }
        float v = 0.0f;
        if (row < rows && col >= 0 && (uint32_t)col < n_valid_cols) v = W[(size_t)row * in_stride + col];
        dst[i] = (typename P::elem_t)v;
    }
}

template <bool H>
__device__ __forceinline__ void fld_stage_all(typename Prec<H>::elem_t *lds, const FieldDims &dm, const FieldLds &lo, const float *__restrict__ pnet,
                                              const float *__restrict__ pden, const float *__restrict__ prgb, bool with_rgb,
                                              uint32_t i0 = threadIdx.x, uint32_t istride = FLD_THREADS) {
    using P = Prec<H>;
    const uint32_t S64 = FLD_HID / P::KS, Senc = dm.enc_pad / P::KS;
    const float *n0 = pnet, *n1 = pnet + FLD_HID * dm.enc_pad;
    const float *n2 = n1 + (dm.n_hidden_geo == 2 ? FLD_HID * FLD_HID : 0);
    fld_stage_layer<H, 0>(lds + lo.off[0], n0, FLD_HID, dm.enc_pad, 2, Senc, dm.enc_pad, i0, istride);
    if (dm.n_hidden_geo == 2) fld_stage_layer<H, 1>(lds + lo.off[1], n1, FLD_HID, FLD_HID, 2, S64, FLD_HID, i0, istride);
    fld_stage_layer<H, 1>(lds + lo.off[2], n2, FLD_HID, FLD_HID, 2, S64, FLD_HID, i0, istride);
    fld_stage_layer<H, 1>(lds + lo.off[3], pden, FLD_HID, FLD_HID, 2, S64, FLD_HID, i0, istride);
    fld_stage_layer<H, 1>(lds + lo.off[4], pden + FLD_HID * FLD_HID, 16, FLD_HID, 1, S64, FLD_HID, i0, istride);
    if (with_rgb) {
        const uint32_t in_r0 = FLD_HID + FLD_DIR;       // 96 = pad16(27 + 64)
        fld_stage_layer<H, 2>(lds + lo.off[5], prgb, FLD_HID, in_r0, 2, in_r0 / P::KS, in_r0, i0, istride);
        fld_stage_layer<H, 1>(lds + lo.off[6], prgb + FLD_HID * in_r0, 16, FLD_HID, 1, S64, FLD_HID, i0, istride);
    }
}

// The staged image as a tensor in global memory (round 6): staging from the float32 parameters — a scattered 4-byte load, a division chain
// and a 2-byte LDS store per weight, 96 of them per thread — costs 14 us per launch (measured by staging twice: k_field_fwd 62.4 -> 76.5 us at
// 1 M samples and 24.3 -> 38.7 at 131 k; k_field_bwd_x2 449 -> 464 / 85 -> 97), three launches per step.  k_field_pack writes the fp16 image
// ONCE per parameter version (cnerf_field_pack_weights: the host keys it on the parameters' versions), and the kernels that are handed an
// image copy it into LDS in 16-byte chunks — same bits, same slots.
__global__ void __launch_bounds__(FLD_THREADS) k_field_pack(_Float16 *__restrict__ img, FieldDims dm, const float *__restrict__ pnet,
                                                            const float *__restrict__ pden, const float *__restrict__ prgb) {
    const FieldLds lo = fld_lds_layout<true>(dm);
    fld_stage_all<true>(img, dm, lo, pnet, pden, prgb, true, blockIdx.x * FLD_THREADS + threadIdx.x, gridDim.x * FLD_THREADS);
}
// image -> LDS, 16-byte chunks (n_halves is a multiple of 512: every layer is whole 64-lane fragments)
__device__ __forceinline__ void fld_copy_image(_Float16 *lds, const void *__restrict__ img, uint32_t n_halves) {
    const uint4 *src = reinterpret_cast<const uint4 *>(img);
    uint4 *dst = reinterpret_cast<uint4 *>(lds);
    for (uint32_t c = threadIdx.x; c < n_halves / 8; c += FLD_THREADS) dst[c] = src[c];
}

// ------------------------------------------------------------------------------------------------ per-wave building blocks
template <bool H>
__device__ __forceinline__ typename Prec<H>::frag_t fld_load_frag(const typename Prec<H>::elem_t *base, uint32_t t, uint32_t S, uint32_t s, uint32_t lane) {
    using P = Prec<H>;
    return *reinterpret_cast<const typename P::frag_t *>(base + ((size_t)(t * S + s) * 64 + lane) * P::J);
}

// acc[t] = sum_s A(t, s) * b[s]    for t < T, s in [s0, s0 + NS) of a layer whose fragment store has S K-steps per tile
template <bool H, int T, int NS>
__device__ __forceinline__ void fld_gemm(const typename Prec<H>::elem_t *wf, uint32_t S, uint32_t s0, const typename Prec<H>::frag_t *b, uint32_t lane,
                                         cn_f16v (&acc)[T]) {
#pragma unroll
    for (int s = 0; s < NS; s++) {
#pragma unroll
        for (int t = 0; t < T; t++) acc[t] = Prec<H>::mfma(fld_load_frag<H>(wf, t, S, s0 + s, lane), b[s], acc[t]);
    }
}

// C registers of two 32-row tiles -> the B fragments of the next layer (optionally through ReLU); fp16 mode rounds to half here
template <bool H, bool RELU>
__device__ __forceinline__ void fld_c_to_b(const cn_f16v (&acc)[2], typename Prec<H>::frag_t *b) {
    using P = Prec<H>;
#pragma unroll
    for (int u = 0; u < 2; u++) {
#pragma unroll
        for (int sub = 0; sub < P::FR; sub++) {
            if constexpr (H) {
                // round first, then ReLU on the packed halves (v_cvt_pk_f16_f32 + v_pk_max_f16 per two values): rounding is monotonic and
                // keeps the sign, so relu(round(x)) == round(relu(x)); fmaxf on the fp32 accumulators costs two v_max_f32 per value (the
                // compiler canonicalises MFMA results before a max)
                union { cn_h8 h; cn_h2 p[4]; } f;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    cn_h2 v = {(_Float16)acc[u][8 * sub + 2 * j], (_Float16)acc[u][8 * sub + 2 * j + 1]};
                    if (RELU) v = __builtin_elementwise_max(v, cn_h2{(_Float16)0, (_Float16)0});
                    f.p[j] = v;
                }
                b[u * P::FR + sub] = f.h;
            } else {
                float v = acc[u][sub];
                if (RELU) v = fmaxf(v, 0.0f);
                b[u * P::FR + sub] = v;
            }
        }
    }
}

template <int T>
__device__ __forceinline__ void fld_zero(cn_f16v (&acc)[T]) {
#pragma unroll
    for (int t = 0; t < T; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.0f;
}

// grid features of one sample as B fragments (natural order): lane (p, hi) reads levels per K-step from enc [L, P, 2]
template <bool H, int SENC>
__device__ __forceinline__ void fld_load_enc(const void *__restrict__ enc, uint32_t P_, uint32_t L, uint32_t p, bool valid, uint32_t hi,
                                             typename Prec<H>::frag_t (&b)[SENC]) {
    if constexpr (H) {
        const uint32_t *e = reinterpret_cast<const uint32_t *>(enc);      // one half2 per (level, sample)
#pragma unroll
        for (int s = 0; s < SENC; s++) {
            union { cn_h8 h; uint32_t u[4]; } f;
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {
                const uint32_t level = 8 * s + 4 * hi + jj;
                f.u[jj] = (valid && level < L) ? e[(size_t)level * P_ + p] : 0u;
            }
            b[s] = f.h;
        }
    } else {
        const float *e = reinterpret_cast<const float *>(enc);
#pragma unroll
        for (int s = 0; s < SENC; s++) {
            const uint32_t feat = 2 * s + hi, level = feat >> 1;          // K-step s holds features (2s, 2s+1) = (level s, c = hi)
            b[s] = (valid && level < L) ? e[((size_t)level * P_ + p) * 2 + (feat & 1)] : 0.0f;
        }
    }
}

template <bool H>
__device__ __forceinline__ void fld_dir_frags(const float *__restrict__ dirs, uint32_t dir_group, uint32_t p, bool valid, uint32_t hi,
                                              typename Prec<H>::frag_t *b) {
    float e[FLD_DIR];
    float dx = 0, dy = 0, dz = 0;
    if (valid) {
        const float *d = dirs + (size_t)(p / dir_group) * 3;
        dx = d[0]; dy = d[1]; dz = d[2];
    }
    fld_dir_features<H>(dx, dy, dz, e);
    if constexpr (H) {
#pragma unroll
        for (int s = 0; s < FLD_DIR / 16; s++) {
            cn_h8 f;
#pragma unroll
            for (int j = 0; j < 8; j++) f[j] = (_Float16)(hi ? e[16 * s + 8 + j] : e[16 * s + j]);
            b[s] = f;
        }
    } else {
#pragma unroll
        for (int s = 0; s < FLD_DIR / 2; s++) b[s] = hi ? e[2 * s + 1] : e[2 * s];
    }
}

__device__ __forceinline__ float fld_round_half(float v) { return (float)(_Float16)v; }

// ------------------------------------------------------------------------------------------------ forward kernel
// fp16 mode asks for two workgroups per CU-SIMD set (launch bound 2 waves per SIMD): with at most 256 registers per lane the compiler
// selects the VGPR-destination form of the MFMAs; with the 512-register budget every MFMA result lands in an accumulation register and
// costs one v_accvgpr_read per value before the VALU can touch it (the kernel was VALU-bound on exactly those moves).
template <bool H, int SENC, int NGEO>
__global__ void __launch_bounds__(FLD_THREADS, H ? 2 : 1) k_field_fwd(const void *__restrict__ enc, const float *__restrict__ xyz, const float *__restrict__ dirs,
                                                           uint32_t dir_group, uint32_t P_, FieldDims dm, const float *__restrict__ pnet,
                                                           const float *__restrict__ pden, const float *__restrict__ prgb,
                                                           float *__restrict__ sigma, float *__restrict__ rgbc, uint32_t enc_stride,
                                                           const void *__restrict__ wimg) {
    using PR = Prec<H>;
    using frag_t = typename PR::frag_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char fld_lds[];
    typename PR::elem_t *wl = reinterpret_cast<typename PR::elem_t *>(fld_lds);
    const FieldLds lo = fld_lds_layout<H>(dm);
    const bool with_rgb = rgbc != nullptr;
#ifdef CNERF_TUNING
    const bool nostage = (dir_group >> 31) != 0;                 // timing aid (results wrong): CNERF_FLD_NOSTAGE
    dir_group &= 0x7FFFFFFFu;
    if (nostage) { fld_stage_all<H>(wl, dm, lo, pnet, pden, prgb, with_rgb); __syncthreads(); }      // (staged TWICE: the delta is the staging time)
#endif
    if constexpr (H) {
        if (wimg) fld_copy_image(wl, wimg, with_rgb ? lo.off[7] : lo.off[5]);
        else fld_stage_all<H>(wl, dm, lo, pnet, pden, prgb, with_rgb);
    } else {
        fld_stage_all<H>(wl, dm, lo, pnet, pden, prgb, with_rgb);
    }
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hi = lane >> 5;
    constexpr uint32_t S64 = FLD_HID / PR::KS, SDIR = FLD_DIR / PR::KS;
    const uint32_t n_tiles = (P_ + FLD_TILE - 1) / FLD_TILE;
    unsigned char *dir_scratch = fld_lds + lo.off[7] * sizeof(typename PR::elem_t) + wave * 64;       // 64 B per wave (fld_dir_frags_uniform)
    const bool dir_uniform = H && (dir_group % FLD_TILE) == 0;
    for (uint32_t tile = blockIdx.x * FLD_WAVES + wave; tile < n_tiles; tile += gridDim.x * FLD_WAVES) {
        // keep the weight fragments in LDS: without this barrier the compiler hoists every fragment load out of the
        // persistent loop (hundreds of VGPRs, one wave per SIMD, spills in the backward)
        asm volatile("" ::: "memory");
        const uint32_t p = tile * FLD_TILE + (lane & 31);
        const bool valid = p < P_;

        frag_t x0[SENC];
        fld_load_enc<H, SENC>(enc, enc_stride, dm.L, p, valid, hi, x0);      // enc_stride = samples per level of the enc buffer (>= P_)

        cn_f16v acc[2];
        frag_t h[2 * PR::FR];                    // 64 features as B fragments
        fld_zero(acc);
        fld_gemm<H, 2, SENC>(wl + lo.off[0], SENC, 0, x0, lane, acc);
        fld_c_to_b<H, true>(acc, h);
        if (NGEO == 2) {
            fld_zero(acc);
            fld_gemm<H, 2, S64>(wl + lo.off[1], S64, 0, h, lane, acc);
            fld_c_to_b<H, true>(acc, h);
        }
        frag_t fea[2 * PR::FR];
        fld_zero(acc);
        fld_gemm<H, 2, S64>(wl + lo.off[2], S64, 0, h, lane, acc);
        fld_c_to_b<H, false>(acc, fea);           // network output: no activation (network_grid.py:98-104)

        // density head
        fld_zero(acc);
        fld_gemm<H, 2, S64>(wl + lo.off[3], S64, 0, fea, lane, acc);
        fld_c_to_b<H, true>(acc, h);
        cn_f16v out[1];
        fld_zero(out);
        fld_gemm<H, 1, S64>(wl + lo.off[4], S64, 0, h, lane, out);
        if (valid && hi == 0) {
            float raw = out[0][0];                // row 0 of the padded 16-row output
            if (H) raw = fld_round_half(raw);
            const float x = xyz[(size_t)p * 3], y = xyz[(size_t)p * 3 + 1], z = xyz[(size_t)p * 3 + 2];
            // fp16 mode: the hardware exponential (v_exp_f32, ~1 ulp + the rounding of the log2(e) product: relative error < 2e-6 at |x| <= 25,
            // far below the 1e-3 resolution of the half-precision `raw` it is applied to); float32 mode keeps the libm forms
            const float g = H ? 5.0f * __expf(-(x * x + y * y + z * z) * (1.0f / 0.08f)) : 5.0f * expf(-(x * x + y * y + z * z) / 0.08f);   // network_grid.py:150-156
            sigma[p] = H ? __expf(raw + g) : expf(raw + g);                           // trunc_exp forward (provider_utils.py:20-22)
        }

        if (with_rgb) {
            frag_t dfr[SDIR];
            if constexpr (H) {
                if (dir_uniform) {
                    const float *dp = dirs + (size_t)(__builtin_amdgcn_readfirstlane(tile) * FLD_TILE / dir_group) * 3;      // one direction per tile
                    fld_dir_frags_uniform(dp[0], dp[1], dp[2], lane, hi, dir_scratch, dfr);
                } else {
                    fld_dir_frags<H>(dirs, dir_group, p, valid, hi, dfr);
                }
            } else {
                fld_dir_frags<H>(dirs, dir_group, p, valid, hi, dfr);
            }
            constexpr uint32_t SR0 = S64 + SDIR;
            fld_zero(acc);
            fld_gemm<H, 2, S64>(wl + lo.off[5], SR0, 0, fea, lane, acc);
            fld_gemm<H, 2, SDIR>(wl + lo.off[5], SR0, S64, dfr, lane, acc);
            fld_c_to_b<H, true>(acc, h);
            fld_zero(out);
            fld_gemm<H, 1, S64>(wl + lo.off[6], S64, 0, h, lane, out);
            if (valid && hi == 0) {
                float o4[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    float v = H ? __builtin_amdgcn_rcpf(1.0f + __expf(-out[0][k])) : 1.0f / (1.0f + expf(-out[0][k]));
                    if (H) v = fld_round_half(v);
                    o4[k] = (k < (int)dm.n_rgb_out) ? v : 0.0f;
                }
                *reinterpret_cast<float4 *>(rgbc + (size_t)p * 4) = make_float4(o4[0], o4[1], o4[2], o4[3]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ C-ABI
template <bool H>
static int fld_launch_fwd(const void *enc, const float *xyz, const float *dirs, uint32_t dir_group, uint32_t P_, const FieldDims &dm,
                          const float *pnet, const float *pden, const float *prgb, float *sigma, float *rgbc, hipStream_t st, uint32_t enc_stride,
                          const void *wimg = nullptr) {
    const FieldLds lo = fld_lds_layout<H>(dm);
    const uint32_t lds_bytes = lo.off[7] * sizeof(typename Prec<H>::elem_t) + FLD_WAVES * 64;       // weight fragments + per-wave direction scratch
    const uint32_t n_tiles = cn_div_up(P_, FLD_TILE);
    uint32_t blocks = cn_div_up(n_tiles, FLD_WAVES);
    const uint32_t max_blocks = H ? 768 : 256;            // persistent: LDS allows 3 (fp16) / 1 (fp32) workgroups per CU
    if (blocks > max_blocks) blocks = max_blocks;
    const uint32_t senc = dm.enc_pad / Prec<H>::KS;
#ifdef CNERF_TUNING
    static const int nostage = cn_tune_env("CNERF_FLD_NOSTAGE", 0);
    if (nostage) dir_group |= 0x80000000u;
#endif
#define FLD_FWD_CASE(SE, NG)                                                                                                           \
    {                                                                                                                                  \
        auto kern = k_field_fwd<H, SE, NG>;                                                                                            \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);        \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(FLD_THREADS), lds_bytes, st, enc, xyz, dirs, dir_group, P_, dm, pnet, pden, prgb, sigma, rgbc, enc_stride, wimg); \
    }
    const uint32_t se16 = dm.enc_pad / 16;               // 1..4
    if (dm.n_hidden_geo == 1) {
        switch (se16) {
            case 1: FLD_FWD_CASE(16 / Prec<H>::KS, 1) break;
            case 2: FLD_FWD_CASE(32 / Prec<H>::KS, 1) break;
            case 3: FLD_FWD_CASE(48 / Prec<H>::KS, 1) break;
            case 4: FLD_FWD_CASE(64 / Prec<H>::KS, 1) break;
            default: return CNERF_EINVAL;
        }
    } else {
        switch (se16) {
            case 1: FLD_FWD_CASE(16 / Prec<H>::KS, 2) break;
            case 2: FLD_FWD_CASE(32 / Prec<H>::KS, 2) break;
            case 3: FLD_FWD_CASE(48 / Prec<H>::KS, 2) break;
            case 4: FLD_FWD_CASE(64 / Prec<H>::KS, 2) break;
            default: return CNERF_EINVAL;
        }
    }
    (void)senc;
    return cn_launch_status();
}

static int fld_dims(uint32_t enc_dim, uint32_t n_hidden_geo, uint32_t n_rgb_out, FieldDims &dm) {
    if (enc_dim == 0 || enc_dim > 64 || (enc_dim & 1)) return CNERF_EINVAL;
    if (n_hidden_geo < 1 || n_hidden_geo > 2) return CNERF_EINVAL;
    if (n_rgb_out != 3 && n_rgb_out != 4) return CNERF_EINVAL;
    dm.enc_dim = enc_dim;
    dm.enc_pad = (enc_dim + 15) / 16 * 16;
    dm.n_hidden_geo = n_hidden_geo;
    dm.n_rgb_out = n_rgb_out;
    dm.L = enc_dim / 2;
    return CNERF_OK;
}

extern "C" {

int cnerf_field_weight_image_bytes(uint32_t enc_dim, uint32_t n_hidden_geo, uint32_t n_rgb_out, uint64_t *bytes) {
    FieldDims dm;
    int rc = fld_dims(enc_dim, n_hidden_geo, n_rgb_out, dm);
    if (rc) return rc;
    if (!bytes) return CNERF_ENULL;
    *bytes = (uint64_t)fld_lds_layout<true>(dm).off[7] * sizeof(_Float16);
    return CNERF_OK;
}

int cnerf_field_pack_weights(uint32_t enc_dim, uint32_t n_hidden_geo, uint32_t n_rgb_out, const float *params_net, const float *params_den,
                             const float *params_rgb, void *image, uint64_t image_bytes, void *stream) {
    FieldDims dm;
    int rc = fld_dims(enc_dim, n_hidden_geo, n_rgb_out, dm);
    if (rc) return rc;
    if (!params_net || !params_den || !params_rgb || !image) return CNERF_ENULL;
    const uint32_t n = fld_lds_layout<true>(dm).off[7];
    if ((((uintptr_t)image) & 15) || image_bytes < (uint64_t)n * sizeof(_Float16)) return CNERF_EINVAL;
    // ~4 weights per thread: the packing is latency, not bandwidth (25 k scattered loads)
    hipLaunchKernelGGL(k_field_pack, dim3(cn_div_up(n, 4 * FLD_THREADS)), dim3(FLD_THREADS), 0, CN_STREAM(stream), reinterpret_cast<_Float16 *>(image), dm,
                       params_net, params_den, params_rgb);
    return cn_launch_status();
}

int cnerf_field_forward_img(const void *enc, const float *xyz, const float *dirs, uint32_t dir_group, uint32_t P_, uint32_t enc_dim,
                            uint32_t n_hidden_geo, uint32_t n_rgb_out, const float *params_net, const float *params_den, const float *params_rgb,
                            float *sigma, float *rgbc, int dtype, uint32_t enc_level_stride, const void *weight_image, void *stream) {
    FieldDims dm;
    int rc = fld_dims(enc_dim, n_hidden_geo, n_rgb_out, dm);
    if (rc) return rc;
    if (!weight_image || dtype != CNERF_F16)
        return cnerf_field_forward_strided(enc, xyz, dirs, dir_group, P_, enc_dim, n_hidden_geo, n_rgb_out, params_net, params_den, params_rgb, sigma, rgbc,
                                           dtype, enc_level_stride, stream);
    if (((uintptr_t)weight_image) & 15) return CNERF_EINVAL;
    if (enc_level_stride == 0) enc_level_stride = P_;
    if (enc_level_stride < P_) return CNERF_EINVAL;
    if (P_ == 0) return CNERF_OK;
    if (!enc || !xyz || !params_net || !params_den || !sigma) return CNERF_ENULL;
    if (rgbc && (!dirs || !params_rgb || dir_group == 0)) return CNERF_ENULL;
    if (rgbc && (((uintptr_t)rgbc) & 15)) return CNERF_EINVAL;
    return fld_launch_fwd<true>(enc, xyz, dirs, dir_group, P_, dm, params_net, params_den, params_rgb, sigma, rgbc, CN_STREAM(stream), enc_level_stride,
                                weight_image);
}

int cnerf_field_forward_strided(const void *enc, const float *xyz, const float *dirs, uint32_t dir_group, uint32_t P_, uint32_t enc_dim,
                                uint32_t n_hidden_geo, uint32_t n_rgb_out, const float *params_net, const float *params_den, const float *params_rgb,
                                float *sigma, float *rgbc, int dtype, uint32_t enc_level_stride, void *stream) {
    FieldDims dm;
    int rc = fld_dims(enc_dim, n_hidden_geo, n_rgb_out, dm);
    if (rc) return rc;
    if (dtype != CNERF_F32 && dtype != CNERF_F16) return CNERF_EINVAL;
    if (enc_level_stride == 0) enc_level_stride = P_;
    if (enc_level_stride < P_) return CNERF_EINVAL;
    if (P_ == 0) return CNERF_OK;
    if (!enc || !xyz || !params_net || !params_den || !sigma) return CNERF_ENULL;
    if (rgbc && (!dirs || !params_rgb || dir_group == 0)) return CNERF_ENULL;
    if (rgbc && (((uintptr_t)rgbc) & 15)) return CNERF_EINVAL;
    if (dtype == CNERF_F16)
        return fld_launch_fwd<true>(enc, xyz, dirs, dir_group, P_, dm, params_net, params_den, params_rgb, sigma, rgbc, CN_STREAM(stream), enc_level_stride);
    return fld_launch_fwd<false>(enc, xyz, dirs, dir_group, P_, dm, params_net, params_den, params_rgb, sigma, rgbc, CN_STREAM(stream), enc_level_stride);
}

int cnerf_field_forward(const void *enc, const float *xyz, const float *dirs, uint32_t dir_group, uint32_t P_, uint32_t enc_dim,
                        uint32_t n_hidden_geo, uint32_t n_rgb_out, const float *params_net, const float *params_den, const float *params_rgb,
                        float *sigma, float *rgbc, int dtype, void *stream) {
    return cnerf_field_forward_strided(enc, xyz, dirs, dir_group, P_, enc_dim, n_hidden_geo, n_rgb_out, params_net, params_den, params_rgb, sigma, rgbc, dtype,
                                       P_, stream);
}

}  // extern "C"
