// Fused field backward (gfx950 matrix cores), two launches:
//   k_field_bwd_data  per 32-sample tile: recompute the forward (nothing was saved), run the activation-gradient chain
//                     dz_l = (W_{l+1}^T dz_{l+1}) * relu'(.) with the C-register trick of field_common.h (transposed weight
//                     fragments, again no cross-lane traffic), write d(loss)/d(grid features) in the encoder's [L,P,2]
//                     layout, and spill every dz_l / layer input once as [row][sample] matrices;
//   k_field_bwd_dw    dW_l = dz_l . a_{l-1}^T : MFMA GEMMs whose contraction runs over the samples, reading those
//                     [row][sample] matrices with 16-byte per-lane loads; split-K over the sample axis, per-workgroup LDS
//                     reduction, one float atomic per weight per split (24.5 k x splits — negligible).
// trunc_exp backward clamps the exponent to [-15, 15] (provider_utils.py:26-29).
#pragma once
#include "field_common.h"

// ---- re-declared helpers shared with field.hip (kept header-free on purpose: both TUs instantiate their own copies)
template <bool H>
__device__ __forceinline__ typename Prec<H>::frag_t fb_load_frag(const typename Prec<H>::elem_t *base, uint32_t t, uint32_t S, uint32_t s, uint32_t lane) {
    using P = Prec<H>;
    return *reinterpret_cast<const typename P::frag_t *>(base + ((size_t)(t * S + s) * 64 + lane) * P::J);
}

template <int T>
__device__ __forceinline__ void fb_zero(cn_f16v (&acc)[T]) {
#pragma unroll
    for (int t = 0; t < T; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.0f;
}

template <bool H, int T, int NS>
__device__ __forceinline__ void fb_gemm(const typename Prec<H>::elem_t *wf, uint32_t S, uint32_t s0, const typename Prec<H>::frag_t *b, uint32_t lane,
                                        cn_f16v (&acc)[T]) {
    asm volatile("" ::: "memory");      // scheduling fence: keeps this layer's fragment loads from being hoisted above the previous layer
#pragma unroll
    for (int s = 0; s < NS; s++) {
#pragma unroll
        for (int t = 0; t < T; t++) acc[t] = Prec<H>::mfma(fb_load_frag<H>(wf, t, S, s0 + s, lane), b[s], acc[t]);
    }
}

// forward-order staging (same as field.hip).  SWZ: 16-byte fragment slots bank-swizzled for the transposing reads of field_bwd_x2.hip
// (slot bits 2..3 ^= (K-step parity, half) — halves index i, i.e. byte offset 2 i; needs an even S)
template <bool H, int KIND, bool SWZ = false>
__device__ __forceinline__ void fb_stage_layer(typename Prec<H>::elem_t *dst, const float *__restrict__ W, uint32_t rows, uint32_t in_stride,
                                               uint32_t T, uint32_t S, uint32_t n_valid_cols) {
    using P = Prec<H>;
    const uint32_t total = T * S * 64 * P::J;
    for (uint32_t i = threadIdx.x; i < total; i += blockDim.x) {
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
        dst[SWZ ? (i ^ (((i >> 8) & 3u) << 5)) : i] = (typename P::elem_t)v;
    }
}

// transposed staging: A fragment of W^T.  tile t runs over the layer's INPUT features (col0 + 32 t + i), the K-slots over its
// OUTPUT rows in C-register order: dst[((t S + s) 64 + lane) J + j] = W[clayout(s, hi, j)][col0 + 32 t + (lane & 31)]
template <bool H>
__device__ __forceinline__ void fb_stage_layer_T(typename Prec<H>::elem_t *dst, const float *__restrict__ W, uint32_t rows, uint32_t in_stride,
                                                 uint32_t col0, uint32_t n_in, uint32_t T, uint32_t S) {
    using P = Prec<H>;
    const uint32_t total = T * S * 64 * P::J;
    for (uint32_t i = threadIdx.x; i < total; i += blockDim.x) {
        const uint32_t j = i % P::J, lane = (i / P::J) % 64, ts = i / (P::J * 64);
        const uint32_t s = ts % S, t = ts / S;
        const uint32_t col = 32 * t + (lane & 31), hi = lane >> 5;
        const uint32_t row = (uint32_t)fld_col_clayout<H>(s, hi, j);
        float v = 0.0f;
        if (row < rows && col < n_in) v = W[(size_t)row * in_stride + col0 + col];
        dst[i] = (typename P::elem_t)v;
    }
}

// transposed A fragment straight from the row-major float32 parameters (fp32 mode: no LDS room for a second copy;
// lanes i read consecutive columns -> coalesced)
__device__ __forceinline__ float fb_frag_T_global(const float *__restrict__ W, uint32_t rows, uint32_t in_stride, uint32_t col0, uint32_t n_in,
                                                  uint32_t t, uint32_t s, uint32_t lane) {
    const uint32_t col = 32 * t + (lane & 31), hi = lane >> 5;
    const uint32_t row = (uint32_t)fld_col_clayout<false>(s, hi, 0);
    return (row < rows && col < n_in) ? W[(size_t)row * in_stride + col0 + col] : 0.0f;
}

// da[t] += W^T(t, s) dz[s]   — one transposed layer product
template <bool H, int T, int NS>
__device__ __forceinline__ void fb_gemm_T(const typename Prec<H>::elem_t *wt_lds, const float *__restrict__ W, uint32_t rows, uint32_t in_stride,
                                          uint32_t col0, uint32_t n_in, uint32_t S, const typename Prec<H>::frag_t *b, uint32_t lane, cn_f16v (&acc)[T]) {
    asm volatile("" ::: "memory");
#pragma unroll
    for (int s = 0; s < NS; s++) {
#pragma unroll
        for (int t = 0; t < T; t++) {
            typename Prec<H>::frag_t a;
            if constexpr (H) a = fb_load_frag<H>(wt_lds, t, S, s, lane);
            else a = fb_frag_T_global(W, rows, in_stride, col0, n_in, t, s, lane);
            acc[t] = Prec<H>::mfma(a, b[s], acc[t]);
        }
    }
}

template <bool H, bool RELU>
__device__ __forceinline__ void fb_c_to_b(const cn_f16v (&acc)[2], typename Prec<H>::frag_t *b) {
    using P = Prec<H>;
#pragma unroll
    for (int u = 0; u < 2; u++) {
#pragma unroll
        for (int sub = 0; sub < P::FR; sub++) {
            if constexpr (H) {
                cn_h8 f;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    float v = acc[u][8 * sub + j];
                    if (RELU) v = fmaxf(v, 0.0f);
                    f[j] = (_Float16)v;
                }
                b[u * P::FR + sub] = f;
            } else {
                float v = acc[u][sub];
                if (RELU) v = fmaxf(v, 0.0f);
                b[u * P::FR + sub] = v;
            }
        }
    }
}

// dz = da * [act > 0]   (act = the forward's post-ReLU fragments, same register mapping as the C tiles)
template <bool H>
__device__ __forceinline__ void fb_c_to_b_masked(const cn_f16v (&acc)[2], const typename Prec<H>::frag_t *act, typename Prec<H>::frag_t *b) {
    using P = Prec<H>;
#pragma unroll
    for (int u = 0; u < 2; u++) {
#pragma unroll
        for (int sub = 0; sub < P::FR; sub++) {
            if constexpr (H) {
                cn_h8 f;
                const cn_h8 a = act[u * P::FR + sub];
#pragma unroll
                for (int j = 0; j < 8; j++) f[j] = (a[j] > (_Float16)0) ? (_Float16)acc[u][8 * sub + j] : (_Float16)0;
                b[u * P::FR + sub] = f;
            } else {
                b[u * P::FR + sub] = (act[u * P::FR + sub] > 0.0f) ? acc[u][sub] : 0.0f;
            }
        }
    }
}

// spill 64 rows held as C-ordered B fragments into a [row][sample] matrix (row stride ld)
template <bool H>
__device__ __forceinline__ void fb_dump_clayout(typename Prec<H>::elem_t *__restrict__ M, size_t ld, uint32_t p, uint32_t hi, const typename Prec<H>::frag_t *b) {
    using P = Prec<H>;
#pragma unroll
    for (int s = 0; s < 2 * P::FR; s++) {
#pragma unroll
        for (int j = 0; j < P::J; j++) {
            const int row = fld_col_clayout<H>(s, hi, j);
            if constexpr (H) M[(size_t)row * ld + p] = b[s][j];
            else M[(size_t)row * ld + p] = b[s];
        }
    }
}

// spill natural-ordered fragments (grid features / direction features): rows < n_rows only
template <bool H, int NS>
__device__ __forceinline__ void fb_dump_natural(typename Prec<H>::elem_t *__restrict__ M, size_t ld, uint32_t p, uint32_t hi, const typename Prec<H>::frag_t *b,
                                                uint32_t n_rows) {
    using P = Prec<H>;
#pragma unroll
    for (int s = 0; s < NS; s++) {
#pragma unroll
        for (int j = 0; j < P::J; j++) {
            const uint32_t row = (uint32_t)fld_col_natural<H>(s, hi, j);
            if (row < n_rows) {
                if constexpr (H) M[(size_t)row * ld + p] = b[s][j];
                else M[(size_t)row * ld + p] = b[s];
            }
        }
    }
}

template <bool H, int SENC>
__device__ __forceinline__ void fb_load_enc(const void *__restrict__ enc, uint32_t P_, uint32_t L, uint32_t p, bool valid, uint32_t hi,
                                            typename Prec<H>::frag_t (&b)[SENC]) {
    if constexpr (H) {
        const uint32_t *e = reinterpret_cast<const uint32_t *>(enc);
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
            const uint32_t feat = 2 * s + hi, level = feat >> 1;
            b[s] = (valid && level < L) ? e[((size_t)level * P_ + p) * 2 + (feat & 1)] : 0.0f;
        }
    }
}

// direction features from an already loaded direction (padded samples must spill zeros: cos(0) = 1 otherwise)
template <bool H>
__device__ __forceinline__ void fb_dir_frags_from(float dx, float dy, float dz, bool valid, uint32_t hi, typename Prec<H>::frag_t *b) {
    float e[FLD_DIR];
    fld_dir_features<H>(dx, dy, dz, e);
    if (!valid) {
#pragma unroll
        for (int q = 0; q < FLD_DIR; q++) e[q] = 0.0f;
    }
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

template <bool H>
__device__ __forceinline__ void fb_dir_frags(const float *__restrict__ dirs, uint32_t dir_group, uint32_t p, bool valid, uint32_t hi,
                                             typename Prec<H>::frag_t *b) {
    float dx = 0, dy = 0, dz = 0;
    if (valid) {
        const float *d = dirs + (size_t)(p / dir_group) * 3;
        dx = d[0]; dy = d[1]; dz = d[2];
    }
    fb_dir_frags_from<H>(dx, dy, dz, valid, hi, b);
}

// LDS layout of the backward kernel: forward fragment stores (as field.hip) then the transposed stores (fp16 only)
struct FieldLdsT {
    uint32_t off[8];   // n0T, n1T, n2T, d0T, doT, r0T, roT, end   (elements, relative to the transposed area)
};
template <bool H>
__host__ __device__ __forceinline__ FieldLdsT fb_ldsT_layout(const FieldDims &d) {
    FieldLdsT l;
    const uint32_t t0 = (d.enc_pad + 31) / 32;
    uint32_t o = 0;
    l.off[0] = o; o += 32 * t0 * FLD_HID;                                    // n0T: t0 tiles x K=64
    l.off[1] = o; o += (d.n_hidden_geo == 2) ? FLD_HID * FLD_HID : 0;
    l.off[2] = o; o += FLD_HID * FLD_HID;
    l.off[3] = o; o += FLD_HID * FLD_HID;
    l.off[4] = o; o += FLD_HID * 32;                                         // doT: 2 tiles x K=32
    l.off[5] = o; o += FLD_HID * FLD_HID;                                    // r0T (fea columns only)
    l.off[6] = o; o += FLD_HID * 32;
    l.off[7] = o;
    return l;
}

