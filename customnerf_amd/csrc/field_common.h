// Fused NeRF field (grid features -> sigma / rgb MLPs) on the gfx950 matrix cores: shared definitions.
//
// The three bias-free MLPs of nerf/network_grid.py:98-139 are evaluated per 32-sample tile by one wavefront with
// v_mfma_f32_32x32x16_f16 (fp16 mode: tinycudann FullyFusedMLP numerics) or v_mfma_f32_32x32x2_f32 (fp32 mode: exact
// float32, same rate as the vector ALU but with the same data flow).  Orientation: Y[out, sample] = W[out, k] X[k, sample]:
// A = weights (from LDS, pre-arranged in fragment order), B = activations, C = 32 outputs x 32 samples.
//
// Key layout fact (CDNA4 MFMA): lane l of a wave owns sample (l & 31); for a C tile it holds output rows
//   rho(r, hi) = (r & 3) + 8 (r >> 2) + 4 hi,  r = 0..15, hi = l >> 5,
// and for a B fragment it holds K-slots (hi, j).  Since the contraction index can be permuted freely as long as A and B
// agree, the next layer takes the C registers of the previous one *as they are* as its B fragments (K-slot (s, hi, j) :=
// feature 32 u + rho(r, hi) with r = (s mod FR) J + j, u = s / FR) and the weight fragments are staged into LDS in that
// permuted column order once per workgroup.  No cross-lane traffic between layers.
#pragma once
#include "common.h"

typedef _Float16 cn_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 cn_h2 __attribute__((ext_vector_type(2)));
typedef float cn_f16v __attribute__((ext_vector_type(16)));

#define FLD_THREADS 256
#define FLD_WAVES 4
#define FLD_TILE 32                 // samples per wave tile
#define FLD_HID 64                  // hidden width (the only width the reference uses)
#define FLD_DIR 32                  // 27 frequency features of the view direction, padded to 32
#define FLD_NDIR 27

template <bool HALF> struct Prec;
template <> struct Prec<true> {
    using elem_t = _Float16;
    using frag_t = cn_h8;
    static constexpr int KS = 16;   // k consumed per MFMA
    static constexpr int J = 8;     // elements per lane per fragment
    static constexpr int FR = 2;    // B fragments one 32-row C tile turns into
    static __device__ __forceinline__ cn_f16v mfma(frag_t a, frag_t b, cn_f16v c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ frag_t zero() { return frag_t{0, 0, 0, 0, 0, 0, 0, 0}; }
};
template <> struct Prec<false> {
    using elem_t = float;
    using frag_t = float;
    static constexpr int KS = 2;
    static constexpr int J = 1;
    static constexpr int FR = 16;
    static __device__ __forceinline__ cn_f16v mfma(frag_t a, frag_t b, cn_f16v c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ frag_t zero() { return 0.0f; }
};

__host__ __device__ __forceinline__ int fld_rho(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// feature index of K-slot (s, hi, j) for activations that arrive in C-register order / in natural order
template <bool H> __host__ __device__ __forceinline__ int fld_col_clayout(int s, int hi, int j) {
    const int u = s / Prec<H>::FR, r = (s % Prec<H>::FR) * Prec<H>::J + j;
    return 32 * u + fld_rho(r, hi);
}
template <bool H> __host__ __device__ __forceinline__ int fld_col_natural(int s, int hi, int j) { return Prec<H>::KS * s + Prec<H>::J * hi + j; }

// Network geometry (all widths in elements)
struct FieldDims {
    uint32_t enc_dim;        // L * C (<= 64)
    uint32_t enc_pad;        // padded to x16
    uint32_t n_hidden_geo;   // 1 or 2 hidden layers in `network`
    uint32_t n_rgb_out;      // 3 or 4
    uint32_t L;              // levels (enc_dim / 2)
};

// LDS fragment-store offsets (in elements) of the seven layers: n0, n1, n2, d0, do, r0, ro
struct FieldLds {
    uint32_t off[8];
};

template <bool H>
__host__ __device__ __forceinline__ FieldLds fld_lds_layout(const FieldDims &d) {
    FieldLds l;
    uint32_t o = 0;
    l.off[0] = o; o += FLD_HID * d.enc_pad;                              // n0  [64, enc_pad]
    l.off[1] = o; o += (d.n_hidden_geo == 2) ? FLD_HID * FLD_HID : 0;    // n1  [64, 64]
    l.off[2] = o; o += FLD_HID * FLD_HID;                                // n2  [64, 64]
    l.off[3] = o; o += FLD_HID * FLD_HID;                                // d0  [64, 64]
    l.off[4] = o; o += 32 * FLD_HID;                                     // do  one 32-row tile (16 parameter rows + zeros)
    l.off[5] = o; o += FLD_HID * (FLD_HID + FLD_DIR);                    // r0  [64, 64 fea + 32 dir]
    l.off[6] = o; o += 32 * FLD_HID;                                     // ro
    l.off[7] = o;
    return l;
}

// Direction features of a 32-sample tile whose samples share ONE direction (dir_group a multiple of the tile: the renderer's run() path,
// one direction per ray).  Instead of every lane evaluating all 24 sines / cosines, lane q < 27 evaluates feature q, the 32 halves go
// through a 64-byte LDS scratch of the wave and come back as the two natural-order B fragments (broadcast reads).  `scratch` must be
// 16-byte aligned and private to the wave; DS operations of one wave execute in order, no barrier is involved.
__device__ __forceinline__ void fld_dir_frags_uniform(float dx, float dy, float dz, uint32_t lane, uint32_t hi, unsigned char *scratch, cn_h8 *b) {
    const uint32_t q = lane & 31;
    const uint32_t qq = q >= 3 ? q - 3 : 0, k = qq / 6, r = qq % 6, c = q < 3 ? q : (r < 3 ? r : r - 3);
    const float d = c == 0 ? dx : (c == 1 ? dy : dz);
    const float a = d * (float)(1u << k);
    const float sc = r < 3 ? __sinf(a) : __cosf(a);
    const float v = q < 3 ? d : (q < FLD_NDIR ? sc : 0.0f);
    if (lane < 32) reinterpret_cast<_Float16 *>(scratch)[q] = (_Float16)v;
    b[0] = *reinterpret_cast<const cn_h8 *>(scratch + 16 * hi);
    b[1] = *reinterpret_cast<const cn_h8 *>(scratch + 32 + 16 * hi);
}

// frequency encoding of a direction (nerf/base.py:42-60): [d, sin(2^k d), cos(2^k d)]_{k=0..3}, 27 values padded to 32
template <bool FAST>
__device__ __forceinline__ void fld_dir_features(float dx, float dy, float dz, float (&e)[FLD_DIR]) {
    const float d[3] = {dx, dy, dz};
#pragma unroll
    for (int c = 0; c < 3; c++) e[c] = d[c];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float f = (float)(1 << k);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float s, co;
            if (FAST) { s = __sinf(d[c] * f); co = __cosf(d[c] * f); }
            else { s = sinf(d[c] * f); co = cosf(d[c] * f); }
            e[3 + 6 * k + c] = s;
            e[6 + 6 * k + c] = co;
        }
    }
#pragma unroll
    for (int q = FLD_NDIR; q < FLD_DIR; q++) e[q] = 0.0f;
}
