// Multiresolution hash / tiled grid encoder for gfx950 (CDNA4).
//
// Replaces the reference's `_gridencoder` extension (gridencoder/src/gridencoder.cu) behind the C-ABI of
// include/customnerf_hip.h.  MI355X-first choices:
//   * level geometry (table offset/size, scale, resolution) is computed once on the host and travels as kernel
//     arguments — no per-thread exp2f/ceil, and the oracle and the kernel see bit-identical geometry;
//   * one thread = one (point, level); the 1-D grid is swizzled so that the two levels an XCD works on stay in that
//     XCD's private 4 MiB L2 (a 2^19-entry level is 2 MiB in fp16, 4 MiB in fp32) — XCD x owns a contiguous slice
//     of the level-major work list, and the level list is interleaved coarse/fine so the slices cost the same;
//   * corner features are fetched with one vector load per corner (4 B for fp16 C=2, 8 B for fp32 C=2);
//   * the backward scatter uses hardware float atomics (global_atomic_add_f32) into a float32 gradient table
//     for both table dtypes.
// Arithmetic follows oracle/gridencoder_ref.c operation by operation (compiled -ffp-contract=off, explicit fma).
#include "grid_common.h"

template <typename T, int D, int C>
__global__ void __launch_bounds__(GE_BLOCK) k_grid_fwd(const float *__restrict__ inputs, const T *__restrict__ grid, const GridLevels lv,
                                                       T *__restrict__ outputs, uint32_t B, uint32_t L, uint32_t n_levels, uint32_t nb,
                                                       T *__restrict__ dy_dx, uint32_t gridtype, int align_corners, uint32_t interp, int swizzle,
                                                       uint32_t ostride) {
    uint32_t level, pb;
    if (!ge_work_item(nb, n_levels, swizzle, lv, level, pb)) return;
    const uint32_t b = pb * GE_BLOCK + threadIdx.x;
    if (b >= B) return;

    using Vec = FeatVec<T, C>;
    const Vec *__restrict__ table = reinterpret_cast<const Vec *>(grid) + lv.offset[level];
    Vec *out = reinterpret_cast<Vec *>(outputs) + ((size_t)level * ostride + b);      // ostride = rows per level of the output buffer (>= B)

    float in[D];
    bool oob = false;
    ge_load_coords<D>(inputs, b, in);
#pragma unroll
    for (int d = 0; d < D; d++) oob = oob || (in[d] < 0 || in[d] > 1);
    if (oob) {
        Vec z;
#pragma unroll
        for (int c = 0; c < C; c++) z.v[c] = ge_from_float<T>(0.0f);
        *out = z;
        if (dy_dx) {
            T *dd = dy_dx + (size_t)b * D * L * C + (size_t)level * D * C;
#pragma unroll
            for (int i = 0; i < D * C; i++) dd[i] = ge_from_float<T>(0.0f);
        }
        return;
    }

    const uint32_t hashmap_size = lv.size[level];
    const float scale = lv.scale[level];
    const uint32_t resolution = lv.resolution[level];

    float pos[D], pos_deriv[D];
    uint32_t pos_grid[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
        pos[d] = cn_fma(in[d], scale, align_corners ? 0.0f : 0.5f);
        const float fl = floorf(pos[d]);
        pos_grid[d] = (uint32_t)fl;
        pos[d] -= (float)pos_grid[d];
        if (interp == 1) {
            pos_deriv[d] = ge_smoothstep_derivative(pos[d]);
            pos[d] = ge_smoothstep(pos[d]);
        } else {
            pos_deriv[d] = 1.0f;
        }
    }

    // Corner gathers, two corners per request where memory allows it.  The corners 2q / 2q+1 differ only in x: on dense
    // (un-hashed) levels their entries are neighbours in memory, and on hashed levels they share one aligned entry pair
    // whenever x is even (the x term of the hash is the identity prime, so x -> x+1 flips only bit 0).  One under-aligned
    // two-entry load then serves both (global_load_dwordx2 for fp16 C=2, dwordx4 for fp32 C=2); the second, lane-masked
    // load runs only for lanes whose pair is not adjacent.  The gather is bound by the number of cache-line requests, not by
    // bytes, so this removes a quarter (hashed) to a half (dense) of the kernel's work.  All loads are issued before the
    // accumulation, which keeps the reference's corner order (bit-exact results).
    // Hashed levels with 4-byte entries (fp16 C=2) use an aligned FOUR-entry window instead (global_load_dwordx4): x -> x+1 flips the
    // low bits of x only, so the partner lies in the same aligned quad unless x = 3 mod 4 (XOR distance 1 or 3): the masked second
    // load then runs for a quarter of the lanes instead of half.
    struct alignas(sizeof(T) * C) VecPair { Vec a, b; };
    struct alignas(sizeof(T) * C * 4) VecQuad { Vec e[4]; };
    Vec corner[1 << D];
    float wgt[1 << D];
    bool quad = false;
    if constexpr (sizeof(Vec) == 4) {
        quad = gridtype == 0 && ((lv.offset[level] | hashmap_size) & 3u) == 0 && ge_is_hashed<D>(align_corners, hashmap_size, resolution);
    }
    ge_dispatch_mode(ge_level_mode<D>(gridtype, align_corners, hashmap_size, resolution), [&](auto mode_c) {
#pragma unroll
    for (int q = 0; q < (1 << (D - 1)); q++) {
        uint32_t pgl[D];
#pragma unroll
        for (int d = 1; d < D; d++) pgl[d] = (((2 * q) & (1 << d)) == 0) ? pos_grid[d] : pos_grid[d] + 1;
        // weights in the reference's multiplication order: x factor first (gridencoder.cu:171-178)
        float w0 = 1 - pos[0], w1 = pos[0];
#pragma unroll
        for (int d = 1; d < D; d++) {
            const float f = (((2 * q) & (1 << d)) == 0) ? (1 - pos[d]) : pos[d];
            w0 *= f; w1 *= f;
        }
        wgt[2 * q] = w0; wgt[2 * q + 1] = w1;
        pgl[0] = pos_grid[0];
        const uint32_t i0 = ge_index_m<D, decltype(mode_c)::value>(gridtype, align_corners, hashmap_size, resolution, pgl);
        pgl[0] = pos_grid[0] + 1;
        const uint32_t i1 = ge_index_m<D, decltype(mode_c)::value>(gridtype, align_corners, hashmap_size, resolution, pgl);
        if (quad) {
            if constexpr (sizeof(Vec) == 4) {
                const VecQuad v = *reinterpret_cast<const VecQuad *>(table + (i0 & ~3u));
                const uint32_t k0 = i0 & 3u, k1 = i1 & 3u;
                const Vec a01 = (k0 & 1) ? v.e[1] : v.e[0], a23 = (k0 & 1) ? v.e[3] : v.e[2];
                corner[2 * q] = (k0 & 2) ? a23 : a01;
                const Vec b01 = (k1 & 1) ? v.e[1] : v.e[0], b23 = (k1 & 1) ? v.e[3] : v.e[2];
                Vec e1 = (k1 & 2) ? b23 : b01;
                if ((i0 ^ i1) > 3u) e1 = table[i1];
                corner[2 * q + 1] = e1;
            }
        } else {
            const uint32_t lo = min(i0, i1), hi_ = max(i0, i1);
            const uint32_t b = (lo + 1 < hashmap_size) ? lo : lo - 1;          // two-entry window stays inside the level (tables hold >= 8 entries)
            const VecPair v = *reinterpret_cast<const VecPair *>(table + b);
            const Vec e_lo = (b == lo) ? v.a : v.b;
            Vec e_hi = v.b;
            if (hi_ - lo != 1) e_hi = table[hi_];
            corner[2 * q] = (i0 == lo) ? e_lo : e_hi;
            corner[2 * q + 1] = (i0 == lo) ? e_hi : e_lo;
        }
    }
    });
    Vec res;
    if constexpr (sizeof(T) == 2 && C == 2) {
        __half2 r2 = __floats2half2_rn(0.0f, 0.0f);
#pragma unroll
        for (int idx = 0; idx < (1 << D); idx++) ge_accum2(r2, wgt[idx], *reinterpret_cast<const __half2 *>(&corner[idx]));
        res = *reinterpret_cast<const Vec *>(&r2);
    } else {
#pragma unroll
        for (int c = 0; c < C; c++) res.v[c] = ge_from_float<T>(0.0f);
#pragma unroll
        for (int idx = 0; idx < (1 << D); idx++) {
#pragma unroll
            for (int c = 0; c < C; c++) ge_accum(res.v[c], wgt[idx], corner[idx].v[c]);
        }
    }
    *out = res;

    if (dy_dx) {
        T *dd = dy_dx + (size_t)b * D * L * C + (size_t)level * D * C;
#pragma unroll
        for (int gd = 0; gd < D; gd++) {
            T rg[C];
#pragma unroll
            for (int c = 0; c < C; c++) rg[c] = ge_from_float<T>(0.0f);
#pragma unroll
            for (int idx = 0; idx < (1 << (D - 1)); idx++) {
                float w = scale;
                uint32_t pgl[D];
#pragma unroll
                for (int nd = 0; nd < D - 1; nd++) {
                    const int d = (nd >= gd) ? (nd + 1) : nd;
                    if ((idx & (1 << nd)) == 0) { w *= 1 - pos[d]; pgl[d] = pos_grid[d]; }
                    else { w *= pos[d]; pgl[d] = pos_grid[d] + 1; }
                }
                pgl[gd] = pos_grid[gd];
                const Vec left = table[ge_index<D>(gridtype, align_corners, hashmap_size, resolution, pgl)];
                pgl[gd] = pos_grid[gd] + 1;
                const Vec right = table[ge_index<D>(gridtype, align_corners, hashmap_size, resolution, pgl)];
#pragma unroll
                for (int c = 0; c < C; c++) {
                    if constexpr (sizeof(T) == 2) {
                        const float diff = __half2float(__float2half_rn(ge_opaque(__half2float(right.v[c]) - __half2float(left.v[c]))));
                        const float prod = ge_opaque(w * diff * pos_deriv[gd]);
                        rg[c] = __float2half_rn(ge_opaque(__half2float(rg[c]) + __half2float(__float2half_rn(prod))));
                    } else {
                        const float diff = right.v[c] - left.v[c];
                        rg[c] = cn_fma(w * diff, pos_deriv[gd], rg[c]);
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < C; c++) dd[gd * C + c] = rg[c];
        }
    }
}

// ------------------------------------------------------------------------------------------------ specialised forward gather
// k_grid_fwd above is generic over table dtype / D / C / grid type / interpolation / dy_dx and pays for it in instructions: PMC passes of
// round 2 (profiles/r02_*) show the gather busy ISSUING ~75 % of its time (202 vector + 152 scalar instructions per (sample, level) wave
// item), not waiting on L2.  This kernel is the configuration CustomNeRF runs (fp16 table, D = 3, C = 2, linear interpolation, no input
// gradient, align_corners = false) as straight-line code per level mode, chosen by one scalar branch per workgroup:
//   * 32-bit byte offsets against a scalar table base (global_load ... , off-by-SGPR) instead of 64-bit address arithmetic;
//   * the half x float products of the trilinear sum as v_fma_mix_f32 (the half operand is widened inside the FMA: bit-identical to
//     convert-then-multiply, the conversion being exact; addend -0.0 keeps the sign of a zero product);
//   * hashed levels: x + 1 differs from x in its trailing-ones run only, so the partner entry of an x-pair is entry k ^ 1 (x even) or
//     k ^ 3 (x = 1 mod 4) of the same aligned 16-byte window; only x = 3 mod 4 needs a second, lane-masked load;
//   * the weights in the reference's multiplication order ((wx wy) wz, gridencoder.cu:171-178) shared across the corner pairs.
// Results are bit-identical to k_grid_fwd (same operations on the same values in the same order).
#include "grid_fwd_eval.h"

__global__ void __launch_bounds__(GE_BLOCK) k_grid_fwd_fast(const float *__restrict__ inputs, const __half *__restrict__ grid, const GridLevels lv,
                                                            __half *__restrict__ outputs, uint32_t B, uint32_t n_levels, uint32_t nb, uint32_t gridtype,
                                                            int swizzle, uint32_t ostride) {
    uint32_t level, pb;
    if (!ge_work_item(nb, n_levels, swizzle, lv, level, pb)) return;
    const uint32_t b = pb * GE_BLOCK + threadIdx.x;
    if (b >= B) return;
    uint32_t *out = reinterpret_cast<uint32_t *>(outputs) + ((size_t)level * ostride + b);
    float in[3];
    ge_load_coords<3>(inputs, b, in);
    if (in[0] < 0 || in[0] > 1 || in[1] < 0 || in[1] > 1 || in[2] < 0 || in[2] > 1) { *out = 0u; return; }

    const unsigned char *__restrict__ table = reinterpret_cast<const unsigned char *>(grid) + (size_t)lv.offset[level] * 4;   // 4 bytes per entry
    *out = gf_eval_level(in, table, lv.size[level], lv.resolution[level], lv.scale[level], gridtype);
}

// Sample-major traversal: a workgroup evaluates ALL levels of a tile of 256 SPT consecutive samples (level loop outside, the tile's samples
// inside).  Every XCD touches every table, so this loses to the level-major list when the samples are spread over the volume (the 16 tables
// do not fit an L2) and wins when neighbours of the list are neighbours in space — the importance samples of a fitted field, whose rows are
// re-used from L1 / L2 across the tile (scratch/fused_fwd_lab.hip, profiles/r05_fused_fwd_lab.log: 153 vs 188 us).  Bit-identical results.
template <int SPT>
__global__ void __launch_bounds__(GE_BLOCK) k_grid_fwd_fast_sm(const float *__restrict__ inputs, const __half *__restrict__ grid, const GridLevels lv,
                                                               __half *__restrict__ outputs, uint32_t B, uint32_t n_levels, uint32_t gridtype,
                                                               uint32_t ostride) {
    const uint32_t t0 = blockIdx.x * (GE_BLOCK * SPT) + threadIdx.x;
    if (t0 >= B) return;
    float in[SPT][3];
    bool ok[SPT];
#pragma unroll
    for (int s = 0; s < SPT; s++) {
        ge_load_coords<3>(inputs, min(t0 + s * GE_BLOCK, B - 1), in[s]);
        ok[s] = !(in[s][0] < 0 || in[s][0] > 1 || in[s][1] < 0 || in[s][1] > 1 || in[s][2] < 0 || in[s][2] > 1);
#pragma unroll
        for (int d = 0; d < 3; d++) in[s][d] = ok[s] ? in[s][d] : 0.5f;
    }
    uint32_t *out = reinterpret_cast<uint32_t *>(outputs);
    for (uint32_t i = 0; i < n_levels; i++) {
        const uint32_t level = lv.order[i];
        const unsigned char *__restrict__ table = reinterpret_cast<const unsigned char *>(grid) + (size_t)lv.offset[level] * 4;
        const uint32_t size = lv.size[level], res = lv.resolution[level];
        const float scale = lv.scale[level];
#pragma unroll
        for (int s = 0; s < SPT; s++) {
            const uint32_t b = t0 + s * GE_BLOCK;
            const uint32_t r = gf_eval_level(in[s], table, size, res, scale, gridtype);
            if (b < B) out[(size_t)level * ostride + b] = ok[s] ? r : 0u;
        }
    }
}

template <typename T, int D, int C>
__global__ void __launch_bounds__(GE_BLOCK) k_grid_bwd(const T *__restrict__ grad, const float *__restrict__ inputs, const GridLevels lv,
                                                       float *__restrict__ grad_grid, uint32_t B, uint32_t n_levels, uint32_t nb,
                                                       uint32_t gridtype, int align_corners, uint32_t interp, int swizzle, float *__restrict__ found_inf) {
    uint32_t level, pb;
    if (!ge_work_item(nb, n_levels, swizzle, lv, level, pb)) return;
    const uint32_t b = pb * GE_BLOCK + threadIdx.x;
    if (b >= B) return;

    float in[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
        in[d] = inputs[(size_t)b * D + d];
        if (in[d] < 0 || in[d] > 1) return;
    }
    using Vec = FeatVec<T, C>;
    const Vec g = reinterpret_cast<const Vec *>(grad)[(size_t)level * B + b];
    float *__restrict__ gtable = grad_grid + (size_t)lv.offset[level] * C;
    const uint32_t hashmap_size = lv.size[level];
    const float scale = lv.scale[level];
    const uint32_t resolution = lv.resolution[level];

    float pos[D];
    uint32_t pos_grid[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
        pos[d] = cn_fma(in[d], scale, align_corners ? 0.0f : 0.5f);
        pos_grid[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)pos_grid[d];
        if (interp == 1) pos[d] = ge_smoothstep(pos[d]);
    }
    float gf[C];
#pragma unroll
    for (int c = 0; c < C; c++) {
        gf[c] = ge_to_float(g.v[c]);
        if (!(fabsf(gf[c]) <= 3.4e38f) && found_inf) *found_inf = 1.0f;             // cnerf_scaler_watch (the atomics below carry the value into the table)
    }
#pragma unroll
    for (int idx = 0; idx < (1 << D); idx++) {
        float w = 1;
        uint32_t pgl[D];
#pragma unroll
        for (int d = 0; d < D; d++) {
            if ((idx & (1 << d)) == 0) { w *= 1 - pos[d]; pgl[d] = pos_grid[d]; }
            else { w *= pos[d]; pgl[d] = pos_grid[d] + 1; }
        }
        const uint32_t index = ge_index<D>(gridtype, align_corners, hashmap_size, resolution, pgl) * C;
#pragma unroll
        for (int c = 0; c < C; c++) unsafeAtomicAdd(&gtable[index + c], w * gf[c]);
    }
}

template <typename T, int D, int C>
__global__ void __launch_bounds__(GE_BLOCK) k_input_bwd(const T *__restrict__ grad, const T *__restrict__ dy_dx, float *__restrict__ grad_inputs,
                                                        uint32_t B, uint32_t L) {
    const uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const T *dd = dy_dx + (size_t)b * L * D * C;
    float result = 0;
    for (uint32_t l = 0; l < L; l++) {
#pragma unroll
        for (int c = 0; c < C; c++)
            result = cn_fma(ge_to_float(grad[(size_t)l * B * C + (size_t)b * C + c]), ge_to_float(dd[l * D * C + d * C + c]), result);
    }
    grad_inputs[t] = result;
}

template <int D, int C>
__global__ void __launch_bounds__(GE_BLOCK) k_grad_tv(const float *__restrict__ inputs, const float *__restrict__ grid, float *__restrict__ grad,
                                                      const GridLevels lv, float weight, uint32_t B, uint32_t n_levels, uint32_t nb,
                                                      uint32_t gridtype, int align_corners) {
    uint32_t level, pb;
    if (!ge_work_item(nb, n_levels, 0, lv, level, pb)) return;
    const uint32_t b = pb * GE_BLOCK + threadIdx.x;
    if (b >= B) return;
    float in[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
        in[d] = inputs[(size_t)b * D + d];
        if (in[d] < 0 || in[d] > 1) return;
    }
    const float *__restrict__ table = grid + (size_t)lv.offset[level] * C;
    float *__restrict__ gtable = grad + (size_t)lv.offset[level] * C;
    const uint32_t hashmap_size = lv.size[level], resolution = lv.resolution[level];
    const float scale = lv.scale[level];
    uint32_t pos_grid[D];
#pragma unroll
    for (int d = 0; d < D; d++) pos_grid[d] = (uint32_t)floorf(cn_fma(in[d], scale, align_corners ? 0.0f : 0.5f));
    float results[C], idelta[C];
#pragma unroll
    for (int c = 0; c < C; c++) { results[c] = 0; idelta[c] = 0; }
    const uint32_t index = ge_index<D>(gridtype, align_corners, hashmap_size, resolution, pos_grid) * C;
    const float w = weight / (2 * D);
#pragma unroll
    for (int d = 0; d < D; d++) {
        const uint32_t cur_d = pos_grid[d];
        if (cur_d < resolution) {
            pos_grid[d] = cur_d + 1;
            const uint32_t ir = ge_index<D>(gridtype, align_corners, hashmap_size, resolution, pos_grid) * C;
#pragma unroll
            for (int c = 0; c < C; c++) {
                const float gv = table[index + c] - table[ir + c];
                results[c] += gv;
                idelta[c] = cn_fma(gv, gv, idelta[c]);
            }
        }
        if (cur_d > 0) {
            pos_grid[d] = cur_d - 1;
            const uint32_t il = ge_index<D>(gridtype, align_corners, hashmap_size, resolution, pos_grid) * C;
#pragma unroll
            for (int c = 0; c < C; c++) {
                const float gv = table[index + c] - table[il + c];
                results[c] += gv;
                idelta[c] = cn_fma(gv, gv, idelta[c]);
            }
        }
        pos_grid[d] = cur_d;
    }
#pragma unroll
    for (int c = 0; c < C; c++) unsafeAtomicAdd(&gtable[index + c], w * results[c] * (1.0f / sqrtf(idelta[c] + 1e-9f)));
}

__global__ void __launch_bounds__(256) k_cast_f32_f16(const float *__restrict__ src, __half *__restrict__ dst, uint64_t n) {
    const uint64_t n8 = n / 8;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        const float4 a = reinterpret_cast<const float4 *>(src)[i * 2], b = reinterpret_cast<const float4 *>(src)[i * 2 + 1];
        union { __half2 h[4]; uint4 u; } o;
        o.h[0] = __floats2half2_rn(a.x, a.y);
        o.h[1] = __floats2half2_rn(a.z, a.w);
        o.h[2] = __floats2half2_rn(b.x, b.y);
        o.h[3] = __floats2half2_rn(b.z, b.w);
        reinterpret_cast<uint4 *>(dst)[i] = o.u;
    }
    for (uint64_t i = n8 * 8 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = __float2half_rn(src[i]);
}

// ------------------------------------------------------------------------------------------------ host side
static int ge_swizzle_default() {
    static const int v = cn_tune_env("CNERF_GRID_SWIZZLE", 2);
    return v;
}
static double ge_dense_weight() {
    static const double w = cn_tune_env_f("CNERF_GRID_DENSE_W", 0.7);   // measured: 0.2 / 0.3 / 0.5 / 0.7 -> 274 / 260 / 243 / 239 us per launch (uniform slices: 247)
    return w;
}

// every level in use must be one of the three modes k_grid_fwd_fast implements (hashed levels additionally 4-entry aligned)
static bool ge_fast_eligible(const GridLevels &lv, uint32_t nl, uint32_t gridtype) {
    static const int on = cn_tune_env("CNERF_GRID_FAST", 1);
    if (!on) return false;
    for (uint32_t l = 0; l < nl; l++) {
        const uint64_t step = (uint64_t)lv.resolution[l] + 1, cells = step * step * step;
        const uint32_t size = lv.size[l];
        const bool pow2 = (size & (size - 1)) == 0;
        if (size >= (1u << 24)) return false;                                   // 32-bit byte offsets
        if (cells <= size) continue;                                           // GE_MODE_DENSE
        if (!pow2 || size < 8) return false;
        if (gridtype == 0 && ((lv.offset[l] | size) & 3u)) return false;        // GE_MODE_HASH2 with aligned 16-byte windows
    }
    return true;
}

// traversal of the specialised forward gather: GE_TRAV_LEVEL = level-major XCD-sliced list (the default), GE_TRAV_SAMPLE = sample-major tiles
// (cnerf_grid_encode_forward_ordered: the caller knows that neighbours of its list are neighbours in space)
#define GE_TRAV_LEVEL 0u
#define GE_TRAV_SAMPLE 1u
#ifndef GE_FAST_SPT
#define GE_FAST_SPT 16                              // samples per thread of the sample-major kernel (tuning builds: CNERF_GRID_SPT)
#endif

template <typename T, int D>
static int ge_fwd_C(const float *inputs, const T *emb, const GridLevels &lv, T *out, uint32_t B, uint32_t C, uint32_t L, uint32_t nl, T *dy_dx,
                    uint32_t gridtype, int ac, uint32_t interp, hipStream_t st, uint32_t ostride, uint32_t traversal = GE_TRAV_LEVEL) {
    uint32_t nb = cn_div_up(B, GE_BLOCK);
    const dim3 block(GE_BLOCK);
    int sw = ge_swizzle_default();
    // The XCD slices (swizzle 2) pay when the one or two levels an XCD works on FIT its 4 MiB L2 (T = 2^19: 2 MiB per level in fp16).  The
    // reference field's own table (T = 2^21: 8 MiB per level) does not: there the plain order — consecutive workgroups, i.e. all eight XCDs, on
    // the same level at the same time, each L2 holding what its neighbours also fetch through the Infinity Cache — is faster: coarse pass
    // 364 -> 283 us, importance pass 229 -> 215 us (profiles/r06_bear_gather_ab.txt).
    if (sw == 2 && cn_tune_env("CNERF_GRID_SWIZZLE", -1) < 0) {
        uint64_t biggest = 0;
        for (uint32_t l = 0; l < nl; l++) biggest = biggest > lv.size[l] ? biggest : lv.size[l];
        if (biggest * C * sizeof(T) > (4ull << 20)) sw = 0;
    }
    GridLevels lvb = lv;
    if constexpr (std::is_same<T, __half>::value && D == 3) {
        if (C == 2 && !dy_dx && interp == 0 && !ac && ge_fast_eligible(lvb, nl, gridtype)) {
            static const int force = cn_tune_env("CNERF_GRID_TRAV", -1);       // tuning builds: override the caller's traversal
            if (force >= 0) traversal = (uint32_t)force;
            if (traversal == GE_TRAV_SAMPLE) {
                static const int spt = cn_tune_env("CNERF_GRID_SPT", GE_FAST_SPT);
#define GE_SM(SPT_) case SPT_: hipLaunchKernelGGL(k_grid_fwd_fast_sm<SPT_>, dim3(cn_div_up(B, GE_BLOCK * SPT_)), block, 0, st, inputs, emb, lvb, out, B, nl, gridtype, ostride); break;
                switch (spt) { GE_SM(4) GE_SM(8) GE_SM(32) default: GE_SM(16) }      // (12 / 20 / 24 measured: 292 / 220 / 256 us against 187 at 16 — tiles that do not align with the 64-sample rays)
#undef GE_SM
                return cn_launch_status();
            }
        }
    }
    dim3 grid(nb * nl);
    if (sw == 2) grid = dim3(CN_NXCD * ge_balance(lvb, nl, nb, D, gridtype, ac != 0, ge_dense_weight()));
    const GridLevels &lv_ = lvb;
    if constexpr (std::is_same<T, __half>::value && D == 3) {
        if (C == 2 && !dy_dx && interp == 0 && !ac && ge_fast_eligible(lv_, nl, gridtype)) {
            hipLaunchKernelGGL(k_grid_fwd_fast, grid, block, 0, st, inputs, emb, lv_, out, B, nl, nb, gridtype, sw, ostride);
            return cn_launch_status();
        }
    }
    switch (C) {
        case 1: hipLaunchKernelGGL((k_grid_fwd<T, D, 1>), grid, block, 0, st, inputs, emb, lv_, out, B, L, nl, nb, dy_dx, gridtype, ac, interp, sw, ostride); break;
        case 2: hipLaunchKernelGGL((k_grid_fwd<T, D, 2>), grid, block, 0, st, inputs, emb, lv_, out, B, L, nl, nb, dy_dx, gridtype, ac, interp, sw, ostride); break;
        case 4: hipLaunchKernelGGL((k_grid_fwd<T, D, 4>), grid, block, 0, st, inputs, emb, lv_, out, B, L, nl, nb, dy_dx, gridtype, ac, interp, sw, ostride); break;
        case 8: hipLaunchKernelGGL((k_grid_fwd<T, D, 8>), grid, block, 0, st, inputs, emb, lv_, out, B, L, nl, nb, dy_dx, gridtype, ac, interp, sw, ostride); break;
        default: return CNERF_EINVAL;
    }
    return cn_launch_status();
}

template <typename T>
static int ge_fwd_D(const float *inputs, const T *emb, const GridLevels &lv, T *out, uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t nl,
                    T *dy_dx, uint32_t gridtype, int ac, uint32_t interp, hipStream_t st, uint32_t ostride, uint32_t traversal = GE_TRAV_LEVEL) {
    switch (D) {
        case 2: return ge_fwd_C<T, 2>(inputs, emb, lv, out, B, C, L, nl, dy_dx, gridtype, ac, interp, st, ostride);
        case 3: return ge_fwd_C<T, 3>(inputs, emb, lv, out, B, C, L, nl, dy_dx, gridtype, ac, interp, st, ostride, traversal);
        case 4: return ge_fwd_C<T, 4>(inputs, emb, lv, out, B, C, L, nl, dy_dx, gridtype, ac, interp, st, ostride);
        case 5: return ge_fwd_C<T, 5>(inputs, emb, lv, out, B, C, L, nl, dy_dx, gridtype, ac, interp, st, ostride);
        default: return CNERF_EINVAL;
    }
}

template <typename T, int D>
static int ge_bwd_C(const T *grad, const float *inputs, const GridLevels &lv, float *gemb, uint32_t B, uint32_t C, uint32_t L, uint32_t nl,
                    const T *dy_dx, float *grad_inputs, uint32_t gridtype, int ac, uint32_t interp, hipStream_t st) {
    const uint32_t nb = cn_div_up(B, GE_BLOCK);
    const dim3 grid(nb * nl), block(GE_BLOCK);
    const int sw = ge_swizzle_default() ? 1 : 0;          // the atomic scatter keeps the plain bijective chunking
    const dim3 gin(cn_div_up(B * D, GE_BLOCK));
    switch (C) {
        case 1:
            hipLaunchKernelGGL((k_grid_bwd<T, D, 1>), grid, block, 0, st, grad, inputs, lv, gemb, B, nl, nb, gridtype, ac, interp, sw, g_cn_found_inf);
            if (dy_dx) hipLaunchKernelGGL((k_input_bwd<T, D, 1>), gin, block, 0, st, grad, dy_dx, grad_inputs, B, L);
            break;
        case 2:
            hipLaunchKernelGGL((k_grid_bwd<T, D, 2>), grid, block, 0, st, grad, inputs, lv, gemb, B, nl, nb, gridtype, ac, interp, sw, g_cn_found_inf);
            if (dy_dx) hipLaunchKernelGGL((k_input_bwd<T, D, 2>), gin, block, 0, st, grad, dy_dx, grad_inputs, B, L);
            break;
        case 4:
            hipLaunchKernelGGL((k_grid_bwd<T, D, 4>), grid, block, 0, st, grad, inputs, lv, gemb, B, nl, nb, gridtype, ac, interp, sw, g_cn_found_inf);
            if (dy_dx) hipLaunchKernelGGL((k_input_bwd<T, D, 4>), gin, block, 0, st, grad, dy_dx, grad_inputs, B, L);
            break;
        case 8:
            hipLaunchKernelGGL((k_grid_bwd<T, D, 8>), grid, block, 0, st, grad, inputs, lv, gemb, B, nl, nb, gridtype, ac, interp, sw, g_cn_found_inf);
            if (dy_dx) hipLaunchKernelGGL((k_input_bwd<T, D, 8>), gin, block, 0, st, grad, dy_dx, grad_inputs, B, L);
            break;
        default: return CNERF_EINVAL;
    }
    return cn_launch_status();
}

template <typename T>
static int ge_bwd_D(const T *grad, const float *inputs, const GridLevels &lv, float *gemb, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                    uint32_t nl, const T *dy_dx, float *grad_inputs, uint32_t gridtype, int ac, uint32_t interp, hipStream_t st) {
    switch (D) {
        case 2: return ge_bwd_C<T, 2>(grad, inputs, lv, gemb, B, C, L, nl, dy_dx, grad_inputs, gridtype, ac, interp, st);
        case 3: return ge_bwd_C<T, 3>(grad, inputs, lv, gemb, B, C, L, nl, dy_dx, grad_inputs, gridtype, ac, interp, st);
        case 4: return ge_bwd_C<T, 4>(grad, inputs, lv, gemb, B, C, L, nl, dy_dx, grad_inputs, gridtype, ac, interp, st);
        case 5: return ge_bwd_C<T, 5>(grad, inputs, lv, gemb, B, C, L, nl, dy_dx, grad_inputs, gridtype, ac, interp, st);
        default: return CNERF_EINVAL;
    }
}

template <int D>
static int ge_tv_C(const float *inputs, const float *emb, float *grad, const GridLevels &lv, float weight, uint32_t B, uint32_t C, uint32_t L,
                   uint32_t gridtype, int ac, hipStream_t st) {
    const uint32_t nb = cn_div_up(B, GE_BLOCK);
    const dim3 grid(nb * L), block(GE_BLOCK);
    switch (C) {
        case 1: hipLaunchKernelGGL((k_grad_tv<D, 1>), grid, block, 0, st, inputs, emb, grad, lv, weight, B, L, nb, gridtype, ac); break;
        case 2: hipLaunchKernelGGL((k_grad_tv<D, 2>), grid, block, 0, st, inputs, emb, grad, lv, weight, B, L, nb, gridtype, ac); break;
        case 4: hipLaunchKernelGGL((k_grad_tv<D, 4>), grid, block, 0, st, inputs, emb, grad, lv, weight, B, L, nb, gridtype, ac); break;
        case 8: hipLaunchKernelGGL((k_grad_tv<D, 8>), grid, block, 0, st, inputs, emb, grad, lv, weight, B, L, nb, gridtype, ac); break;
        default: return CNERF_EINVAL;
    }
    return cn_launch_status();
}

// atomic-free binned scatter (gridencoder_binned.hip)
bool bn_eligible(uint32_t B, uint32_t D, uint32_t C, uint32_t nl, const GridLevels &lv);
uint64_t bn_workspace_bytes(uint32_t B, uint32_t nl, const GridLevels &lv, int dtype);
int bn_backward(const void *grad, const float *inputs, const GridLevels &lv, float *gemb, uint32_t B, uint32_t nl, uint32_t gridtype, int ac,
                uint32_t interp, int dtype, void *workspace, hipStream_t st, bool prepared);
int bn_prepare(const float *inputs, const GridLevels &lv, uint32_t B, uint32_t nl, uint32_t gridtype, int ac, uint32_t interp, int dtype,
               void *workspace, hipStream_t st);
int bn_prepare_rows(const float *inputs, const GridLevels &lv, uint32_t B, uint32_t nl, uint32_t gridtype, int ac, uint32_t interp, int dtype,
                    void *workspace, hipStream_t st, uint32_t row0, uint32_t rows);
int bn_prepare_finish(const GridLevels &lv, uint32_t B, uint32_t nl, int dtype, void *workspace, hipStream_t st);
uint32_t bn_hist_block_points(int dtype);
bool bn_needs_plan(uint32_t B, uint32_t nl, const GridLevels &lv, int dtype, uint32_t gridtype);
void bn_grid_adam_arm(const CnerfGridAdam *cfg);
int bn_grid_adam_consumed();
#define BN_MIN_UPDATES (1u << 20)        // below this many (point, level) pairs the plain atomic kernel is cheaper than five launches

extern "C" {

int cnerf_grid_backward_adam(const CnerfGridAdam *cfg) {
    if (cfg) {
        if (!cfg->p || !cfg->g || !cfg->m || !cfg->v || !cfg->scaler_state) return CNERF_ENULL;
        if (cfg->n == 0 || (cfg->n & 3) || ((((uintptr_t)cfg->p) | ((uintptr_t)cfg->g) | ((uintptr_t)cfg->m) | ((uintptr_t)cfg->v)) & 15) ||
            (((uintptr_t)cfg->p_half) & 7))
            return CNERF_EINVAL;
        if (!(cfg->beta1 >= 0.0f && cfg->beta1 < 1.0f) || !(cfg->beta2 >= 0.0f && cfg->beta2 < 1.0f) || !(cfg->eps >= 0.0f)) return CNERF_EINVAL;
    }
    bn_grid_adam_arm(cfg);
    return CNERF_OK;
}

int cnerf_grid_backward_adam_consumed(int *yes) {
    if (!yes) return CNERF_ENULL;
    *yes = bn_grid_adam_consumed();
    return CNERF_OK;
}

int cnerf_grid_encode_forward_ordered(const float *inputs, const void *embeddings, const int32_t *offsets_host, void *outputs, uint32_t B, uint32_t D,
                                      uint32_t C, uint32_t L, uint32_t max_level, float S, uint32_t H, void *dy_dx, uint32_t gridtype, int align_corners,
                                      uint32_t interp, int dtype, uint32_t out_level_stride, uint32_t traversal, void *stream) {
    if (gridtype > 1 || interp > 1 || traversal > GE_TRAV_SAMPLE) return CNERF_EINVAL;
    GridLevels lv;
    const uint32_t nl = max_level < L ? max_level : L;
    int rc = ge_levels(offsets_host, L, nl, S, H, lv);
    if (rc) return rc;
    if (D < 2 || D > 5 || !(C == 1 || C == 2 || C == 4 || C == 8)) return CNERF_EINVAL;
    if (dtype != CNERF_F32 && dtype != CNERF_F16) return CNERF_EINVAL;
    if (out_level_stride == 0) out_level_stride = B;
    if (out_level_stride < B) return CNERF_EINVAL;
    if (B == 0 || nl == 0) return CNERF_OK;
    if (!inputs || !embeddings || !outputs) return CNERF_ENULL;
    if (dtype == CNERF_F32)
        return ge_fwd_D<float>(inputs, (const float *)embeddings, lv, (float *)outputs, B, D, C, L, nl, (float *)dy_dx, gridtype, align_corners, interp,
                               CN_STREAM(stream), out_level_stride);
    return ge_fwd_D<__half>(inputs, (const __half *)embeddings, lv, (__half *)outputs, B, D, C, L, nl, (__half *)dy_dx, gridtype, align_corners, interp,
                            CN_STREAM(stream), out_level_stride, traversal);
}

int cnerf_grid_encode_forward_strided(const float *inputs, const void *embeddings, const int32_t *offsets_host, void *outputs, uint32_t B, uint32_t D,
                                      uint32_t C, uint32_t L, uint32_t max_level, float S, uint32_t H, void *dy_dx, uint32_t gridtype, int align_corners,
                                      uint32_t interp, int dtype, uint32_t out_level_stride, void *stream) {
    return cnerf_grid_encode_forward_ordered(inputs, embeddings, offsets_host, outputs, B, D, C, L, max_level, S, H, dy_dx, gridtype, align_corners, interp,
                                             dtype, out_level_stride, GE_TRAV_LEVEL, stream);
}

int cnerf_grid_encode_forward(const float *inputs, const void *embeddings, const int32_t *offsets_host, void *outputs, uint32_t B, uint32_t D,
                              uint32_t C, uint32_t L, uint32_t max_level, float S, uint32_t H, void *dy_dx, uint32_t gridtype, int align_corners,
                              uint32_t interp, int dtype, void *stream) {
    return cnerf_grid_encode_forward_strided(inputs, embeddings, offsets_host, outputs, B, D, C, L, max_level, S, H, dy_dx, gridtype, align_corners, interp,
                                             dtype, B, stream);
}

int cnerf_grid_encode_backward_workspace_bytes(const int32_t *offsets_host, uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level,
                                               float S, uint32_t H, int dtype, uint64_t *bytes) {
    if (!bytes) return CNERF_ENULL;
    *bytes = 0;
    GridLevels lv;
    const uint32_t nl = max_level < L ? max_level : L;
    int rc = ge_levels(offsets_host, L, nl, S, H, lv);
    if (rc) return rc;
    if (dtype != CNERF_F32 && dtype != CNERF_F16) return CNERF_EINVAL;
    if ((uint64_t)B * nl >= BN_MIN_UPDATES && bn_eligible(B, D, C, nl, lv)) *bytes = bn_workspace_bytes(B, nl, lv, dtype);
    return CNERF_OK;
}

int cnerf_grid_encode_backward(const void *grad, const float *inputs, const int32_t *offsets_host, float *grad_embeddings, uint32_t B, uint32_t D,
                               uint32_t C, uint32_t L, uint32_t max_level, float S, uint32_t H, const void *dy_dx, float *grad_inputs,
                               uint32_t gridtype, int align_corners, uint32_t interp, int dtype, void *workspace, uint64_t workspace_bytes,
                               void *stream) {
    if (gridtype > 1 || interp > 1) return CNERF_EINVAL;
    GridLevels lv;
    const uint32_t nl = max_level < L ? max_level : L;
    int rc = ge_levels(offsets_host, L, nl, S, H, lv);
    if (rc) return rc;
    if (D < 2 || D > 5 || !(C == 1 || C == 2 || C == 4 || C == 8)) return CNERF_EINVAL;
    if (dtype != CNERF_F32 && dtype != CNERF_F16) return CNERF_EINVAL;
    if (!grad_embeddings) return CNERF_ENULL;
    if (B == 0 || nl == 0) return CNERF_OK;
    if (!grad || !inputs) return CNERF_ENULL;
    if (dy_dx && !grad_inputs) return CNERF_ENULL;
    if (workspace && !dy_dx && (uint64_t)B * nl >= BN_MIN_UPDATES && bn_eligible(B, D, C, nl, lv) &&
        workspace_bytes >= bn_workspace_bytes(B, nl, lv, dtype) && !(((uintptr_t)workspace) & 255))
        return bn_backward(grad, inputs, lv, grad_embeddings, B, nl, gridtype, align_corners, interp, dtype, workspace, CN_STREAM(stream), false);
    if (dtype == CNERF_F32)
        return ge_bwd_D<float>((const float *)grad, inputs, lv, grad_embeddings, B, D, C, L, nl, (const float *)dy_dx, grad_inputs, gridtype, align_corners, interp, CN_STREAM(stream));
    if (dtype == CNERF_F16)
        return ge_bwd_D<__half>((const __half *)grad, inputs, lv, grad_embeddings, B, D, C, L, nl, (const __half *)dy_dx, grad_inputs, gridtype, align_corners, interp, CN_STREAM(stream));
    return CNERF_EINVAL;
}

// The coordinate-only half of the binned backward (histogram + scans), issued ahead of time — typically on a second stream right after
// the samples exist, so that it overlaps the forward and the field backward.  Returns CNERF_OK with *prepared = 1 when the binned path
// applies and the plan now sits in `workspace`; *prepared = 0 (nothing launched) when cnerf_grid_encode_backward would take the atomic
// kernel for this shape.  The matching cnerf_grid_encode_backward_prepared must get the same inputs / shape / workspace.
int cnerf_grid_encode_backward_prepare(const float *inputs, const int32_t *offsets_host, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                       uint32_t max_level, float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp, int dtype,
                                       void *workspace, uint64_t workspace_bytes, int *prepared, void *stream) {
    if (!prepared) return CNERF_ENULL;
    *prepared = 0;
    if (gridtype > 1 || interp > 1) return CNERF_EINVAL;
    GridLevels lv;
    const uint32_t nl = max_level < L ? max_level : L;
    int rc = ge_levels(offsets_host, L, nl, S, H, lv);
    if (rc) return rc;
    if (dtype != CNERF_F32 && dtype != CNERF_F16) return CNERF_EINVAL;
    if (B == 0 || nl == 0) return CNERF_OK;
    if (!inputs) return CNERF_ENULL;
    if (!(workspace && (uint64_t)B * nl >= BN_MIN_UPDATES && bn_eligible(B, D, C, nl, lv) && workspace_bytes >= bn_workspace_bytes(B, nl, lv, dtype) &&
          !(((uintptr_t)workspace) & 255)))
        return CNERF_OK;
    if (!bn_needs_plan(B, nl, lv, dtype, gridtype)) return CNERF_OK;          // (the scatter counts inside its emit kernel: *prepared = 0, nothing launched)
    rc = bn_prepare(inputs, lv, B, nl, gridtype, align_corners, interp, dtype, workspace, CN_STREAM(stream));
    if (rc == 0) *prepared = 1;
    return rc;
}

// cnerf_grid_encode_backward_prepare in pieces: the histogram of rows [row0, row0 + rows) of the B-sample list can be taken as soon as THOSE
// coordinates exist (row0 a multiple of *block_points of cnerf_grid_encode_backward_prepare_block; the range ends on a block border or at B), each
// piece on whatever stream suits the caller; ..._finish runs the scans once every row has been counted and leaves the plan that
// cnerf_grid_encode_backward_prepared consumes.  *prepared = 0 (nothing launched): the shape takes the atomic kernel, or the records are float32.
int cnerf_grid_encode_backward_prepare_block(int dtype, uint32_t *block_points) {
    if (!block_points) return CNERF_ENULL;
    *block_points = bn_hist_block_points(dtype);
    return CNERF_OK;
}

int cnerf_grid_encode_backward_needs_plan(const int32_t *offsets_host, uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level, float S,
                                          uint32_t H, uint32_t gridtype, int dtype, int *needs_plan) {
    if (!needs_plan) return CNERF_ENULL;
    *needs_plan = 0;
    GridLevels lv;
    const uint32_t nl = max_level < L ? max_level : L;
    int rc = ge_levels(offsets_host, L, nl, S, H, lv);
    if (rc) return rc;
    if (dtype != CNERF_F32 && dtype != CNERF_F16) return CNERF_EINVAL;
    if (B && nl && (uint64_t)B * nl >= BN_MIN_UPDATES && bn_eligible(B, D, C, nl, lv) && bn_needs_plan(B, nl, lv, dtype, gridtype)) *needs_plan = 1;
    return CNERF_OK;
}

static int ge_prepare_common(const int32_t *offsets_host, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype,
                             uint32_t interp, int dtype, void *workspace, uint64_t workspace_bytes, GridLevels &lv, bool &go) {
    go = false;
    if (gridtype > 1 || interp > 1) return CNERF_EINVAL;
    int rc = ge_levels(offsets_host, L, L, S, H, lv);
    if (rc) return rc;
    if (dtype != CNERF_F32 && dtype != CNERF_F16) return CNERF_EINVAL;
    go = dtype == CNERF_F16 && B && workspace && (uint64_t)B * L >= BN_MIN_UPDATES && bn_eligible(B, D, C, L, lv) &&
         workspace_bytes >= bn_workspace_bytes(B, L, lv, dtype) && !(((uintptr_t)workspace) & 255) && bn_hist_block_points(dtype) != 0 &&
         bn_needs_plan(B, L, lv, dtype, gridtype);
    return CNERF_OK;
}

int cnerf_grid_encode_backward_prepare_rows(const float *inputs, const int32_t *offsets_host, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                            uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp, int dtype, uint32_t row0, uint32_t rows,
                                            void *workspace, uint64_t workspace_bytes, int *prepared, void *stream) {
    if (!prepared) return CNERF_ENULL;
    *prepared = 0;
    GridLevels lv;
    bool go;
    int rc = ge_prepare_common(offsets_host, B, D, C, L, S, H, gridtype, interp, dtype, workspace, workspace_bytes, lv, go);
    if (rc || !go) return rc;
    if (!inputs) return CNERF_ENULL;
    rc = bn_prepare_rows(inputs, lv, B, L, gridtype, align_corners, interp, dtype, workspace, CN_STREAM(stream), row0, rows);
    if (rc == 0) *prepared = 1;
    return rc;
}

int cnerf_grid_encode_backward_prepare_finish(const int32_t *offsets_host, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                              uint32_t gridtype, uint32_t interp, int dtype, void *workspace, uint64_t workspace_bytes, int *prepared,
                                              void *stream) {
    if (!prepared) return CNERF_ENULL;
    *prepared = 0;
    GridLevels lv;
    bool go;
    int rc = ge_prepare_common(offsets_host, B, D, C, L, S, H, gridtype, interp, dtype, workspace, workspace_bytes, lv, go);
    if (rc || !go) return rc;
    rc = bn_prepare_finish(lv, B, L, dtype, workspace, CN_STREAM(stream));
    if (rc == 0) *prepared = 1;
    return rc;
}

int cnerf_grid_encode_backward_prepared(const void *grad, const float *inputs, const int32_t *offsets_host, float *grad_embeddings, uint32_t B,
                                        uint32_t D, uint32_t C, uint32_t L, uint32_t max_level, float S, uint32_t H, uint32_t gridtype,
                                        int align_corners, uint32_t interp, int dtype, void *workspace, uint64_t workspace_bytes, void *stream) {
    if (gridtype > 1 || interp > 1) return CNERF_EINVAL;
    GridLevels lv;
    const uint32_t nl = max_level < L ? max_level : L;
    int rc = ge_levels(offsets_host, L, nl, S, H, lv);
    if (rc) return rc;
    if (dtype != CNERF_F32 && dtype != CNERF_F16) return CNERF_EINVAL;
    if (!grad_embeddings || !grad || !inputs || !workspace) return CNERF_ENULL;
    if (!((uint64_t)B * nl >= BN_MIN_UPDATES && bn_eligible(B, D, C, nl, lv) && workspace_bytes >= bn_workspace_bytes(B, nl, lv, dtype) &&
          !(((uintptr_t)workspace) & 255)))
        return CNERF_EINVAL;                                  // prepare() would have reported *prepared = 0 for this shape
    return bn_backward(grad, inputs, lv, grad_embeddings, B, nl, gridtype, align_corners, interp, dtype, workspace, CN_STREAM(stream), true);
}

int cnerf_grad_total_variation(const float *inputs, const float *embeddings, float *grad, const int32_t *offsets_host, float weight, uint32_t B,
                               uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners, void *stream) {
    if (!inputs || !embeddings || !grad) return CNERF_ENULL;
    if (gridtype > 1) return CNERF_EINVAL;
    GridLevels lv;
    int rc = ge_levels(offsets_host, L, L, S, H, lv);
    if (rc) return rc;
    for (uint32_t l = 0; l < L; l++) lv.order[l] = (uint8_t)l;
    if (B == 0) return CNERF_OK;
    switch (D) {
        case 2: return ge_tv_C<2>(inputs, embeddings, grad, lv, weight, B, C, L, gridtype, align_corners, CN_STREAM(stream));
        case 3: return ge_tv_C<3>(inputs, embeddings, grad, lv, weight, B, C, L, gridtype, align_corners, CN_STREAM(stream));
        case 4: return ge_tv_C<4>(inputs, embeddings, grad, lv, weight, B, C, L, gridtype, align_corners, CN_STREAM(stream));
        case 5: return ge_tv_C<5>(inputs, embeddings, grad, lv, weight, B, C, L, gridtype, align_corners, CN_STREAM(stream));
        default: return CNERF_EINVAL;
    }
}

int cnerf_cast_f32_to_f16(const float *src, void *dst, uint64_t n, void *stream) {
    if (!src || !dst) return CNERF_ENULL;
    if (n == 0) return CNERF_OK;
    if ((((uintptr_t)src) & 15) || (((uintptr_t)dst) & 15)) return CNERF_EINVAL;
    const uint32_t blocks = (uint32_t)(cn_div_up64(cn_div_up64(n, 8), 256) < 2048 ? cn_div_up64(cn_div_up64(n, 8), 256) : 2048);
    hipLaunchKernelGGL(k_cast_f32_f16, dim3(blocks ? blocks : 1), dim3(256), 0, CN_STREAM(stream), src, (__half *)dst, n);
    return cn_launch_status();
}

}  // extern "C"
