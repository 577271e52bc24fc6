/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path.
 *
 * Plain-C, single-thread CPU restatement of the reference's multiresolution
 * grid encoder CUDA extension (reference: gridencoder/src/gridencoder.cu).
 * The reference cannot be compiled here (no nvcc), so this is its executable
 * form for parity purposes.
 *
 * Conventions:
 *   - -ffp-contract=off; nvcc's default a*b+c contraction written as fmaf().
 *   - `scale = exp2f(level*S)*H - 1` (gridencoder.cu:138) is evaluated with the
 *     host libm exp2f.  The product library evaluates the SAME expression on the
 *     host (also libm) and hands the per-level scale/resolution to the kernel,
 *     so both sides see identical level geometry.
 *   - dtype 0 = float32 embeddings/outputs, 1 = float16 (IEEE binary16, stored
 *     as uint16).  For float16 the reference accumulates in at::Half
 *     (`scalar_t results[C]`, gridencoder.cu:163,186): every corner contribution
 *     is rounded to half and added in half.  That is restated exactly.
 *   - backward: the reference scatters with float / __half2 atomicAdd
 *     (gridencoder.cu:324-337), order-nondeterministic.  The oracle accumulates
 *     sequentially (point order) into a float32 buffer for both dtypes; the
 *     product also accumulates in float32 (documented deviation for the fp16
 *     path: more precise than the reference's half atomics).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* ---- binary16 helpers (gcc 11 has no _Float16 on x86-64) ---- */
static inline float h2f(uint16_t h) {
    const uint32_t s = (uint32_t)(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1f, m = h & 0x3ffu, bits;
    if (e == 0) {
        if (m == 0) bits = s;
        else { /* subnormal */
            int sh = 0;
            while (!(m & 0x400u)) { m <<= 1; sh++; }
            m &= 0x3ffu;
            bits = s | ((uint32_t)(127 - 15 - sh + 1) << 23) | (m << 13);
        }
    } else if (e == 31) bits = s | 0x7f800000u | (m << 13);
    else bits = s | ((e + 112u) << 23) | (m << 13);
    float f; memcpy(&f, &bits, 4); return f;
}
static inline uint16_t f2h(float f) { /* round-to-nearest-even */
    uint32_t x; memcpy(&x, &f, 4);
    const uint32_t s = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(s | 0x7c00u | ((x > 0x7f800000u) ? 0x200u : 0));
    if (x >= 0x477ff000u) return (uint16_t)(s | 0x7c00u); /* rounds to inf (>= 65520) */
    if (x < 0x33000001u) return (uint16_t)s;               /* rounds to zero (<= 2^-25) */
    int e = (int)(x >> 23) - 127;
    uint32_t m = (x & 0x7fffffu) | 0x800000u;
    int shift;
    uint32_t he;
    if (e < -14) { shift = 13 + (-14 - e); he = 0; } else { shift = 13; he = (uint32_t)(e + 15); }
    uint32_t r = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (r & 1))) r++;
    if (he == 0) return (uint16_t)(s | r);             /* subnormal (r may carry into exponent: still right) */
    r += (he << 10) - 0x400u;                           /* remove implicit bit, add exponent; carry handled */
    return (uint16_t)(s | r);
}
void orc_f2h(const float *src, uint16_t *dst, uint32_t n) { for (uint32_t i = 0; i < n; i++) dst[i] = f2h(src[i]); }
void orc_h2f(const uint16_t *src, float *dst, uint32_t n) { for (uint32_t i = 0; i < n; i++) dst[i] = h2f(src[i]); }

#define MAXD 5
#define MAXC 8

/* gridencoder.cu:50-63 */
static inline uint32_t fast_hash(uint32_t D, const uint32_t *pos_grid) {
    static const uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
    uint32_t result = 0;
    for (uint32_t i = 0; i < D; ++i) result ^= pos_grid[i] * primes[i];
    return result;
}

/* gridencoder.cu:66-84 */
static inline uint32_t get_grid_index(uint32_t D, uint32_t C, uint32_t gridtype, int align_corners, uint32_t ch,
                                      uint32_t hashmap_size, uint32_t resolution, const uint32_t *pos_grid) {
    uint32_t stride = 1, index = 0;
    for (uint32_t d = 0; d < D && stride <= hashmap_size; d++) {
        index += pos_grid[d] * stride;
        stride *= align_corners ? resolution : (resolution + 1);
    }
    if (gridtype == 0 && stride > hashmap_size) index = fast_hash(D, pos_grid);
    return (index % hashmap_size) * C + ch;
}

static inline float smoothstep_(float v) { return v * v * (3.0f - 2.0f * v); }             /* :40-42 */
static inline float smoothstep_derivative_(float v) { return 6 * v * (1.0f - v); }         /* :45-47 */

/* level geometry, gridencoder.cu:137-139 */
void orc_level_geometry(uint32_t level, float S, uint32_t H, float *scale, uint32_t *resolution) {
    *scale = exp2f(level * S) * H - 1.0f;
    *resolution = (uint32_t)ceilf(*scale) + 1;
}

static inline float ldg(const void *grid, int dtype, size_t i) {
    return dtype ? h2f(((const uint16_t *)grid)[i]) : ((const float *)grid)[i];
}

/* kernel_grid, gridencoder.cu:87-244.  outputs [L,B,C], dy_dx [B, L*D*C] (may be NULL). */
void orc_grid_encode_forward(const float *inputs, const void *embeddings, const int *offsets, void *outputs,
                             uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level, float S, uint32_t H,
                             void *dy_dx, uint32_t gridtype, int align_corners, uint32_t interp, int dtype) {
    /* levels are independent (disjoint output rows): one OpenMP task per level — bench.py's cpu_baseline leg uses the host's cores;
     * within a level the points are visited in order, so the results do not depend on the thread count */
#pragma omp parallel for schedule(dynamic, 1)
    for (uint32_t level = 0; level < max_level; level++) {
        const size_t goff = (size_t)(uint32_t)offsets[level] * C;
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        float scale; uint32_t resolution;
        orc_level_geometry(level, S, H, &scale, &resolution);
        for (uint32_t b = 0; b < B; b++) {
            const float *in = inputs + (size_t)b * D;
            const size_t ooff = (size_t)level * B * C + (size_t)b * C;
            const size_t doff = (size_t)b * D * L * C + (size_t)level * D * C;
            int oob = 0;
            for (uint32_t d = 0; d < D; d++) if (in[d] < 0 || in[d] > 1) oob = 1;
            if (oob) {
                for (uint32_t ch = 0; ch < C; ch++) { if (dtype) ((uint16_t *)outputs)[ooff + ch] = 0; else ((float *)outputs)[ooff + ch] = 0; }
                if (dy_dx) for (uint32_t i = 0; i < D * C; i++) { if (dtype) ((uint16_t *)dy_dx)[doff + i] = 0; else ((float *)dy_dx)[doff + i] = 0; }
                continue;
            }
            float pos[MAXD], pos_deriv[MAXD];
            uint32_t pos_grid[MAXD];
            for (uint32_t d = 0; d < D; d++) {
                pos[d] = fmaf(in[d], scale, align_corners ? 0.0f : 0.5f);
                pos_grid[d] = (uint32_t)floorf(pos[d]);
                pos[d] -= (float)pos_grid[d];
                if (interp == 1) { pos_deriv[d] = smoothstep_derivative_(pos[d]); pos[d] = smoothstep_(pos[d]); }
                else pos_deriv[d] = 1.0f;
            }
            float res_f[MAXC] = {0};
            uint16_t res_h[MAXC] = {0};
            for (uint32_t idx = 0; idx < (1u << D); idx++) {
                float w = 1;
                uint32_t pgl[MAXD];
                for (uint32_t d = 0; d < D; d++) {
                    if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pgl[d] = pos_grid[d]; }
                    else { w *= pos[d]; pgl[d] = pos_grid[d] + 1; }
                }
                const uint32_t index = get_grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pgl);
                for (uint32_t ch = 0; ch < C; ch++) {
                    if (dtype) res_h[ch] = f2h(h2f(res_h[ch]) + h2f(f2h(w * ldg(embeddings, 1, goff + index + ch))));
                    else res_f[ch] = fmaf(w, ldg(embeddings, 0, goff + index + ch), res_f[ch]);
                }
            }
            for (uint32_t ch = 0; ch < C; ch++) { if (dtype) ((uint16_t *)outputs)[ooff + ch] = res_h[ch]; else ((float *)outputs)[ooff + ch] = res_f[ch]; }

            if (dy_dx) {
                for (uint32_t gd = 0; gd < D; gd++) {
                    float rg_f[MAXC] = {0};
                    uint16_t rg_h[MAXC] = {0};
                    for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
                        float w = scale;
                        uint32_t pgl[MAXD];
                        for (uint32_t nd = 0; nd < D - 1; nd++) {
                            const uint32_t d = (nd >= gd) ? (nd + 1) : nd;
                            if ((idx & (1u << nd)) == 0) { w *= 1 - pos[d]; pgl[d] = pos_grid[d]; }
                            else { w *= pos[d]; pgl[d] = pos_grid[d] + 1; }
                        }
                        pgl[gd] = pos_grid[gd];
                        const uint32_t il = get_grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pgl);
                        pgl[gd] = pos_grid[gd] + 1;
                        const uint32_t ir = get_grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pgl);
                        for (uint32_t ch = 0; ch < C; ch++) {
                            if (dtype) {
                                const float diff = h2f(f2h(ldg(embeddings, 1, goff + ir + ch) - ldg(embeddings, 1, goff + il + ch)));
                                rg_h[ch] = f2h(h2f(rg_h[ch]) + h2f(f2h(w * diff * pos_deriv[gd])));
                            } else {
                                const float diff = ldg(embeddings, 0, goff + ir + ch) - ldg(embeddings, 0, goff + il + ch);
                                rg_f[ch] = fmaf(w * diff, pos_deriv[gd], rg_f[ch]);
                            }
                        }
                    }
                    for (uint32_t ch = 0; ch < C; ch++) {
                        if (dtype) ((uint16_t *)dy_dx)[doff + gd * C + ch] = rg_h[ch];
                        else ((float *)dy_dx)[doff + gd * C + ch] = rg_f[ch];
                    }
                }
            }
        }
    }
}

/* kernel_grid_backward, gridencoder.cu:247-339 (+ kernel_input_backward :342-368).
 * grad: [L,B,C] (dtype), grad_embeddings: float32 [sum T, C] pre-zeroed by the caller (grid.py:83),
 * dy_dx [B, L*D*C] (dtype, may be NULL), grad_inputs float32 [B,D] (may be NULL). */
void orc_grid_encode_backward(const void *grad, const float *inputs, const int *offsets, float *grad_embeddings,
                              uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level, float S, uint32_t H,
                              const void *dy_dx, float *grad_inputs,
                              uint32_t gridtype, int align_corners, uint32_t interp, int dtype) {
    /* one OpenMP task per level: a level's updates land in its own slice of grad_embeddings and are applied in point order (the
     * sequential sum order of this restatement is kept whatever the thread count) */
#pragma omp parallel for schedule(dynamic, 1)
    for (uint32_t level = 0; level < max_level; level++) {
        const size_t goff = (size_t)(uint32_t)offsets[level] * C;
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        float scale; uint32_t resolution;
        orc_level_geometry(level, S, H, &scale, &resolution);
        for (uint32_t b = 0; b < B; b++) {
            const float *in = inputs + (size_t)b * D;
            int oob = 0;
            for (uint32_t d = 0; d < D; d++) if (in[d] < 0 || in[d] > 1) oob = 1;
            if (oob) continue;
            float pos[MAXD];
            uint32_t pos_grid[MAXD];
            for (uint32_t d = 0; d < D; d++) {
                pos[d] = fmaf(in[d], scale, align_corners ? 0.0f : 0.5f);
                pos_grid[d] = (uint32_t)floorf(pos[d]);
                pos[d] -= (float)pos_grid[d];
                if (interp == 1) pos[d] = smoothstep_(pos[d]);
            }
            float g[MAXC];
            for (uint32_t c = 0; c < C; c++) g[c] = ldg(grad, dtype, (size_t)level * B * C + (size_t)b * C + c);
            for (uint32_t idx = 0; idx < (1u << D); idx++) {
                float w = 1;
                uint32_t pgl[MAXD];
                for (uint32_t d = 0; d < D; d++) {
                    if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pgl[d] = pos_grid[d]; }
                    else { w *= pos[d]; pgl[d] = pos_grid[d] + 1; }
                }
                const uint32_t index = get_grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pgl);
                for (uint32_t c = 0; c < C; c++) grad_embeddings[goff + index + c] += w * g[c];
            }
        }
    }
    if (dy_dx && grad_inputs) {
        for (uint32_t t = 0; t < B * D; t++) {
            const uint32_t b = t / D, d = t - b * D;
            float result = 0;
            for (uint32_t l = 0; l < L; l++)
                for (uint32_t ch = 0; ch < C; ch++)
                    result = fmaf(ldg(grad, dtype, (size_t)l * B * C + (size_t)b * C + ch),
                                  ldg(dy_dx, dtype, (size_t)b * L * D * C + (size_t)l * D * C + d * C + ch), result);
            grad_inputs[t] = result;
        }
    }
}

/* kernel_grad_tv, gridencoder.cu:505-609.  float32 only (grid.py:171 forces autocast off).
 * grad is accumulated in place (sequential point order). */
void orc_grad_total_variation(const float *inputs, const float *embeddings, float *grad, const int *offsets,
                              float weight, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                              uint32_t gridtype, int align_corners) {
    for (uint32_t level = 0; level < L; level++) {
        const size_t goff = (size_t)(uint32_t)offsets[level] * C;
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        float scale; uint32_t resolution;
        orc_level_geometry(level, S, H, &scale, &resolution);
        for (uint32_t b = 0; b < B; b++) {
            const float *in = inputs + (size_t)b * D;
            int oob = 0;
            for (uint32_t d = 0; d < D; d++) if (in[d] < 0 || in[d] > 1) oob = 1;
            if (oob) continue;
            uint32_t pos_grid[MAXD];
            for (uint32_t d = 0; d < D; d++) pos_grid[d] = (uint32_t)floorf(fmaf(in[d], scale, align_corners ? 0.0f : 0.5f));
            float results[MAXC] = {0}, idelta[MAXC] = {0};
            const uint32_t index = get_grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pos_grid);
            const float w = weight / (2 * D);
            for (uint32_t d = 0; d < D; d++) {
                const uint32_t cur_d = pos_grid[d];
                if (cur_d < resolution) {
                    pos_grid[d] = cur_d + 1;
                    const uint32_t ir = get_grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pos_grid);
                    for (uint32_t ch = 0; ch < C; ch++) {
                        const float gv = embeddings[goff + index + ch] - embeddings[goff + ir + ch];
                        results[ch] += gv; idelta[ch] = fmaf(gv, gv, idelta[ch]);
                    }
                }
                if (cur_d > 0) {
                    pos_grid[d] = cur_d - 1;
                    const uint32_t il = get_grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pos_grid);
                    for (uint32_t ch = 0; ch < C; ch++) {
                        const float gv = embeddings[goff + index + ch] - embeddings[goff + il + ch];
                        results[ch] += gv; idelta[ch] = fmaf(gv, gv, idelta[ch]);
                    }
                }
                pos_grid[d] = cur_d;
            }
            for (uint32_t ch = 0; ch < C; ch++)
                grad[goff + index + ch] += w * results[ch] * (1.0f / sqrtf(idelta[ch] + 1e-9f));
        }
    }
}
