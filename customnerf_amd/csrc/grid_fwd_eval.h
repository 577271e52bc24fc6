// The specialised forward gather of one (sample, level): fp16 table, D = 3, C = 2, linear interpolation, align_corners = false — the
// configuration CustomNeRF runs.  Shared by k_grid_fwd_fast (gridencoder.hip) and the fused gather + field kernel (field_fused_fwd.hip);
// bit-identical to the generic k_grid_fwd and to oracle/gridencoder_ref.c (same operations on the same values in the same order).
#pragma once
#include "grid_common.h"

__device__ __forceinline__ float gf_mix_lo(float w, uint32_t g2, float negzero) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,1,0]" : "=v"(r) : "v"(w), "v"(g2), "v"(negzero));
    return r;
}
__device__ __forceinline__ float gf_mix_hi(float w, uint32_t g2, float negzero) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(w), "v"(g2), "v"(negzero));
    return r;
}
// acc (half2) += half2(w * g.x, w * g.y), each product rounded to half, then the packed half add — ge_accum2's arithmetic
__device__ __forceinline__ void gf_accum(cn_gf_h2 &acc, float w, uint32_t g2, float negzero) {
    const float p0 = gf_mix_lo(w, g2, negzero), p1 = gf_mix_hi(w, g2, negzero);
    const cn_gf_h2 p = {(_Float16)p0, (_Float16)p1};
    acc = acc + p;
}
__device__ __forceinline__ uint32_t gf_ld1(const unsigned char *__restrict__ base, uint32_t byte_off) { return *reinterpret_cast<const uint32_t *>(base + byte_off); }
struct __attribute__((packed, aligned(4))) gf_u2 { uint32_t x, y; };        // read at 4-byte-aligned entry addresses (still one global_load_dwordx2)
struct alignas(16) gf_u4 { uint32_t x, y, z, w; };

// `in` inside [0, 1]^3 (the caller handles out-of-range samples: zero features); `table` = first byte of the level's entries (4 B per entry);
// -> the two half features packed in one dword
__device__ __forceinline__ uint32_t gf_eval_level(const float (&in)[3], const unsigned char *__restrict__ table, uint32_t size, uint32_t resolution,
                                                  float scale, uint32_t gridtype) {
    float fr[3], om[3];
    uint32_t pg[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const float pos = cn_fma(in[d], scale, 0.5f);
        pg[d] = (uint32_t)floorf(pos);
        fr[d] = pos - (float)pg[d];
        om[d] = 1 - fr[d];
    }
    // weights of the four (y, z) rows for x and x + 1, reference order: x factor first, then y, then z
    const float a00 = om[0] * om[1], a10 = fr[0] * om[1], a01 = om[0] * fr[1], a11 = fr[0] * fr[1];
    const float w[8] = {a00 * om[2], a10 * om[2], a01 * om[2], a11 * om[2], a00 * fr[2], a10 * fr[2], a01 * fr[2], a11 * fr[2]};
    uint32_t c[8];                                            // corner entries (half2 bit patterns), corner index = x + 2 y + 4 z
    const int mode = ge_level_mode<3>(gridtype, false, size, resolution);         // level-uniform (workgroup-uniform in k_grid_fwd_fast)
    if (mode == GE_MODE_DENSE) {
        // index = x + y s + z s^2 < size, x + 1 <= resolution: the two x corners are neighbours -> one 8-byte load per (y, z) row
        const uint32_t s1 = resolution + 1, s2 = s1 * s1;
        const uint32_t i00 = (pg[0] + pg[1] * s1 + pg[2] * s2) * 4u;
        const gf_u2 r0 = *reinterpret_cast<const gf_u2 *>(table + i00);
        const gf_u2 r1 = *reinterpret_cast<const gf_u2 *>(table + (i00 + s1 * 4u));
        const gf_u2 r2 = *reinterpret_cast<const gf_u2 *>(table + (i00 + s2 * 4u));
        const gf_u2 r3 = *reinterpret_cast<const gf_u2 *>(table + (i00 + (s1 + s2) * 4u));
        c[0] = r0.x; c[1] = r0.y; c[2] = r1.x; c[3] = r1.y; c[4] = r2.x; c[5] = r2.y; c[6] = r3.x; c[7] = r3.y;
    } else if (mode == GE_MODE_HASH2) {
        const uint32_t mask = size - 1;
        const uint32_t hy0 = pg[1] * 2654435761u, hy1 = hy0 + 2654435761u, hz0 = pg[2] * 805459861u, hz1 = hz0 + 805459861u;
        const uint32_t x0 = pg[0], xm = x0 ^ (x0 + 1);                         // i1 = i0 ^ (xm & mask): the trailing-ones run of x0, plus one bit
        const bool in_quad = (xm & mask) < 4u;                                  // x0 != 3 mod 4 (or a table of < 4 entries, which ge_levels rules out)
        const bool odd = (x0 & 1u) != 0;                                        // partner = entry ^ 3 instead of entry ^ 1
        const uint32_t hyz[4] = {hy0 ^ hz0, hy1 ^ hz0, hy0 ^ hz1, hy1 ^ hz1};
        uint32_t i0[4];
        gf_u4 v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {                                           // all four window loads first
            i0[q] = (x0 ^ hyz[q]) & mask;
            v[q] = *reinterpret_cast<const gf_u4 *>(table + ((i0[q] & ~3u) * 4u));
        }
        uint32_t far[4] = {0u, 0u, 0u, 0u};
        if (!in_quad) {                                                         // one lane-masked block for the x = 3 mod 4 lanes
#pragma unroll
            for (int q = 0; q < 4; q++) far[q] = gf_ld1(table, ((i0[q] ^ xm) & mask) * 4u);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const bool b0 = (i0[q] & 1u) != 0, b1 = (i0[q] & 2u) != 0;
            const uint32_t plo = b1 ? v[q].z : v[q].x, phi = b1 ? v[q].w : v[q].y;      // the aligned pair holding entry i0
            c[2 * q] = b0 ? phi : plo;
            const uint32_t same = b0 ? plo : phi;                               // entry i0 ^ 1
            const uint32_t olo = b1 ? v[q].x : v[q].z, ohi = b1 ? v[q].y : v[q].w;      // the other pair
            const uint32_t cross = b0 ? olo : ohi;                              // entry i0 ^ 3
            const uint32_t near = odd ? cross : same;
            c[2 * q + 1] = in_quad ? near : far[q];
        }
    } else {
        // GE_MODE_TILED2: index = (x + y s + z s^2) & mask; the x + 1 corner is the next entry modulo the table size
        // (a dimension enters the strided sum only while the running stride still fits the table: ge_index / gridencoder.cu:66-84)
        const uint32_t mask = size - 1, step = resolution + 1;
        const uint32_t s1 = step <= size ? step : 0u;
        const uint32_t s2 = (s1 && step * step <= size) ? step * step : 0u;
        const uint32_t lin = pg[0] + pg[1] * s1 + pg[2] * s2;
        const uint32_t row[4] = {lin, lin + s1, lin + s2, lin + s1 + s2};
        // unaligned 8-byte load of (i0, i0 + 1) except at the wrap (i0 = size - 1: the pair would leave the level): there the window starts
        // one entry earlier and the partner, entry 0 of the level, comes from a lane-masked load
        gf_u2 r[4];
        uint32_t i0[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            i0[q] = row[q] & mask;
            r[q] = *reinterpret_cast<const gf_u2 *>(table + (i0[q] == mask ? i0[q] - 1 : i0[q]) * 4u);
        }
        bool any_wrap = false;
#pragma unroll
        for (int q = 0; q < 4; q++) any_wrap = any_wrap || i0[q] == mask;
        uint32_t first = 0u;
        if (any_wrap) first = gf_ld1(table, 0u);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const bool wrap = i0[q] == mask;
            c[2 * q] = wrap ? r[q].y : r[q].x;
            c[2 * q + 1] = wrap ? first : r[q].y;
        }
    }
    const float negzero = -0.0f;
    cn_gf_h2 acc = {(_Float16)0, (_Float16)0};
#pragma unroll
    for (int k = 0; k < 8; k++) gf_accum(acc, w[k], c[k], negzero);
    return __builtin_bit_cast(uint32_t, acc);
}
