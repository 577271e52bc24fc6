// Fused field backward, single-launch fp16 form (gfx950).
//
// Same activation-gradient chain as k_field_bwd_data, but the weight-gradient GEMMs dW_l = dz_l . a_{l-1}^T never leave the
// chip: after a workgroup's four waves have run the chain for their 4 x 32 samples, the operands are written to a 64 KiB LDS
// staging area as [row][128 samples] (16-byte chunks XOR-swizzled by row so that both the 2-byte column writes and the
// 16-byte fragment reads are conflict-free), and the 24 32x32 weight-gradient tiles — distributed over the four waves, their
// fp32 accumulators persistent in registers for the whole launch — are advanced by 8 MFMA K-steps each.  Four staging
// phases per 128 samples (LDS: 92 KiB weight fragments + 64 KiB staging).  At the end every workgroup writes its partial
// tiles to a [workgroup][parameter] scratch (plain coalesced stores) and a small kernel sums them into the gradients:
// no spill of activations (1.4 KiB/sample x 2 in the two-launch form) and no float atomics.
#include "field_bwd_common.h"

#define FF_ROWB 256                        // bytes per staged row (128 samples x 2 B)
#define FF_ROWS 256                        // staged rows
#define FF_STAGE_BYTES (FF_ROWB * FF_ROWS)
// Re-materialise the staging coordinates right where they are used: the scheduler otherwise computes all ~250 staging
// addresses of an iteration up front and spills them (2 KiB of scratch per lane).
#define FF_PIN(col, hi) asm volatile("" : "+v"(col), "+v"(hi)::"memory")

__device__ __forceinline__ void ff_put(unsigned char *st, uint32_t row, uint32_t col, _Float16 v) {
    *reinterpret_cast<_Float16 *>(st + row * FF_ROWB + (((col >> 3) ^ (row & 15)) << 4) + ((col & 7) << 1)) = v;
}
__device__ __forceinline__ cn_h8 ff_frag(const unsigned char *st, uint32_t row, uint32_t s, uint32_t hi) {
    return *reinterpret_cast<const cn_h8 *>(st + row * FF_ROWB + ((((2 * s + hi)) ^ (row & 15)) << 4));
}
// 64 rows held as C-ordered B fragments -> staging rows row0 + feature, column col
__device__ __forceinline__ void ff_stage_clayout(unsigned char *st, uint32_t row0, uint32_t col, uint32_t hi, const cn_h8 *b) {
#pragma unroll
    for (int s = 0; s < 4; s++) {
        asm volatile("" : "+v"(col), "+v"(hi)::"memory");
#pragma unroll
        for (int j = 0; j < 8; j++) ff_put(st, row0 + (uint32_t)fld_col_clayout<true>(s, hi, j), col, b[s][j]);
    }
}
template <int NS>
__device__ __forceinline__ void ff_stage_natural(unsigned char *st, uint32_t row0, uint32_t col, uint32_t hi, const cn_h8 *b) {
#pragma unroll
    for (int s = 0; s < NS; s++) {
        asm volatile("" : "+v"(col), "+v"(hi)::"memory");
#pragma unroll
        for (int j = 0; j < 8; j++) ff_put(st, row0 + (uint32_t)fld_col_natural<true>(s, hi, j), col, b[s][j]);
    }
}

struct FfTile {
    uint32_t zrow, M, arow, N, mt, nt, dst, stride, col0;
    bool active;
};

// advance one persistent dW tile by the 128 staged samples (8 K-steps)
__device__ __forceinline__ void ff_tile_mma(const unsigned char *st, const FfTile &t, uint32_t li, uint32_t hi, cn_f16v &acc) {
    if (!t.active) return;                                  // wave-uniform
    asm volatile("" : "+v"(li), "+v"(hi)::"memory");
    const uint32_t zr = 32 * t.mt + li, ar = 32 * t.nt + li;
    const bool zok = zr < t.M, aok = ar < t.N;
#pragma unroll
    for (int s = 0; s < 8; s++) {
        const cn_h8 a = zok ? ff_frag(st, t.zrow + zr, s, hi) : Prec<true>::zero();
        const cn_h8 b = aok ? ff_frag(st, t.arow + ar, s, hi) : Prec<true>::zero();
        acc = Prec<true>::mfma(a, b, acc);
    }
}

__device__ __forceinline__ void ff_tile_store(float *__restrict__ part, const FfTile &t, uint32_t li, uint32_t hi, const cn_f16v &acc) {
    if (!t.active) return;
    const uint32_t col = 32 * t.nt + li;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const uint32_t row = 32 * t.mt + (uint32_t)fld_rho(r, hi);
        if (row < t.M && col < t.N) part[t.dst + (size_t)row * t.stride + t.col0 + col] = acc[r];
    }
}

// ---- tile tables of the four phases (idx = slot * 4 + wave).  Offsets into the flat [net | den | rgb] parameter space.
struct FfOff {
    uint32_t n0, n1, n2, d0, dO, r0, rO, total;
};
__host__ __device__ __forceinline__ FfOff ff_offsets(const FieldDims &dm) {
    FfOff o;
    uint32_t p = 0;
    o.n0 = p; p += FLD_HID * dm.enc_pad;
    o.n1 = p; p += (dm.n_hidden_geo == 2) ? 4096 : 0;
    o.n2 = p; p += 4096;
    o.d0 = p; p += 4096;
    o.dO = p; p += 16 * 64;
    o.r0 = p; p += 64 * 96;
    o.rO = p; p += 16 * 64;
    o.total = p;
    return o;
}

// phase A staging rows: zro 0..7 | hr 8..71 | zr 72..135 | fea 136..199 | dir 200..231
__device__ __forceinline__ FfTile ff_tile_A(uint32_t i, const FieldDims &dm, const FfOff &o) {
    FfTile t;
    t.active = i < 8;
    if (i < 2) { t.zrow = 0; t.M = dm.n_rgb_out; t.arow = 8; t.N = 64; t.mt = 0; t.nt = i; t.dst = o.rO; t.stride = 64; t.col0 = 0; }
    else if (i < 4) { t.zrow = 72; t.M = 64; t.arow = 200; t.N = FLD_NDIR; t.mt = i - 2; t.nt = 0; t.dst = o.r0; t.stride = 96; t.col0 = 0; }
    else { const uint32_t k = i - 4; t.zrow = 72; t.M = 64; t.arow = 136; t.N = 64; t.mt = k >> 1; t.nt = k & 1; t.dst = o.r0; t.stride = 96; t.col0 = FLD_NDIR; }
    return t;
}
// phase B: zdo 0..7 | hd 8..71 | zd 72..135 | (fea still at 136..199)
__device__ __forceinline__ FfTile ff_tile_B(uint32_t i, const FieldDims &dm, const FfOff &o) {
    FfTile t;
    t.active = i < 6;
    if (i < 2) { t.zrow = 0; t.M = 1; t.arow = 8; t.N = 64; t.mt = 0; t.nt = i; t.dst = o.dO; t.stride = 64; t.col0 = 0; }
    else { const uint32_t k = i - 2; t.zrow = 72; t.M = 64; t.arow = 136; t.N = 64; t.mt = k >> 1; t.nt = k & 1; t.dst = o.d0; t.stride = 64; t.col0 = 0; }
    return t;
}
// phase C: z3 0..63 | h_last 64..127 | z2 128..191 | h1 192..255
__device__ __forceinline__ FfTile ff_tile_C(uint32_t i, const FieldDims &dm, const FfOff &o) {
    FfTile t;
    t.active = i < (dm.n_hidden_geo == 2 ? 8u : 4u);
    if (i < 4) { t.zrow = 0; t.M = 64; t.arow = 64; t.N = 64; t.mt = i >> 1; t.nt = i & 1; t.dst = o.n2; t.stride = 64; t.col0 = 0; }
    else { const uint32_t k = i - 4; t.zrow = 128; t.M = 64; t.arow = 192; t.N = 64; t.mt = k >> 1; t.nt = k & 1; t.dst = o.n1; t.stride = 64; t.col0 = 0; }
    return t;
}
// phase D: z1 0..63 | enc 64..64+enc_pad
__device__ __forceinline__ FfTile ff_tile_D(uint32_t i, const FieldDims &dm, const FfOff &o) {
    FfTile t;
    const uint32_t ntn = (dm.enc_pad + 31) / 32;
    t.active = i < 2 * ntn;
    t.zrow = 0; t.M = 64; t.arow = 64; t.N = dm.enc_pad; t.mt = i / ntn; t.nt = i % ntn; t.dst = o.n0; t.stride = dm.enc_pad; t.col0 = 0;
    return t;
}

template <int SENC, int NGEO, int TENC>
__global__ void __launch_bounds__(FLD_THREADS) k_field_bwd_fused(const void *__restrict__ enc, const float *__restrict__ xyz, const float *__restrict__ dirs,
                                                                 uint32_t dir_group, uint32_t P_, FieldDims dm, const float *__restrict__ pnet,
                                                                 const float *__restrict__ pden, const float *__restrict__ prgb,
                                                                 const float *__restrict__ g_sigma, const float *__restrict__ g_rgbc,
                                                                 void *__restrict__ grad_enc, float *__restrict__ partials) {
    constexpr bool H = true;
    using PR = Prec<H>;
    using frag_t = typename PR::frag_t;
    using elem_t = typename PR::elem_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char fld_lds[];
    elem_t *wl = reinterpret_cast<elem_t *>(fld_lds);
    const FieldLds lo = fld_lds_layout<H>(dm);
    const FieldLdsT lt = fb_ldsT_layout<H>(dm);
    elem_t *wt = wl + lo.off[7];
    unsigned char *st = reinterpret_cast<unsigned char *>(wt + lt.off[7]);          // 64 KiB staging
    const FfOff po = ff_offsets(dm);

    constexpr uint32_t S64 = FLD_HID / PR::KS, SDIR = FLD_DIR / PR::KS, S32 = 32 / PR::KS, SR0 = S64 + SDIR;
    const uint32_t in_r0 = FLD_HID + FLD_DIR;
    const float *n0 = pnet, *n1 = pnet + FLD_HID * dm.enc_pad;
    const float *n2 = n1 + (NGEO == 2 ? FLD_HID * FLD_HID : 0);
    const float *d0 = pden, *dO = pden + FLD_HID * FLD_HID;
    const float *r0 = prgb, *rO = prgb + FLD_HID * in_r0;

    fb_stage_layer<H, 0>(wl + lo.off[0], n0, FLD_HID, dm.enc_pad, 2, SENC, dm.enc_pad);
    if (NGEO == 2) fb_stage_layer<H, 1>(wl + lo.off[1], n1, FLD_HID, FLD_HID, 2, S64, FLD_HID);
    fb_stage_layer<H, 1>(wl + lo.off[2], n2, FLD_HID, FLD_HID, 2, S64, FLD_HID);
    fb_stage_layer<H, 1>(wl + lo.off[3], d0, FLD_HID, FLD_HID, 2, S64, FLD_HID);
    fb_stage_layer<H, 1>(wl + lo.off[4], dO, 16, FLD_HID, 1, S64, FLD_HID);
    fb_stage_layer<H, 2>(wl + lo.off[5], r0, FLD_HID, in_r0, 2, SR0, in_r0);
    fb_stage_layer<H, 1>(wl + lo.off[6], rO, 16, FLD_HID, 1, S64, FLD_HID);
    fb_stage_layer_T<H>(wt + lt.off[0], n0, FLD_HID, dm.enc_pad, 0, dm.enc_pad, TENC, S64);
    if (NGEO == 2) fb_stage_layer_T<H>(wt + lt.off[1], n1, FLD_HID, FLD_HID, 0, FLD_HID, 2, S64);
    fb_stage_layer_T<H>(wt + lt.off[2], n2, FLD_HID, FLD_HID, 0, FLD_HID, 2, S64);
    fb_stage_layer_T<H>(wt + lt.off[3], d0, FLD_HID, FLD_HID, 0, FLD_HID, 2, S64);
    fb_stage_layer_T<H>(wt + lt.off[4], dO, 16, FLD_HID, 0, FLD_HID, 2, S32);
    fb_stage_layer_T<H>(wt + lt.off[5], r0, FLD_HID, in_r0, FLD_NDIR, FLD_HID, 2, S64);
    fb_stage_layer_T<H>(wt + lt.off[6], rO, 16, FLD_HID, 0, FLD_HID, 2, S32);
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63, li = lane & 31;
    uint32_t hi = lane >> 5;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);        // provably wave-uniform: tile descriptors live in SGPRs
    // this wave's persistent dW tiles: A0/A1, B0/B1, C0/C1, D0 (descriptors are re-derived where used, they are a few SALU ops)
    cn_f16v wA0, wA1, wB0, wX, wC0, wC1;      // wX: phase-B tiles 4,5 on waves 2,3 and phase-D tiles 0,1 on waves 0,1
#pragma unroll
    for (int r = 0; r < 16; r++) { wA0[r] = 0; wA1[r] = 0; wB0[r] = 0; wX[r] = 0; wC0[r] = 0; wC1[r] = 0; }

    const uint32_t n_tiles = (P_ + FLD_TILE - 1) / FLD_TILE;
    for (uint32_t tile0 = blockIdx.x * FLD_WAVES; tile0 < n_tiles; tile0 += gridDim.x * FLD_WAVES) {        // workgroup-uniform trip count
        asm volatile("" ::: "memory");
        const uint32_t tile = tile0 + wave;
        const uint32_t p = tile * FLD_TILE + li;
        const bool valid = p < P_;
        uint32_t col = wave * 32 + li;                                      // this lane's column in the staging area
        // opaque to the optimiser: otherwise the ~250 loop-invariant staging addresses derived from it are hoisted out of the
        // persistent loop and pinned in VGPRs (kilobytes of scratch spills)
        asm volatile("" : "+v"(col));

        // ================= forward recompute
        frag_t x0[SENC];
        fb_load_enc<H, SENC>(enc, P_, dm.L, p, valid, hi, x0);
        cn_f16v acc[2];
        frag_t h1[4], h2[4], fea[4], hd[4], hr[4];
        fb_zero(acc);
        fb_gemm<H, 2, SENC>(wl + lo.off[0], SENC, 0, x0, lane, acc);
        fb_c_to_b<H, true>(acc, h1);
        if (NGEO == 2) {
            fb_zero(acc);
            fb_gemm<H, 2, S64>(wl + lo.off[1], S64, 0, h1, lane, acc);
            fb_c_to_b<H, true>(acc, h2);
        }
        const frag_t *hlast = (NGEO == 2) ? h2 : h1;
        fb_zero(acc);
        fb_gemm<H, 2, S64>(wl + lo.off[2], S64, 0, hlast, lane, acc);
        fb_c_to_b<H, false>(acc, fea);
        fb_zero(acc);
        fb_gemm<H, 2, S64>(wl + lo.off[3], S64, 0, fea, lane, acc);
        fb_c_to_b<H, true>(acc, hd);
        cn_f16v out[1];
        fb_zero(out);
        fb_gemm<H, 1, S64>(wl + lo.off[4], S64, 0, hd, lane, out);
        const float raw = (float)(_Float16)out[0][0];
        frag_t dfr[SDIR];
        fb_dir_frags<H>(dirs, dir_group, p, valid, hi, dfr);
        fb_zero(acc);
        fb_gemm<H, 2, S64>(wl + lo.off[5], SR0, 0, fea, lane, acc);
        fb_gemm<H, 2, SDIR>(wl + lo.off[5], SR0, S64, dfr, lane, acc);
        fb_c_to_b<H, true>(acc, hr);
        fb_zero(out);
        fb_gemm<H, 1, S64>(wl + lo.off[6], S64, 0, hr, lane, out);

        // ================= output-layer gradients
        frag_t bro[1], bdo[1];
        {
            cn_h8 f = PR::zero(), g = PR::zero();
            if (valid && hi == 0) {
                const float x = xyz[(size_t)p * 3], y = xyz[(size_t)p * 3 + 1], z = xyz[(size_t)p * 3 + 2];
                const float gg = 5.0f * expf(-(x * x + y * y + z * z) / 0.08f);
                g[0] = (_Float16)(g_sigma[p] * expf(fminf(fmaxf(raw + gg, -15.0f), 15.0f)));
                const float4 gc = *reinterpret_cast<const float4 *>(g_rgbc + (size_t)p * 4);
                const float gcv[4] = {gc.x, gc.y, gc.z, gc.w};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const float sg = (float)(_Float16)(1.0f / (1.0f + expf(-out[0][k])));
                    f[k] = (_Float16)((k < (int)dm.n_rgb_out) ? gcv[k] * sg * (1.0f - sg) : 0.0f);
                }
            }
            bro[0] = f; bdo[0] = g;
        }
        // ================= chain, interleaved with the four weight-gradient phases so that fragments die early
        // ---- colour head: dz_r, then phase A (dW_ro, dW_r0) — frees hr, dfr
        frag_t zr[4];
        fb_zero(acc);
        fb_gemm_T<H, 2, 1>(wt + lt.off[6], rO, 16, FLD_HID, 0, FLD_HID, S32, bro, lane, acc);
        fb_c_to_b_masked<H>(acc, hr, zr);
        FF_PIN(col, hi);
        if (hi == 0) {
#pragma unroll
            for (int k = 0; k < 8; k++) ff_put(st, k, col, k < 4 ? bro[0][k] : (_Float16)0);
        }
        ff_stage_clayout(st, 8, col, hi, hr);
        FF_PIN(col, hi);
        ff_stage_clayout(st, 72, col, hi, zr);
        FF_PIN(col, hi);
        ff_stage_clayout(st, 136, col, hi, fea);
        FF_PIN(col, hi);
        ff_stage_natural<SDIR>(st, 200, col, hi, dfr);
        __syncthreads();
        ff_tile_mma(st, ff_tile_A(wave, dm, po), li, hi, wA0);
        ff_tile_mma(st, ff_tile_A(4 + wave, dm, po), li, hi, wA1);
        cn_f16v dfea[2];
        fb_zero(dfea);
        fb_gemm_T<H, 2, S64>(wt + lt.off[5], r0, FLD_HID, in_r0, FLD_NDIR, FLD_HID, S64, zr, lane, dfea);
        __syncthreads();
        // ---- density head: dz_d, then phase B (dW_do, dW_d0; fea stays at rows 136..199) — frees hd
        {
            frag_t zd[4];
            fb_zero(acc);
            fb_gemm_T<H, 2, 1>(wt + lt.off[4], dO, 16, FLD_HID, 0, FLD_HID, S32, bdo, lane, acc);
            fb_c_to_b_masked<H>(acc, hd, zd);
            FF_PIN(col, hi);
            if (hi == 0) {
#pragma unroll
                for (int k = 0; k < 8; k++) ff_put(st, k, col, k == 0 ? bdo[0][0] : (_Float16)0);
            }
            ff_stage_clayout(st, 8, col, hi, hd);
            FF_PIN(col, hi);
            ff_stage_clayout(st, 72, col, hi, zd);
            __syncthreads();
            ff_tile_mma(st, ff_tile_B(wave, dm, po), li, hi, wB0);
            if (wave >= 2) ff_tile_mma(st, ff_tile_B(2 + wave, dm, po), li, hi, wX);
            fb_gemm_T<H, 2, S64>(wt + lt.off[3], d0, FLD_HID, FLD_HID, 0, FLD_HID, S64, zd, lane, dfea);
            __syncthreads();
        }
        // ---- geometry network: dz_3 (= d fea), dz_2, then phase C (dW_n2, dW_n1) — frees h2
        frag_t z1[4];
        {
            frag_t z3[4], z2[4];
            fb_c_to_b<H, false>(dfea, z3);
            fb_zero(acc);
            fb_gemm_T<H, 2, S64>(wt + lt.off[2], n2, FLD_HID, FLD_HID, 0, FLD_HID, S64, z3, lane, acc);
            FF_PIN(col, hi);
            ff_stage_clayout(st, 0, col, hi, z3);
            FF_PIN(col, hi);
            ff_stage_clayout(st, 64, col, hi, hlast);
            if (NGEO == 2) {
                fb_c_to_b_masked<H>(acc, h2, z2);
                FF_PIN(col, hi);
                ff_stage_clayout(st, 128, col, hi, z2);
                FF_PIN(col, hi);
                ff_stage_clayout(st, 192, col, hi, h1);
                fb_zero(acc);
                fb_gemm_T<H, 2, S64>(wt + lt.off[1], n1, FLD_HID, FLD_HID, 0, FLD_HID, S64, z2, lane, acc);
            }
            fb_c_to_b_masked<H>(acc, h1, z1);
            __syncthreads();
            ff_tile_mma(st, ff_tile_C(wave, dm, po), li, hi, wC0);
            ff_tile_mma(st, ff_tile_C(4 + wave, dm, po), li, hi, wC1);
            __syncthreads();
        }
        // ---- first layer: phase D (dW_n0) and d(loss)/d(grid features)
        FF_PIN(col, hi);
        ff_stage_clayout(st, 0, col, hi, z1);
        FF_PIN(col, hi);
        {
            frag_t x0b[SENC];                                               // re-read (L2 hit) rather than hold 8 VGPRs across the chain
            fb_load_enc<H, SENC>(enc, P_, dm.L, p, valid, hi, x0b);
            ff_stage_natural<SENC>(st, 64, col, hi, x0b);
        }
        __syncthreads();
        if (wave < 2) ff_tile_mma(st, ff_tile_D(wave, dm, po), li, hi, wX);
        {
            cn_f16v denc[TENC];
            fb_zero(denc);
            fb_gemm_T<H, TENC, S64>(wt + lt.off[0], n0, FLD_HID, dm.enc_pad, 0, dm.enc_pad, S64, z1, lane, denc);
            if (valid) {
#pragma unroll
                for (int t = 0; t < TENC; t++) {
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const uint32_t level = (32 * t + fld_rho(r, hi)) >> 1;
                        if (level < dm.L) {
                            union { _Float16 h[2]; uint32_t u; } v;
                            v.h[0] = (_Float16)denc[t][r]; v.h[1] = (_Float16)denc[t][r + 1];
                            reinterpret_cast<uint32_t *>(grad_enc)[(size_t)level * P_ + p] = v.u;
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    // ---- this workgroup's partial weight gradients
    float *part = partials + (size_t)blockIdx.x * po.total;
    ff_tile_store(part, ff_tile_A(wave, dm, po), li, hi, wA0);
    ff_tile_store(part, ff_tile_A(4 + wave, dm, po), li, hi, wA1);
    ff_tile_store(part, ff_tile_B(wave, dm, po), li, hi, wB0);
    if (wave >= 2) ff_tile_store(part, ff_tile_B(2 + wave, dm, po), li, hi, wX);
    ff_tile_store(part, ff_tile_C(wave, dm, po), li, hi, wC0);
    ff_tile_store(part, ff_tile_C(4 + wave, dm, po), li, hi, wC1);
    if (wave < 2) ff_tile_store(part, ff_tile_D(wave, dm, po), li, hi, wX);
}

// g[i] += sum_b partials[b][i]  over the flat [net | den | rgb] parameter space.
// 64 columns per workgroup; the rows are split over 4 thread groups with 8 independent accumulators each (a single thread walking all
// 256 rows is one long dependent-latency chain: 100 us for 23 MB), combined through LDS.
__global__ void __launch_bounds__(256) k_field_reduce_partials(const float *__restrict__ partials, uint32_t n_blocks, uint32_t total, uint32_t n_net,
                                                               uint32_t n_den, float *__restrict__ g_net, float *__restrict__ g_den,
                                                               float *__restrict__ g_rgb, float *__restrict__ found_inf) {
    __shared__ float red[4][64];
    const uint32_t col = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const uint32_t i = blockIdx.x * 64 + col;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (i < total) {
        uint32_t b = grp;
        for (; b + 28 < n_blocks; b += 32) {
#pragma unroll
            for (int u = 0; u < 8; u++) acc[u] += partials[(size_t)(b + 4 * u) * total + i];
        }
        for (; b < n_blocks; b += 4) acc[0] += partials[(size_t)b * total + i];
    }
    red[grp][col] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (grp == 0 && i < total) {
        const float s = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
        float *dst = i < n_net ? g_net + i : (i < n_net + n_den ? g_den + (i - n_net) : g_rgb + (i - n_net - n_den));
        const float gn = *dst + s;
        *dst = gn;
        if (!(fabsf(gn) <= 3.4e38f) && found_inf) *found_inf = 1.0f;                // cnerf_scaler_watch: a non-finite parameter gradient
    }
}

// ------------------------------------------------------------------------------------------------ host entry (called from field_bwd.hip)
#define FF_MAX_BLOCKS 512              // partial-gradient rows in the workspace (the wave-specialised kernel uses two per workgroup)
uint64_t ff_workspace_bytes(const FieldDims &dm) { return (uint64_t)FF_MAX_BLOCKS * ff_offsets(dm).total * sizeof(float) + 256; }

// field_bwd_x2.hip: the two-pipeline kernel for 32-wide encodings (9..16 levels); the four-wave kernel of this file serves the narrower ones
bool x2_eligible(const FieldDims &dm);
int x2_launch(const void *enc, const float *xyz, const float *dirs, uint32_t dir_group, uint32_t P_, const FieldDims &dm, const float *pnet, const float *pden,
              const float *prgb, const float *g_sigma, const float *g_rgbc, void *grad_enc, float *g_net, float *g_den, float *g_rgb, void *workspace,
              uint32_t max_partials, const uint8_t *tile_live, hipStream_t st, const void *wimg);

void ff_reduce_partials(const float *partials, uint32_t n_partials, uint32_t total, uint32_t n_net, uint32_t n_den, float *g_net, float *g_den, float *g_rgb,
                        hipStream_t st) {
    hipLaunchKernelGGL(k_field_reduce_partials, dim3(cn_div_up(total, 64)), dim3(256), 0, st, partials, n_partials, total, n_net, n_den, g_net, g_den, g_rgb, g_cn_found_inf);
}

int ff_launch(const void *enc, const float *xyz, const float *dirs, uint32_t dir_group, uint32_t P_, const FieldDims &dm, const float *pnet,
              const float *pden, const float *prgb, const float *g_sigma, const float *g_rgbc, void *grad_enc, float *g_net, float *g_den,
              float *g_rgb, void *workspace, const uint8_t *tile_live, hipStream_t st, const void *wimg) {
    if (x2_eligible(dm)) {
        return x2_launch(enc, xyz, dirs, dir_group, P_, dm, pnet, pden, prgb, g_sigma, g_rgbc, grad_enc, g_net, g_den, g_rgb, workspace, FF_MAX_BLOCKS, tile_live, st,
                         wimg);
    }
    (void)wimg;                                                 // (the four-wave kernel stages from the float32 parameters)
    // (the four-wave kernel below serves the narrow encodings: it evaluates every tile, which is always correct — dead rows carry zero gradients)
    const FieldLds lo = fld_lds_layout<true>(dm);
    const FieldLdsT lt = fb_ldsT_layout<true>(dm);
    const uint32_t lds_bytes = (lo.off[7] + lt.off[7]) * sizeof(_Float16) + FF_STAGE_BYTES;
    if (lds_bytes > 160 * 1024) return CNERF_EINVAL;
    const uint32_t n_tiles = cn_div_up(P_, FLD_TILE);
    uint32_t blocks = cn_div_up(n_tiles, FLD_WAVES);
    if (blocks > 256) blocks = 256;
    const FfOff po = ff_offsets(dm);
    float *partials = reinterpret_cast<float *>(workspace);
    hipError_t e = hipMemsetAsync(partials, 0, (size_t)blocks * po.total * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
#define FF_CASE(SE16, NG, TE)                                                                                                                 \
    {                                                                                                                                         \
        auto kern = k_field_bwd_fused<(SE16), NG, TE>;                                                                                        \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);               \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(FLD_THREADS), lds_bytes, st, enc, xyz, dirs, dir_group, P_, dm, pnet, pden, prgb, g_sigma, \
                           g_rgbc, grad_enc, partials);                                                                                       \
    }
    const uint32_t se16 = dm.enc_pad / 16;
    if (dm.n_hidden_geo == 1) {
        switch (se16) {
            case 1: FF_CASE(1, 1, 1) break;
            case 2: FF_CASE(2, 1, 1) break;
            case 3: FF_CASE(3, 1, 2) break;
            case 4: FF_CASE(4, 1, 2) break;
            default: return CNERF_EINVAL;
        }
    } else {
        switch (se16) {
            case 1: FF_CASE(1, 2, 1) break;
            case 2: FF_CASE(2, 2, 1) break;
            case 3: FF_CASE(3, 2, 2) break;
            case 4: FF_CASE(4, 2, 2) break;
            default: return CNERF_EINVAL;
        }
    }
    int rc = cn_launch_status();
    if (rc) return rc;
    const uint32_t n_net = po.d0, n_den = po.r0 - po.d0;
    hipLaunchKernelGGL(k_field_reduce_partials, dim3(cn_div_up(po.total, 64)), dim3(256), 0, st, partials, blocks, po.total, n_net, n_den, g_net,
                       g_den, g_rgb, g_cn_found_inf);
    return cn_launch_status();
}
