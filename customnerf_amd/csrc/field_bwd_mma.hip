// Fused field backward, wave-specialised form (gfx950, fp16, enc_pad == 32).
//
// What the weight gradients need is every dz_l and every layer input with the SAMPLES along the MFMA contraction index, i.e.
// transposed with respect to how the activation chain holds them (lane = sample).  k_field_bwd_fused does that transposition with
// ~365 two-byte LDS writes per lane and tile.  Here the matrix core does it: a B operand that is the identity matrix turns an
// activation fragment (A operand: lane = sample, K-slots = features) into C registers with lane = feature, registers = samples
// — exactly the A/B fragment layout of dW = dz . a^T — at 2 MFMAs per 32 features, exact (x * 1.0 in fp32), no LDS traffic.
//
// The persistent fp32 weight-gradient tiles (24 x 16 registers) do not fit one wave next to the chain's working set, so the
// workgroup's four waves form two producer/consumer pairs over the same 32-sample tile:
//   wave A (net):   forward of the geometry MLP, publishes fea;   one tile later: dz_3 -> dz_1, dW_n2/n1/n0, d(loss)/d(grid features)
//   wave B (heads): density + colour heads forward and backward, dW_d0/dO/r0/rO, publishes dz_3 = d(loss)/d(fea)
// exchanged through 4 KiB LDS buffers (fragment order, 16-byte accesses), software-pipelined so that one workgroup barrier per
// tile is all the synchronisation there is: between two barriers A runs backward(t-1) + forward(t+1) while B runs heads(t).
#include "field_bwd_common.h"

struct MmOff {
    uint32_t n0, n1, n2, d0, dO, r0, rO, total;
};
__host__ __device__ __forceinline__ MmOff mm_offsets(const FieldDims &dm) {
    MmOff o;
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

#define MM_XCHG_BYTES 4096                 // one 64-row operand in fragment order: 4 k-steps x 64 lanes x 16 B

// identity B fragments: Id[k][n] = (feature(k) == n) for the two K-slot orders
__device__ __forceinline__ cn_h8 mm_id_clayout(int q, uint32_t li, uint32_t hi) {       // k-step q of a 32-feature tile in C-register order
    cn_h8 f;
#pragma unroll
    for (int j = 0; j < 8; j++) f[j] = ((uint32_t)fld_rho(8 * q + j, (int)hi) == li) ? (_Float16)1 : (_Float16)0;
    return f;
}
__device__ __forceinline__ cn_h8 mm_id_natural(int s, uint32_t li, uint32_t hi) {
    cn_h8 f;
#pragma unroll
    for (int j = 0; j < 8; j++) f[j] = (16u * s + 8u * hi + j == li) ? (_Float16)1 : (_Float16)0;
    return f;
}

struct MmT {                               // 32 features x 32 samples, lane = feature: the two K = 16 halves of a dW operand
    cn_h8 k0, k1;
};
__device__ __forceinline__ MmT mm_pack(const cn_f16v &c) {
    MmT t;
#pragma unroll
    for (int j = 0; j < 8; j++) { t.k0[j] = (_Float16)c[j]; t.k1[j] = (_Float16)c[8 + j]; }
    return t;
}
// transpose tile u (features 32u .. 32u+31) of a C-ordered 64-row operand
__device__ __forceinline__ MmT mm_tr_c(const cn_h8 *x, int u, const cn_h8 (&idc)[2]) {
    cn_f16v c;
#pragma unroll
    for (int r = 0; r < 16; r++) c[r] = 0.0f;
    c = Prec<true>::mfma(x[2 * u], idc[0], c);
    c = Prec<true>::mfma(x[2 * u + 1], idc[1], c);
    return mm_pack(c);
}
// transpose a natural-ordered 32-row operand (grid features, direction features)
__device__ __forceinline__ MmT mm_tr_n(const cn_h8 *x, const cn_h8 (&idn)[2]) {
    cn_f16v c;
#pragma unroll
    for (int r = 0; r < 16; r++) c[r] = 0.0f;
    c = Prec<true>::mfma(x[0], idn[0], c);
    c = Prec<true>::mfma(x[1], idn[1], c);
    return mm_pack(c);
}
// output-layer dz (<= 16 rows, K-step 0 of the C order only)
__device__ __forceinline__ MmT mm_tr_out(const cn_h8 &x, const cn_h8 (&idc)[2]) {
    cn_f16v c;
#pragma unroll
    for (int r = 0; r < 16; r++) c[r] = 0.0f;
    c = Prec<true>::mfma(x, idc[0], c);
    return mm_pack(c);
}
__device__ __forceinline__ void mm_dw(cn_f16v &acc, const MmT &z, const MmT &a) {
    acc = Prec<true>::mfma(z.k0, a.k0, acc);
    acc = Prec<true>::mfma(z.k1, a.k1, acc);
}
__device__ __forceinline__ void mm_store(float *__restrict__ part, uint32_t dst, uint32_t stride, uint32_t col0, uint32_t M, uint32_t N, uint32_t mt, uint32_t nt,
                                         uint32_t li, uint32_t hi, const cn_f16v &acc) {
    const uint32_t col = 32 * nt + li;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const uint32_t row = 32 * mt + (uint32_t)fld_rho(r, (int)hi);
        if (row < M && col < N) part[dst + (size_t)row * stride + col0 + col] = acc[r];
    }
}
__device__ __forceinline__ void mm_publish(unsigned char *buf, uint32_t lane, const cn_h8 *x) {
#pragma unroll
    for (int s = 0; s < 4; s++) *reinterpret_cast<cn_h8 *>(buf + (s * 64 + lane) * 16) = x[s];
}
__device__ __forceinline__ void mm_fetch(const unsigned char *buf, uint32_t lane, cn_h8 *x) {
#pragma unroll
    for (int s = 0; s < 4; s++) x[s] = *reinterpret_cast<const cn_h8 *>(buf + (s * 64 + lane) * 16);
}

template <int NGEO>
__global__ void __launch_bounds__(FLD_THREADS) k_field_bwd_mma(const void *__restrict__ enc, const float *__restrict__ xyz, const float *__restrict__ dirs,
                                                               uint32_t dir_group, uint32_t P_, FieldDims dm, const float *__restrict__ pnet,
                                                               const float *__restrict__ pden, const float *__restrict__ prgb,
                                                               const float *__restrict__ g_sigma, const float *__restrict__ g_rgbc,
                                                               void *__restrict__ grad_enc, float *__restrict__ partials) {
    constexpr bool H = true;
    constexpr int SENC = 2;                                   // enc_pad == 32
    using PR = Prec<H>;
    using frag_t = typename PR::frag_t;
    using elem_t = typename PR::elem_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char fld_lds[];
    elem_t *wl = reinterpret_cast<elem_t *>(fld_lds);
    const FieldLds lo = fld_lds_layout<H>(dm);
    const FieldLdsT lt = fb_ldsT_layout<H>(dm);
    elem_t *wt = wl + lo.off[7];
    unsigned char *xch = reinterpret_cast<unsigned char *>(wt + lt.off[7]);          // [pair][fea | dz3][2 buffers][4 KiB]
    const MmOff po = mm_offsets(dm);

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
    fb_stage_layer_T<H>(wt + lt.off[0], n0, FLD_HID, dm.enc_pad, 0, dm.enc_pad, 1, S64);
    if (NGEO == 2) fb_stage_layer_T<H>(wt + lt.off[1], n1, FLD_HID, FLD_HID, 0, FLD_HID, 2, S64);
    fb_stage_layer_T<H>(wt + lt.off[2], n2, FLD_HID, FLD_HID, 0, FLD_HID, 2, S64);
    fb_stage_layer_T<H>(wt + lt.off[3], d0, FLD_HID, FLD_HID, 0, FLD_HID, 2, S64);
    fb_stage_layer_T<H>(wt + lt.off[4], dO, 16, FLD_HID, 0, FLD_HID, 2, S32);
    fb_stage_layer_T<H>(wt + lt.off[5], r0, FLD_HID, in_r0, FLD_NDIR, FLD_HID, 2, S64);
    fb_stage_layer_T<H>(wt + lt.off[6], rO, 16, FLD_HID, 0, FLD_HID, 2, S32);
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63, li = lane & 31, hi = lane >> 5;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t pair = wave >> 1, role = wave & 1;                               // role 0 = A (net), 1 = B (heads)
    unsigned char *x_fea = xch + pair * 4 * MM_XCHG_BYTES, *x_dz3 = x_fea + 2 * MM_XCHG_BYTES;
    const cn_h8 idc[2] = {mm_id_clayout(0, li, hi), mm_id_clayout(1, li, hi)};
    const cn_h8 idn[2] = {mm_id_natural(0, li, hi), mm_id_natural(1, li, hi)};

    const uint32_t n_tiles = (P_ + FLD_TILE - 1) / FLD_TILE;
    const uint32_t n_pairs = gridDim.x * 2, gp = blockIdx.x * 2 + pair;
    const uint32_t n_iter = (n_tiles + n_pairs - 1) / n_pairs;                      // workgroup-uniform: same barrier count for every wave
    float *part = partials + (size_t)gp * po.total;

    if (role == 0) {
        // ======================================================================== wave A: geometry network
        cn_f16v wn2[2][2], wn1[2][2], wn0[2];
#pragma unroll
        for (int a = 0; a < 2; a++) {
#pragma unroll
            for (int r = 0; r < 16; r++) wn0[a][r] = 0.0f;
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int r = 0; r < 16; r++) { wn2[a][b][r] = 0.0f; wn1[a][b][r] = 0.0f; }
        }
        frag_t x0p[SENC], h1p[4], h2p[4];                     // the previous tile's activations (its backward runs one barrier later)
        uint32_t p_prev = 0;
        bool valid_prev = false;
        // grid features of the NEXT tile are requested one iteration ahead: at one wave per SIMD nothing else hides the load latency
        frag_t x0n[SENC];
        {
            const uint32_t p0 = gp * FLD_TILE + li;
            fb_load_enc<H, SENC>(enc, P_, dm.L, p0, (n_iter > 0) && p0 < P_, hi, x0n);
        }
        for (uint32_t i = 0; i <= n_iter; i++) {
            asm volatile("" ::: "memory");
            frag_t x0c[SENC], h1c[4], h2c[4];
            uint32_t p_cur = 0;
            bool valid_cur = false;
#pragma unroll
            for (int s = 0; s < SENC; s++) x0c[s] = x0n[s];
            {
                const uint32_t pn = (gp + (i + 1) * n_pairs) * FLD_TILE + li;
                fb_load_enc<H, SENC>(enc, P_, dm.L, pn, (i + 1 < n_iter) && pn < P_, hi, x0n);
            }
            if (i < n_iter) {
                const uint32_t tile = gp + i * n_pairs;
                p_cur = tile * FLD_TILE + li;
                valid_cur = p_cur < P_;
                cn_f16v acc[2];
                fb_zero(acc);
                fb_gemm<H, 2, SENC>(wl + lo.off[0], SENC, 0, x0c, lane, acc);
                fb_c_to_b<H, true>(acc, h1c);
                if (NGEO == 2) {
                    fb_zero(acc);
                    fb_gemm<H, 2, S64>(wl + lo.off[1], S64, 0, h1c, lane, acc);
                    fb_c_to_b<H, true>(acc, h2c);
                }
                frag_t fea[4];
                fb_zero(acc);
                fb_gemm<H, 2, S64>(wl + lo.off[2], S64, 0, (NGEO == 2) ? h2c : h1c, lane, acc);
                fb_c_to_b<H, false>(acc, fea);
                mm_publish(x_fea + (i & 1) * MM_XCHG_BYTES, lane, fea);
            }
            __syncthreads();
            if (i >= 1) {
                frag_t z3[4];
                mm_fetch(x_dz3 + ((i - 1) & 1) * MM_XCHG_BYTES, lane, z3);
                const frag_t *hlast = (NGEO == 2) ? h2p : h1p;
                cn_f16v acc[2];
                // dW_n2 = dz3 . hlast^T
                {
                    const MmT z0 = mm_tr_c(z3, 0, idc), z1 = mm_tr_c(z3, 1, idc), a0 = mm_tr_c(hlast, 0, idc), a1 = mm_tr_c(hlast, 1, idc);
                    mm_dw(wn2[0][0], z0, a0); mm_dw(wn2[0][1], z0, a1); mm_dw(wn2[1][0], z1, a0); mm_dw(wn2[1][1], z1, a1);
                }
                fb_zero(acc);
                fb_gemm_T<H, 2, S64>(wt + lt.off[2], n2, FLD_HID, FLD_HID, 0, FLD_HID, S64, z3, lane, acc);
                frag_t z1f[4];
                if (NGEO == 2) {
                    frag_t z2[4];
                    fb_c_to_b_masked<H>(acc, h2p, z2);
                    {
                        const MmT z0 = mm_tr_c(z2, 0, idc), z1 = mm_tr_c(z2, 1, idc), a0 = mm_tr_c(h1p, 0, idc), a1 = mm_tr_c(h1p, 1, idc);
                        mm_dw(wn1[0][0], z0, a0); mm_dw(wn1[0][1], z0, a1); mm_dw(wn1[1][0], z1, a0); mm_dw(wn1[1][1], z1, a1);
                    }
                    fb_zero(acc);
                    fb_gemm_T<H, 2, S64>(wt + lt.off[1], n1, FLD_HID, FLD_HID, 0, FLD_HID, S64, z2, lane, acc);
                }
                fb_c_to_b_masked<H>(acc, h1p, z1f);
                {
                    const MmT z0 = mm_tr_c(z1f, 0, idc), z1 = mm_tr_c(z1f, 1, idc), a0 = mm_tr_n(x0p, idn);
                    mm_dw(wn0[0], z0, a0); mm_dw(wn0[1], z1, a0);
                }
                cn_f16v denc[1];
                fb_zero(denc);
                fb_gemm_T<H, 1, S64>(wt + lt.off[0], n0, FLD_HID, dm.enc_pad, 0, dm.enc_pad, S64, z1f, lane, denc);
                if (valid_prev) {
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const uint32_t level = (uint32_t)fld_rho(r, (int)hi) >> 1;
                        if (level < dm.L) {
                            union { _Float16 h[2]; uint32_t u; } v;
                            v.h[0] = (_Float16)denc[0][r]; v.h[1] = (_Float16)denc[0][r + 1];
                            reinterpret_cast<uint32_t *>(grad_enc)[(size_t)level * P_ + p_prev] = v.u;
                        }
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < SENC; s++) x0p[s] = x0c[s];
#pragma unroll
            for (int s = 0; s < 4; s++) { h1p[s] = h1c[s]; h2p[s] = h2c[s]; }
            p_prev = p_cur;
            valid_prev = valid_cur;
        }
#pragma unroll
        for (int a = 0; a < 2; a++) {
            mm_store(part, po.n0, dm.enc_pad, 0, 64, dm.enc_pad, a, 0, li, hi, wn0[a]);
#pragma unroll
            for (int b = 0; b < 2; b++) {
                mm_store(part, po.n2, 64, 0, 64, 64, a, b, li, hi, wn2[a][b]);
                if (NGEO == 2) mm_store(part, po.n1, 64, 0, 64, 64, a, b, li, hi, wn1[a][b]);
            }
        }
    } else {
        // ======================================================================== wave B: density and colour heads
        cn_f16v wro[2], wrd[2], wrf[2][2], wdo[2], wd0[2][2];
#pragma unroll
        for (int a = 0; a < 2; a++) {
#pragma unroll
            for (int r = 0; r < 16; r++) { wro[a][r] = 0.0f; wrd[a][r] = 0.0f; wdo[a][r] = 0.0f; }
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int r = 0; r < 16; r++) { wrf[a][b][r] = 0.0f; wd0[a][b][r] = 0.0f; }
        }
        // raw per-sample inputs of the NEXT tile (position, direction, incoming gradients) are requested one iteration ahead
        struct BIn { float x, y, z, dx, dy, dz, gs; float4 gc; };
        auto load_in = [&](uint32_t tile, bool on) __attribute__((always_inline)) {
            BIn r;
            r.x = r.y = r.z = r.dx = r.dy = r.dz = r.gs = 0.0f;
            r.gc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            const uint32_t p = tile * FLD_TILE + li;
            if (on && p < P_) {
                const float *d = dirs + (size_t)(p / dir_group) * 3;
                r.dx = d[0]; r.dy = d[1]; r.dz = d[2];
                if (hi == 0) {
                    r.x = xyz[(size_t)p * 3]; r.y = xyz[(size_t)p * 3 + 1]; r.z = xyz[(size_t)p * 3 + 2];
                    r.gs = g_sigma[p];
                    r.gc = *reinterpret_cast<const float4 *>(g_rgbc + (size_t)p * 4);
                }
            }
            return r;
        };
        BIn nxt = load_in(gp, n_iter > 0);
        for (uint32_t i = 0; i <= n_iter; i++) {
            asm volatile("" ::: "memory");
            __syncthreads();
            if (i >= n_iter) continue;
            const uint32_t tile = gp + i * n_pairs;
            const uint32_t p = tile * FLD_TILE + li;
            const bool valid = p < P_;
            const BIn cur = nxt;
            nxt = load_in(gp + (i + 1) * n_pairs, i + 1 < n_iter);
            frag_t fea[4];
            mm_fetch(x_fea + (i & 1) * MM_XCHG_BYTES, lane, fea);
            // ---- forward of both heads
            cn_f16v acc[2], out[1];
            frag_t hd[4], hr[4], dfr[SDIR];
            fb_zero(acc);
            fb_gemm<H, 2, S64>(wl + lo.off[3], S64, 0, fea, lane, acc);
            fb_c_to_b<H, true>(acc, hd);
            fb_zero(out);
            fb_gemm<H, 1, S64>(wl + lo.off[4], S64, 0, hd, lane, out);
            const float raw = (float)(_Float16)out[0][0];
            fb_dir_frags_from<H>(cur.dx, cur.dy, cur.dz, valid, hi, dfr);
            fb_zero(acc);
            fb_gemm<H, 2, S64>(wl + lo.off[5], SR0, 0, fea, lane, acc);
            fb_gemm<H, 2, SDIR>(wl + lo.off[5], SR0, S64, dfr, lane, acc);
            fb_c_to_b<H, true>(acc, hr);
            fb_zero(out);
            fb_gemm<H, 1, S64>(wl + lo.off[6], S64, 0, hr, lane, out);
            // ---- output-layer gradients (sigmoid', clamped exp': provider_utils.py:26-29)
            frag_t bro[1], bdo[1];
            {
                cn_h8 f = PR::zero(), g = PR::zero();
                if (valid && hi == 0) {
                    const float x = cur.x, y = cur.y, z = cur.z;
                    const float gg = 5.0f * expf(-(x * x + y * y + z * z) / 0.08f);
                    g[0] = (_Float16)(cur.gs * expf(fminf(fmaxf(raw + gg, -15.0f), 15.0f)));
                    const float gcv[4] = {cur.gc.x, cur.gc.y, cur.gc.z, cur.gc.w};
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const float sg = (float)(_Float16)(1.0f / (1.0f + expf(-out[0][k])));
                        f[k] = (_Float16)((k < (int)dm.n_rgb_out) ? gcv[k] * sg * (1.0f - sg) : 0.0f);
                    }
                }
                bro[0] = f; bdo[0] = g;
            }
            const MmT tf0 = mm_tr_c(fea, 0, idc), tf1 = mm_tr_c(fea, 1, idc);       // fea^T: second operand of dW_r0 (fea part) and dW_d0
            cn_f16v dfea[2];
            fb_zero(dfea);
            // ---- colour head
            {
                frag_t zr[4];
                fb_zero(acc);
                fb_gemm_T<H, 2, 1>(wt + lt.off[6], rO, 16, FLD_HID, 0, FLD_HID, S32, bro, lane, acc);
                fb_c_to_b_masked<H>(acc, hr, zr);
                const MmT zo = mm_tr_out(bro[0], idc), a0 = mm_tr_c(hr, 0, idc), a1 = mm_tr_c(hr, 1, idc);
                mm_dw(wro[0], zo, a0); mm_dw(wro[1], zo, a1);
                const MmT z0 = mm_tr_c(zr, 0, idc), z1 = mm_tr_c(zr, 1, idc), td = mm_tr_n(dfr, idn);
                mm_dw(wrd[0], z0, td); mm_dw(wrd[1], z1, td);
                mm_dw(wrf[0][0], z0, tf0); mm_dw(wrf[0][1], z0, tf1); mm_dw(wrf[1][0], z1, tf0); mm_dw(wrf[1][1], z1, tf1);
                fb_gemm_T<H, 2, S64>(wt + lt.off[5], r0, FLD_HID, in_r0, FLD_NDIR, FLD_HID, S64, zr, lane, dfea);
            }
            // ---- density head
            {
                frag_t zd[4];
                fb_zero(acc);
                fb_gemm_T<H, 2, 1>(wt + lt.off[4], dO, 16, FLD_HID, 0, FLD_HID, S32, bdo, lane, acc);
                fb_c_to_b_masked<H>(acc, hd, zd);
                const MmT zo = mm_tr_out(bdo[0], idc), a0 = mm_tr_c(hd, 0, idc), a1 = mm_tr_c(hd, 1, idc);
                mm_dw(wdo[0], zo, a0); mm_dw(wdo[1], zo, a1);
                const MmT z0 = mm_tr_c(zd, 0, idc), z1 = mm_tr_c(zd, 1, idc);
                mm_dw(wd0[0][0], z0, tf0); mm_dw(wd0[0][1], z0, tf1); mm_dw(wd0[1][0], z1, tf0); mm_dw(wd0[1][1], z1, tf1);
                fb_gemm_T<H, 2, S64>(wt + lt.off[3], d0, FLD_HID, FLD_HID, 0, FLD_HID, S64, zd, lane, dfea);
            }
            frag_t z3[4];
            fb_c_to_b<H, false>(dfea, z3);
            mm_publish(x_dz3 + (i & 1) * MM_XCHG_BYTES, lane, z3);
        }
#pragma unroll
        for (int b = 0; b < 2; b++) {
            mm_store(part, po.rO, 64, 0, dm.n_rgb_out, 64, 0, b, li, hi, wro[b]);
            mm_store(part, po.dO, 64, 0, 1, 64, 0, b, li, hi, wdo[b]);
        }
#pragma unroll
        for (int a = 0; a < 2; a++) {
            mm_store(part, po.r0, 96, 0, 64, FLD_NDIR, a, 0, li, hi, wrd[a]);
#pragma unroll
            for (int b = 0; b < 2; b++) {
                mm_store(part, po.r0, 96, FLD_NDIR, 64, 64, a, b, li, hi, wrf[a][b]);
                mm_store(part, po.d0, 64, 0, 64, 64, a, b, li, hi, wd0[a][b]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ host entry (called from field_bwd_fused.hip)
void ff_reduce_partials(const float *partials, uint32_t n_partials, uint32_t total, uint32_t n_net, uint32_t n_den, float *g_net, float *g_den, float *g_rgb,
                        hipStream_t st);

bool mm_eligible(const FieldDims &dm) {
    static int on = -1;
    if (on < 0) {
        const char *e = getenv("CNERF_FIELD_MMA_BWD");
        on = e ? atoi(e) : 1;
    }
    return on && dm.enc_pad == 32 && (dm.n_hidden_geo == 1 || dm.n_hidden_geo == 2);
}

int mm_launch(const void *enc, const float *xyz, const float *dirs, uint32_t dir_group, uint32_t P_, const FieldDims &dm, const float *pnet, const float *pden,
              const float *prgb, const float *g_sigma, const float *g_rgbc, void *grad_enc, float *g_net, float *g_den, float *g_rgb, void *workspace,
              uint32_t max_partials, hipStream_t st) {
    const FieldLds lo = fld_lds_layout<true>(dm);
    const FieldLdsT lt = fb_ldsT_layout<true>(dm);
    const uint32_t lds_bytes = (lo.off[7] + lt.off[7]) * sizeof(_Float16) + 2 * 4 * MM_XCHG_BYTES;
    if (lds_bytes > 160 * 1024) return CNERF_EINVAL;
    const uint32_t n_tiles = cn_div_up(P_, FLD_TILE);
    uint32_t blocks = cn_div_up(n_tiles, 2);
    if (blocks > max_partials / 2) blocks = max_partials / 2;
    if (blocks > 256) blocks = 256;
    const MmOff po = mm_offsets(dm);
    float *partials = reinterpret_cast<float *>(workspace);
    // the partial rows are only written where a layer has rows / columns: zero the 2 * blocks rows in use (padding entries stay 0)
    hipError_t e0 = hipMemsetAsync(partials, 0, (size_t)2 * blocks * po.total * sizeof(float), st);
    if (e0 != hipSuccess) return (int)e0;
    if (dm.n_hidden_geo == 2) {
        auto kern = k_field_bwd_mma<2>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(FLD_THREADS), lds_bytes, st, enc, xyz, dirs, dir_group, P_, dm, pnet, pden, prgb, g_sigma, g_rgbc, grad_enc,
                           partials);
    } else {
        auto kern = k_field_bwd_mma<1>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(FLD_THREADS), lds_bytes, st, enc, xyz, dirs, dir_group, P_, dm, pnet, pden, prgb, g_sigma, g_rgbc, grad_enc,
                           partials);
    }
    int rc = cn_launch_status();
    if (rc) return rc;
    ff_reduce_partials(partials, 2 * blocks, po.total, po.d0, po.r0 - po.d0, g_net, g_den, g_rgb, st);
    return cn_launch_status();
}
