// Implicit-GEMM / GEMM kernel of the SDS path (UNet forward, VAE encoder forward + input-gradient) on the gfx950 matrix cores.
//   C[m][n] = epilogue(alpha * sum_k A(m,k) B[n][k]),  fp16 operands, fp32 accumulation (v_mfma_f32_32x32x16_f16).
// One workgroup (4 waves, 2 x 2) owns a 128 x BN output tile (BN = 128 or 64); each wave 64 x BN/2 = 2 x NT MFMA tiles.
// K advances in steps of 64 through a double-buffered LDS stage; rows are padded to 72 halfs (144 B) which makes the
// 16-byte fragment reads (ds_read_b128, 16-lane groups) conflict-free.  Both operands are K-contiguous in memory — NHWC
// activations make the im2col view K-contiguous per tap, weights are packed [Cout][kh][kw][Cin] — so every global access is a
// 16-byte load and nothing is transposed on the way to the MFMA fragments.
// im2col address arithmetic: when Cin % 64 == 0 (every conv but the first) a 64-wide K step lies inside one tap, so the tap
// (kh, kw, channel offset) is wave-uniform and advances incrementally; per thread only one pixel offset + bounds test remains
// (AMODE 2).  The generic path (AMODE 1) resolves the tap per 16-byte chunk.  Zero padding, nearest 2x upsampling and the
// transposed stride of input gradients are folded into that offset; there is no materialised im2col buffer.
// Epilogue: the fp32 tile is transposed through LDS so that bias / activation / residual / store run on 8 consecutive columns
// per lane (16-byte residual loads and stores instead of 2-byte ones).
// Measured and dropped: a second register set (operand tile two K steps ahead in flight): 611 vs 643 TFLOP/s at 4096^3, +-2 % on the UNet
// shapes — the loop is bound by its LDS traffic (ds_write_b128 staging + fragment reads), not by load latency.
// Small-M problems (the 8x8 / 16x16 UNet levels at batch 2) are split along K over blockIdx.z with fp32 partials in a
// workspace and a separate epilogue pass, so the launch still covers the 256 CUs.
#include "common.h"
#include "../../include/customnerf_sd.h"
#include "sd_gn_fix.h"

typedef _Float16 sd_h8 __attribute__((ext_vector_type(8)));
typedef float sd_f16v __attribute__((ext_vector_type(16)));
typedef float sd_f4 __attribute__((ext_vector_type(4)));

#define SG_BM 128
#define SG_BK 64
#define SG_LDK (SG_BK + 8)          // padded LDS row, halfs
#define SG_THREADS 256

__host__ __device__ __forceinline__ int sg_rho(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }
__host__ __device__ constexpr uint32_t sg_stage_halfs(int bn) { return (SG_BM + bn) * SG_LDK; }
__host__ __device__ constexpr uint32_t sg_lds_bytes(int bn) {
    const uint32_t stages = 2 * sg_stage_halfs(bn) * 2, ctile = SG_BM * (bn + 4) * 4;
    return stages > ctile ? stages : ctile;
}
// LDS-DMA loop: two unpadded stages
__host__ __device__ constexpr uint32_t sg_lds_bytes_glds(int bn) {
    const uint32_t stages = 2 * (SG_BM + bn) * SG_BK * 2, ctile = SG_BM * (bn + 4) * 4;
    return stages > ctile ? stages : ctile;
}

// (the sigmoid forms use the hardware reciprocal, as the GEMM kernel's own epilogue does: the result is rounded to half right after)
__device__ __forceinline__ float sg_act(float v, int act) {
    if (act == 1) return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
    if (act == 2) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
    if (act == 3) return v * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * v));
    return v;                                            // 0, and 4 (GEGLU pairs are combined by the caller)
}

// map (o*stride + k - pad) to an input coordinate; false = zero tap
__device__ __forceinline__ bool sg_coord(int32_t num, uint32_t tstride, uint32_t ups, uint32_t n_in, uint32_t &out) {
    if (num < 0) return false;
    if (tstride == 2) {
        if (num & 1) return false;
        num >>= 1;
    }
    if (ups == 2) num >>= 1;
    out = (uint32_t)num;
    return out < n_in;
}

// ---- epilogue phase 2 (shared by the 128-row kernel and the 256-row kernel of sd_gemm_big.hip): the fp32 tile sits in LDS as ct[BM][BN + 4];
// every thread takes 8 consecutive columns of NIT rows: bias / per-image bias / activation / residual / GroupNorm statistics / 16-byte stores.
// gn_lds: [GI = 4][36][2] 64-bit fixed-point scratch (images a tile can touch x (32 groups + 4) x {sum, sum of squares}).
template <int BM, int BN, int NTHREADS, bool SPLIT>
__device__ __forceinline__ void sg_epilogue_phase2(const CnerfSdGemm &g, const float *ct, long long (*gn_lds)[36][2], uint32_t tid, uint32_t m0, uint32_t n0,
                                                   uint32_t zo, uint32_t zi, uint32_t split, float *__restrict__ partial) {
    constexpr uint32_t LDC = BN + 4;
    constexpr int SG_EPI_THREADS = NTHREADS;
    // ---- phase 2: 8 consecutive columns per lane
    // optional: GroupNorm statistics of the output for the norm that reads it next.  A thread's column chunk is fixed over its rows, so
    // the (<= 2) groups' sums live in registers; flushed per image into a small LDS table, then one global atomic per (image, group) and tile.
    const bool do_gn = !SPLIT && g.gn_sums != nullptr;
    constexpr int GI = 4;                                // images a tile can touch (gn_rows >= 64)
    const uint32_t gn_cg = do_gn ? g.N / g.gn_groups : 1u;
    const uint32_t gn_g0 = n0 / gn_cg, gn_i0 = do_gn ? m0 / g.gn_rows : 0u;
    if (do_gn) {
        for (uint32_t i = tid; i < GI * 36 * 2; i += SG_EPI_THREADS) (&gn_lds[0][0][0])[i] = 0;
        __syncthreads();
    }
    float gs[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
    uint32_t gn_img = 0xFFFFFFFFu;
    _Float16 *C = (!SPLIT && g.C) ? reinterpret_cast<_Float16 *>(g.C) + g.sc_o * zo + g.sc_i * zi : nullptr;
    float *C32 = SPLIT ? partial + (size_t)split * g.M * g.N : (g.C32 ? g.C32 + g.sc_o * zo + g.sc_i * zi : nullptr);
    const _Float16 *R = (!SPLIT && g.residual) ? reinterpret_cast<const _Float16 *>(g.residual) + g.sc_o * zo + g.sc_i * zi : nullptr;
    const uint32_t ldc32 = SPLIT ? g.N : g.ldc;
    constexpr uint32_t CPR = BN / 8;                    // column chunks per row
    // The epilogue of a tile used to take 9.4 k cycles — 4 us, two thirds of a single-K-step launch and ~13 % of a large convolution's tile
    // time (cycle stamps) — almost all of it instruction issue: every element of every item went through its own `n + e < N` exec-mask
    // branch (three times), the activation switch and, for SiLU, an IEEE division.  Tiles that lie inside the problem (tile-uniform test)
    // take a guard-free instantiation; the bias of a thread's column chunk (the same for all its items) is read once; the activation is
    // selected once per item; the sigmoid forms use the hardware reciprocal (their result is rounded to half right after).
    float bias8[8];
    {
        const uint32_t nb_ = n0 + (tid % CPR) * 8;
#pragma unroll
        for (int e = 0; e < 8; e++) bias8[e] = (!SPLIT && g.bias && nb_ + e < g.N) ? g.bias[nb_ + e] : 0.0f;
    }
    auto phase2 = [&](auto inside_c) __attribute__((always_inline)) {
    constexpr bool INSIDE = decltype(inside_c)::value;   // the whole tile is inside [M, N]
    constexpr int NIT = BM * CPR / SG_EPI_THREADS;            // items per thread (8 | 4): fixed trip count, unrolled — the LDS reads of all items
#pragma unroll                                           // are in flight together instead of one read -> convert -> store chain per item
    for (int k = 0; k < NIT; k++) {
        const uint32_t c = tid + k * SG_EPI_THREADS;
        const uint32_t row = c / CPR, cc = (c % CPR) * 8;
        const uint32_t m = m0 + row, n = n0 + cc;
        if (!INSIDE && (m >= g.M || n >= g.N)) continue;
        const sd_f4 v0 = *reinterpret_cast<const sd_f4 *>(ct + row * LDC + cc), v1 = *reinterpret_cast<const sd_f4 *>(ct + row * LDC + cc + 4);
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        const bool full = INSIDE || n + 8 <= g.N;
        if (!SPLIT) {
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = v[e] * g.alpha + bias8[e];            // (columns beyond N are never stored)
            if (g.bias_rows) {
                const float *brow_ = g.bias_rows + (size_t)(m / g.rows_per_bias_row) * (g.ld_bias_rows ? g.ld_bias_rows : g.N);
                if (full) {
#pragma unroll
                    for (int e = 0; e < 8; e++) v[e] += brow_[n + e];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; e++)
                        if (INSIDE || n + e < g.N) v[e] += brow_[n + e];
                }
            }
            if (g.act == 1) {
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] = v[e] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[e]));
            } else if (g.act == 2) {
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752f));
            } else if (g.act == 3) {
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] = v[e] * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * v[e]));
            }
            if (R) {
                const _Float16 *rp = R + (size_t)m * g.ldr + n;
                if (full && ((g.ldr | n) & 7) == 0) {
                    const sd_h8 rv = *reinterpret_cast<const sd_h8 *>(rp);
#pragma unroll
                    for (int e = 0; e < 8; e++) v[e] += (float)rv[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; e++)
                        if (INSIDE || n + e < g.N) v[e] += (float)rp[e];
                }
            }
        }
        if (!SPLIT && g.act == 4) {
            // GEGLU on interleaved (value, gate) column pairs: out[m][n / 2 + e] = v[2e] * gelu(v[2e + 1]); the output is N / 2 wide
            _Float16 *cp = C + (size_t)m * g.ldc + n / 2;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                if (INSIDE || n + 2 * e + 1 < g.N) {
                    const float gt = v[2 * e + 1];
                    cp[e] = (_Float16)(v[2 * e] * (0.5f * gt * (1.0f + erff(gt * 0.70710678118654752f))));
                }
            }
            continue;
        }
        if (do_gn) {
            const uint32_t img = m / g.gn_rows;
            const uint32_t g_lo = n / gn_cg, split_c = (g_lo + 1) * gn_cg - n;      // columns [0, split_c) of the chunk belong to g_lo
            if (img != gn_img) {
                if (gn_img != 0xFFFFFFFFu) {
                    const uint32_t gl = n / gn_cg - gn_g0;
                    gn_add(&gn_lds[gn_img - gn_i0][gl][0], gs[0][0]);
                    gn_add(&gn_lds[gn_img - gn_i0][gl][1], gs[0][1]);
                    gn_add(&gn_lds[gn_img - gn_i0][gl + 1][0], gs[1][0]);
                    gn_add(&gn_lds[gn_img - gn_i0][gl + 1][1], gs[1][1]);
                    gs[0][0] = gs[0][1] = gs[1][0] = gs[1][1] = 0.0f;
                }
                gn_img = img;
            }
#pragma unroll
            for (int e = 0; e < 8; e++) {
                if (INSIDE || n + e < g.N) {
                    const float xf = (float)(_Float16)v[e];                         // what the consumer will read
                    const int k = ((uint32_t)e < split_c) ? 0 : 1;
                    gs[k][0] += xf;
                    gs[k][1] += xf * xf;
                }
            }
        }
        if (C) {
            _Float16 *cp = C + (size_t)m * g.ldc + n;
            if (full && ((g.ldc | n) & 7) == 0 && ((((uintptr_t)C) & 15) == 0)) {
                sd_h8 o;
#pragma unroll
                for (int e = 0; e < 8; e++) o[e] = (_Float16)v[e];
                *reinterpret_cast<sd_h8 *>(cp) = o;
            } else {
#pragma unroll
                for (int e = 0; e < 8; e++)
                    if (INSIDE || n + e < g.N) cp[e] = (_Float16)v[e];
            }
        }
        if (C32) {
            float *cp = C32 + (size_t)m * ldc32 + n;
#pragma unroll
            for (int e = 0; e < 8; e++)
                if (INSIDE || n + e < g.N) cp[e] = v[e];
        }
    }
    };
    if (m0 + BM <= g.M && n0 + BN <= g.N) phase2(std::true_type{});
    else phase2(std::false_type{});
    if (do_gn) {
        if (gn_img != 0xFFFFFFFFu) {
            const uint32_t gl = (n0 + (tid % CPR) * 8) / gn_cg - gn_g0;
            gn_add(&gn_lds[gn_img - gn_i0][gl][0], gs[0][0]);
            gn_add(&gn_lds[gn_img - gn_i0][gl][1], gs[0][1]);
            gn_add(&gn_lds[gn_img - gn_i0][gl + 1][0], gs[1][0]);
            gn_add(&gn_lds[gn_img - gn_i0][gl + 1][1], gs[1][1]);
        }
        __syncthreads();
        const uint32_t n_img = g.M / g.gn_rows;
        for (uint32_t i = tid; i < GI * 36; i += SG_EPI_THREADS) {
            const uint32_t im = i / 36, gl = i % 36, img = gn_i0 + im, grp = gn_g0 + gl;
            const long long a = gn_lds[im][gl][0], b2 = gn_lds[im][gl][1];
            if (img < n_img && grp < g.gn_groups) {
                long long *dst = reinterpret_cast<long long *>(g.gn_sums) + ((size_t)img * g.gn_groups + grp) * 2;
                gn_add_fixed(dst, a);
                gn_add_fixed(dst + 1, b2);
            }
        }
    }
}

// AMODE 0: dense A.  1: implicit conv, tap resolved per 16-byte chunk.  2: implicit conv with Cin % 64 == 0 (tap uniform per K step).
// row r of a [rows][64 halfs] LDS stage, 16-byte chunk c: XOR swizzle keyed on (row >> 1) & 7 — every ds_read_b128 lane group (16 lanes,
// consecutive-ish rows, one chunk column) then covers all sixteen 16-byte slots of the 256-byte bank row: conflict-free.
__device__ __forceinline__ uint32_t sg_swz(uint32_t row, uint32_t chunk) { return chunk ^ ((row >> 1) & 7u); }

// Measured and dropped: a 256 x 128 workgroup tile (each wave 128 x 64: 24 fragment reads per 32 MFMAs instead of 16 per 16) — 96..144 KiB of
// stages leave one workgroup = one wave per SIMD on a CU, and even with three stages, double-buffered fragments and the DMA rounds slotted
// between the MFMA groups it ran 480 / 607 / 655 TFLOP/s on the three large VAE convolutions against 608 / 727 / 797 for this tile.
// SIMPLE (LDS-DMA loop, AMODE 2): plain strided convolution — no transposed stride, no upsampling; the tap is a pixel offset + bounds test.
// Measured and dropped: four stages (three K steps in flight, counted vmcnt + bare s_barrier) for launches of at most one workgroup per CU.
// A lone workgroup's K step costs 0.35 us whatever the prefetch depth and whether the tile is 64 or 128 wide (scratch/gemm_kslope.py): the
// 24-32 KiB of a stage pass the CU's 64 B/clk vector-memory path in 380-510 cycles — the step is bound by that fill rate, not by latency.
// (The same rate bounds the chip: 32 KiB per 2.1 MFLOP tile step x 256 CUs x 64 B/clk = 2.5 PFLOP/s, i.e. the fill path and the matrix cores
// would both have to be busy all the time; the kernel's 0.6-0.8 PFLOP/s on the large convolutions is where the two pipelines overlap today.)
// A launch's fixed cost is 5.9 us (graph node 1.6 + kernel-argument load, first stage, epilogue's LDS round trip and stores).
template <int AMODE, int NT, bool SPLIT, bool GLDS = false, bool SIMPLE = false>
__global__ void __launch_bounds__(SG_THREADS) k_sd_gemm(const CnerfSdGemm g, float *__restrict__ partial, uint32_t k_tiles_per_split) {
    constexpr int BN = 64 * NT;
    constexpr int MT = 2, BM = SG_BM;                   // 32-row MFMA tiles per wave along M
    constexpr int CB = BN * 8 / SG_THREADS;             // 16-byte chunks of B per thread per K step (4 | 2)
    constexpr uint32_t STAGE = sg_stage_halfs(BN);
    extern __shared__ __attribute__((aligned(16))) unsigned char sg_lds[];
    _Float16 *lds = reinterpret_cast<_Float16 *>(sg_lds);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, li = lane & 31;
    const uint32_t wm = wave >> 1, wn = wave & 1;
    const uint32_t m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    uint32_t z = 0, split = 0;
    if (SPLIT) split = blockIdx.z; else z = blockIdx.z;
    const uint32_t zo = z / g.batch_inner, zi = z - zo * g.batch_inner;
    const _Float16 *A = reinterpret_cast<const _Float16 *>(g.A) + g.sa_o * zo + g.sa_i * zi;
    const _Float16 *B = reinterpret_cast<const _Float16 *>(g.B) + g.sb_o * zo + g.sb_i * zi;

    // ---- staging roles.  A: 2 threads per row, 4 consecutive chunks each.  B: 8 / CB threads per row, CB chunks each.
    const uint32_t arow = tid >> 1, ak = (tid & 1) * 32;
    const uint32_t brow = tid / (8 / CB), bk = (tid % (8 / CB)) * CB * 8;
    const _Float16 *abase;
    int32_t oh_s = 0, ow_s = 0;
    bool a_valid;
    {
        const uint32_t m = m0 + arow;
        a_valid = m < g.M;
        if (AMODE == 0) abase = A + (size_t)m * g.lda;
        else {
            const uint32_t hw = g.H_out * g.W_out;
            const uint32_t img = m / hw, rem = m - img * hw, oh = rem / g.W_out, ow = rem - oh * g.W_out;
            abase = A + (size_t)img * g.H_in * g.W_in * g.Cin;
            oh_s = (int32_t)(oh * g.stride) - (int32_t)g.pad_t;
            ow_s = (int32_t)(ow * g.stride) - (int32_t)g.pad_l;
        }
    }
    const uint32_t nrow = n0 + brow;
    const bool b_valid = nrow < g.N;
    const _Float16 *bbase = B + (size_t)nrow * g.ldb;

    const uint32_t n_ktiles = (g.K + SG_BK - 1) / SG_BK;
    uint32_t kt0 = 0, kt1 = n_ktiles;
    if (SPLIT) { kt0 = split * k_tiles_per_split; kt1 = min(kt0 + k_tiles_per_split, n_ktiles); }

    // uniform tap state of the fast conv path (advanced once per K step)
    uint32_t t_kh = 0, t_kw = 0, t_c0 = 0;
    if (AMODE == 2) {
        const uint32_t tap = (kt0 * SG_BK) / g.Cin;
        t_c0 = kt0 * SG_BK - tap * g.Cin;
        t_kh = tap / g.KW;
        t_kw = tap - t_kh * g.KW;
    }

    sd_f16v acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    if constexpr (!GLDS) {
        uint4 sa[4], sb[CB];
        auto fetch = [&](uint32_t kt) __attribute__((always_inline)) {
            const uint32_t k = kt * SG_BK;
            if (AMODE == 0) {
    #pragma unroll
                for (int c = 0; c < 4; c++) sa[c] = (a_valid && k + ak + 8 * c < g.K) ? *reinterpret_cast<const uint4 *>(abase + k + ak + 8 * c) : make_uint4(0, 0, 0, 0);
            } else if (AMODE == 2) {
                uint32_t ih = 0, iw = 0;
                const bool ok = a_valid && sg_coord(oh_s + (int32_t)t_kh, g.tstride, g.ups, g.H_in, ih) && sg_coord(ow_s + (int32_t)t_kw, g.tstride, g.ups, g.W_in, iw);
                const _Float16 *p = abase + ((size_t)ih * g.W_in + iw) * g.Cin + t_c0 + ak;
    #pragma unroll
                for (int c = 0; c < 4; c++) sa[c] = ok ? *reinterpret_cast<const uint4 *>(p + 8 * c) : make_uint4(0, 0, 0, 0);
                t_c0 += SG_BK;                                   // next K step (uniform)
                if (t_c0 >= g.Cin) {
                    t_c0 = 0;
                    if (++t_kw == g.KW) { t_kw = 0; ++t_kh; }
                }
            } else {
    #pragma unroll
                for (int c = 0; c < 4; c++) {
                    const uint32_t kk = k + ak + 8 * c;
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (a_valid && kk < g.K) {
                        const uint32_t tap = kk / g.Cin, ch = kk - tap * g.Cin, kh = tap / g.KW, kw = tap - kh * g.KW;
                        uint32_t ih = 0, iw = 0;
                        if (sg_coord(oh_s + (int32_t)kh, g.tstride, g.ups, g.H_in, ih) && sg_coord(ow_s + (int32_t)kw, g.tstride, g.ups, g.W_in, iw))
                            v = *reinterpret_cast<const uint4 *>(abase + ((size_t)ih * g.W_in + iw) * g.Cin + ch);
                    }
                    sa[c] = v;
                }
            }
    #pragma unroll
            for (int c = 0; c < CB; c++) sb[c] = (b_valid && k + bk + 8 * c < g.K) ? *reinterpret_cast<const uint4 *>(bbase + k + bk + 8 * c) : make_uint4(0, 0, 0, 0);
        };
        auto commit = [&](uint32_t stage) __attribute__((always_inline)) {
            _Float16 *sA = lds + stage * STAGE, *sB = sA + SG_BM * SG_LDK;
    #pragma unroll
            for (int c = 0; c < 4; c++) *reinterpret_cast<uint4 *>(sA + arow * SG_LDK + ak + 8 * c) = sa[c];
    #pragma unroll
            for (int c = 0; c < CB; c++) *reinterpret_cast<uint4 *>(sB + brow * SG_LDK + bk + 8 * c) = sb[c];
        };
        if (kt0 < kt1) {
            fetch(kt0);
            commit(0);
        }
        __syncthreads();
        for (uint32_t kt = kt0; kt < kt1; kt++) {
            const uint32_t stage = (kt - kt0) & 1;
            if (kt + 1 < kt1) fetch(kt + 1);
            const _Float16 *sA = lds + stage * STAGE, *sB = sA + SG_BM * SG_LDK;
    #pragma unroll
            for (int s = 0; s < SG_BK / 16; s++) {
                sd_h8 a[2], b[NT];
    #pragma unroll
                for (int i = 0; i < 2; i++) a[i] = *reinterpret_cast<const sd_h8 *>(sA + (wm * 64 + i * 32 + li) * SG_LDK + s * 16 + 8 * hi);
    #pragma unroll
                for (int j = 0; j < NT; j++) b[j] = *reinterpret_cast<const sd_h8 *>(sB + (wn * 32 * NT + j * 32 + li) * SG_LDK + s * 16 + 8 * hi);
    #pragma unroll
                for (int i = 0; i < 2; i++)
    #pragma unroll
                    for (int j = 0; j < NT; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            if (kt + 1 < kt1) commit(stage ^ 1);
            __syncthreads();
        }


    } else {
        // ---- LDS-DMA staging (global_load_lds_dwordx4): no staging registers, no ds_write pass.  Stage = A [128][64] + B [BN][64] halfs,
        // unpadded 128-byte rows.  Round r of an operand covers rows 32 r .. 32 r + 31: a wave-instruction fills 8 rows x 8 chunks
        // (lane-linear destination, as the instruction requires; 8 full 128-byte lines on the global side), thread t fills LDS slot
        // (row 32 r + t / 8, chunk t % 8) from global chunk sg_swz(row, t % 8): the XOR swizzle lives on the SOURCE address and on the
        // fragment reads.  (Measured and dropped: a chunk-major stage — one row per thread, no swizzle — touches 64 lines per
        // wave-instruction on the global side and ran 1.4-1.6x slower.)
        constexpr uint32_t GSTAGE = (BM + BN) * SG_BK * 2;                      // bytes per stage
        constexpr int RA = BM / 32;                                             // rounds of A per K step
        constexpr int RB = BN / 32;                                             // rounds of B per K step
        const uint32_t grow = tid >> 3, gch = tid & 7u;                         // this thread's row inside a round, LDS chunk column
        // Both operands go through buffer descriptors: 32-bit per-lane byte offsets instead of 64-bit pointer arithmetic, and a lane whose
        // offset lies beyond the buffer (set on purpose: a tap in the zero padding, a row or K chunk beyond the problem) gets ZEROS written
        // to its LDS slot by the hardware range check (scratch/buffer_lds_oob_test.hip) — no predication, no zero page.
        constexpr uint32_t OOB = 0xFFFFFF00u;
        const uint32_t a_batch = AMODE == 0 ? 1u : g.M / (g.H_out * g.W_out);
        const uint32_t a_bytes = AMODE == 0 ? (uint32_t)(((size_t)(g.M - 1) * g.lda + g.K) * 2) : (uint32_t)((size_t)a_batch * g.H_in * g.W_in * g.Cin * 2);
        const uint32_t b_bytes = (uint32_t)(((size_t)(g.N - 1) * g.ldb + g.K) * 2);
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(A), 0, a_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(B), 0, b_bytes, 0x00020000);
        constexpr bool simple = SIMPLE;
        const uint32_t wave_u = __builtin_amdgcn_readfirstlane(wave);           // LDS destinations (M0) stay scalar arithmetic
        uint32_t a_off[RA];                                                     // byte offset of this thread's source chunk in round r (tap (0, 0))
        int32_t a_oh[RA], a_ow[RA];
        bool a_ok[RA];
        const uint32_t a_sc0 = sg_swz(grow, gch) * 8;                            // source chunk (halfs): (row >> 1) & 7 is the same for rows 32 r + grow
#pragma unroll
        for (int r = 0; r < RA; r++) {
            const uint32_t row = 32 * r + grow, m = m0 + row;
            a_ok[r] = m < g.M;
            a_oh[r] = a_ow[r] = 0;
            const uint32_t mm = a_ok[r] ? m : 0;
            if (AMODE == 0) a_off[r] = (mm * g.lda + a_sc0) * 2;
            else {
                const uint32_t hw = g.H_out * g.W_out;
                const uint32_t img = mm / hw, rem = mm - img * hw, oh = rem / g.W_out, ow = rem - oh * g.W_out;
                a_oh[r] = (int32_t)(oh * g.stride) - (int32_t)g.pad_t;
                a_ow[r] = (int32_t)(ow * g.stride) - (int32_t)g.pad_l;
                // wrapping 32-bit arithmetic: for a tap inside the image the sum with the tap offset is the true byte offset
                a_off[r] = ((img * g.H_in + (uint32_t)a_oh[r]) * g.W_in + (uint32_t)a_ow[r]) * g.Cin * 2 + a_sc0 * 2;
                if (!simple) a_off[r] = img * g.H_in * g.W_in * g.Cin * 2 + a_sc0 * 2;
            }
        }
        uint32_t b_off[RB];
#pragma unroll
        for (int r = 0; r < RB; r++) {
            const uint32_t row = 32 * r + grow, n = n0 + row;
            b_off[r] = n < g.N ? (n * g.ldb + sg_swz(row, gch) * 8) * 2 : OOB;
        }
        // One K step's DMA: uniform state, then RA + RB rounds (one wave-instruction each), then the tap advance.  Offsets are built without
        // branches (bounds test -> OR with OOB): a `cond ? off : OOB` in front of the load turned into two exec-masked copies of the load.
        uint32_t i_k = 0, i_tap = 0;
        unsigned char *i_sA = sg_lds, *i_sB = sg_lds;
        bool i_kin = true;
        auto issue_begin = [&](uint32_t kt, uint32_t stage) __attribute__((always_inline)) {
            i_k = kt * SG_BK;
            i_sA = sg_lds + (size_t)stage * GSTAGE;
            i_sB = i_sA + BM * SG_BK * 2;
            i_kin = i_k + a_sc0 < g.K;                                           // (lane-dependent only in a ragged last K step of a dense problem)
            i_tap = AMODE == 2 ? ((t_kh * g.W_in + t_kw) * g.Cin + t_c0) * 2 : i_k * 2;      // uniform
        };
        auto issue_round = [&](int q) __attribute__((always_inline)) {
            if (q < RA) {
                const int r = q;
                uint32_t voff;
                if (AMODE == 0) {
                    voff = (a_off[r] + i_tap) | ((a_ok[r] && i_kin) ? 0u : OOB);
                } else if (simple) {
                    const bool ok = a_ok[r] && (uint32_t)(a_oh[r] + (int32_t)t_kh) < g.H_in && (uint32_t)(a_ow[r] + (int32_t)t_kw) < g.W_in;
                    voff = (a_off[r] + i_tap) | (ok ? 0u : OOB);
                } else {
                    uint32_t ih = 0, iw = 0;
                    const bool ok = a_ok[r] && sg_coord(a_oh[r] + (int32_t)t_kh, g.tstride, g.ups, g.H_in, ih) &&
                                    sg_coord(a_ow[r] + (int32_t)t_kw, g.tstride, g.ups, g.W_in, iw);
                    voff = ok ? a_off[r] + ((ih * g.W_in + iw) * g.Cin + t_c0) * 2 : OOB;
                }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void *)(i_sA + (32 * r + 8 * wave_u) * 128), 16, voff, 0, 0, 0);
            } else {
                const int r = q - RA;
                const uint32_t voff = b_off[r] | (i_kin ? 0u : OOB);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void *)(i_sB + (32 * r + 8 * wave_u) * 128), 16, voff, i_k * 2, 0, 0);
            }
        };
        auto issue_end = [&]() __attribute__((always_inline)) {
            if (AMODE == 2) {
                t_c0 += SG_BK;                                   // next K step (uniform)
                if (t_c0 >= g.Cin) {
                    t_c0 = 0;
                    if (++t_kw == g.KW) { t_kw = 0; ++t_kh; }
                }
            }
        };
        auto issue = [&](uint32_t kt, uint32_t stage) __attribute__((always_inline)) {
            issue_begin(kt, stage);
#pragma unroll
            for (int q = 0; q < RA + RB; q++) issue_round(q);
            issue_end();
        };
        // fragment read offsets: the swizzled chunk position depends on the K sub-step only through its two high bits — four offsets per row set
        uint32_t a_rd[MT][4], b_rd[NT][4];
#pragma unroll
        for (int s2 = 0; s2 < 4; s2++) {
#pragma unroll
            for (int i = 0; i < MT; i++) {
                const uint32_t row = wm * 32 * MT + i * 32 + li;
                a_rd[i][s2] = row * 128 + sg_swz(row, 2 * s2 + hi) * 16;
            }
#pragma unroll
            for (int j = 0; j < NT; j++) {
                const uint32_t row = wn * 32 * NT + j * 32 + li;
                b_rd[j][s2] = row * 128 + sg_swz(row, 2 * s2 + hi) * 16;
            }
        }
        // Two stages, one K step in flight.  (Measured and dropped: three stages with a counted vmcnt and a bare s_barrier — two steps in
        // flight — cost the 128-wide tile its second resident workgroup (96 KiB of stages) and ran 1.2-1.4x slower there; no change for
        // the 64-wide tile.)
        // The LDS-DMA data of OTHER waves is read after each barrier: every wave drains its own DMA (vmcnt(0)) before it arrives.  The wait is
        // spelled out — a workgroup-scope barrier does not promise it (hipcc happens to emit one today; Composable Kernel writes it too).
        if (kt0 < kt1) issue(kt0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (uint32_t kt = kt0; kt < kt1; kt++) {
            const uint32_t stage = (kt - kt0) & 1;
            if (kt + 1 < kt1) issue(kt + 1, stage ^ 1);
            const unsigned char *sA = sg_lds + (size_t)stage * GSTAGE, *sB = sA + BM * SG_BK * 2;
            // fragments of K sub-step s2 + 1 are requested before the MFMAs of s2 are issued (two register sets; the reads issue in the shadow of
            // the previous sub-step's MFMAs).  Left alone the compiler keeps four fragment registers and waits on every read.
            sd_h8 a[2][MT], b[2][NT];
#pragma unroll
            for (int i = 0; i < MT; i++) a[0][i] = *reinterpret_cast<const sd_h8 *>(sA + a_rd[i][0]);
#pragma unroll
            for (int j = 0; j < NT; j++) b[0][j] = *reinterpret_cast<const sd_h8 *>(sB + b_rd[j][0]);
#pragma unroll
            for (int s2 = 0; s2 < SG_BK / 16; s2++) {
                const int cur = s2 & 1, nxt = cur ^ 1;
                if (s2 + 1 < SG_BK / 16) {
#pragma unroll
                    for (int i = 0; i < MT; i++) a[nxt][i] = *reinterpret_cast<const sd_h8 *>(sA + a_rd[i][s2 + 1]);
#pragma unroll
                    for (int j = 0; j < NT; j++) b[nxt][j] = *reinterpret_cast<const sd_h8 *>(sB + b_rd[j][s2 + 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < MT; i++)
#pragma unroll
                    for (int j = 0; j < NT; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of stage ^ 1 has landed before anybody reads it
            __syncthreads();
        }
    }

    // ---- epilogue phase 1: fp32 tile -> LDS [128][BN + 4] (lane owns column li of its MFMA tile, rows rho(r, hi))
    constexpr uint32_t LDC = BN + 4;
    float *ct = reinterpret_cast<float *>(sg_lds);
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) ct[(wm * 32 * MT + i * 32 + sg_rho(r, hi)) * LDC + wn * 32 * NT + j * 32 + li] = acc[i][j][r];
    __syncthreads();
    // ---- phase 2: 8 consecutive columns per lane (sg_epilogue_phase2)
    __shared__ long long gn_lds[4][36][2];            // 64-bit fixed point (sd_gn_fix.h): exact, order-independent sums
    sg_epilogue_phase2<BM, BN, SG_THREADS, SPLIT>(g, ct, gn_lds, tid, m0, n0, zo, zi, split, partial);
}

// ================================================================================================ 256-row tile, eight waves, three stages (round 5)
// TUNING BUILDS ONLY (`make TUNING=1`, CNERF_SG_BIG=1): measured slower than the 128-row kernel on every shape it was built for — 652 / 655-683 /
// 756-810 TFLOP/s against 689 / 728-764 / 840-845 on the VAE's 512^2 x 128, 256^2 x 256 and 128^2 x 512 convolutions, whatever the schedule of the
// DMA rounds (profiles/r05_gemm_big_tile.txt).  Reading: 144 KiB of stages leave ONE workgroup per CU, so the prologue (two stages of DMA latency)
// and the epilogue (135 KiB of fp32 through LDS, 8 items per thread) of a tile are no longer hidden behind a second resident workgroup's main loop —
// with K = 1152 ... 4608 (18 ... 72 K steps) that is 10-30 % of a tile's time, more than the shared B tile and the deeper ring buy.
#ifdef CNERF_TUNING
// The 128 x {64, 128} kernel above fills 24-32 KiB of LDS per 2.1 MFLOP K step and drains its single step of prefetch at every barrier
// (`vmcnt(0)` + `__syncthreads()`): on the large-M convolutions of the VAE (M = 16 K ... 262 K rows, N = 128 ... 512) its matrix pipe is busy
// 28 % of the time (profiles/r04_sd_mfma_pmc.json).  This kernel takes a 256 x 128 tile with EIGHT waves (4 x 2, each the same 64 x 64 wave
// tile and the same 32x32x16 MFMA sequence per K step as above: the accumulation order, hence every output bit, is unchanged):
//   * the B tile (weights) is shared by 256 rows instead of 128: 48 KiB of LDS fill per 4.2 MFLOP instead of 64;
//   * THREE 48 KiB stages as a ring, two K steps in flight: `s_waitcnt vmcnt(6)` (the six LDS-DMA instructions of the newest step may still be
//     in flight) + a bare `s_barrier` per K step — the DMA queue never drains inside the loop (the guide's counted-vmcnt rule; the 128-row
//     kernel cannot afford a third stage without losing its second resident workgroup);
//   * one workgroup per CU = two waves per SIMD, as before (two four-wave workgroups), but they now share one barrier domain and one B tile;
//   * consecutive M tiles go to the same XCD (bijective swizzle): the 3 x 3 halo rows of neighbouring tiles are served by one L2.
// Dense / implicit-conv (Cin % 64 == 0) operands through the same buffer-descriptor LDS-DMA rounds (64 rows per round with 512 threads), the
// same XOR swizzle on the source chunk and on the fragment reads, the same epilogue (sg_epilogue_phase2), no split-K (the launches it serves have
// >= one workgroup per CU by themselves).
#define SGB_BM 256
#define SGB_BN 128
#define SGB_THREADS 512
#define SGB_STAGE ((SGB_BM + SGB_BN) * SG_BK * 2)                 // bytes
#define SGB_STAGES 3
__host__ __device__ constexpr uint32_t sgb_lds_bytes() {
    return (SGB_STAGES * SGB_STAGE > SGB_BM * (SGB_BN + 4) * 4 ? SGB_STAGES * SGB_STAGE : SGB_BM * (SGB_BN + 4) * 4) + 4 * 36 * 2 * 8;
}

// SCHED (how the six LDS-DMA instructions of a K step sit beside its sixteen MFMAs; measured: DESIGN.md B.11):
//   0  all waves: DMA, then the MFMAs
//   1  waves 0-3: DMA then MFMAs; waves 4-7 (the second wave of each SIMD): MFMAs then DMA — while one wave of a SIMD issues memory instructions
//      its partner has the matrix pipe (the guide's staggered wave groups, with one barrier per K step)
//   2  the DMA rounds slotted between the four MFMA groups of the K step (2, 2, 1, 1)
template <int AMODE, bool SIMPLE, int SCHED>
__global__ void __launch_bounds__(SGB_THREADS) k_sd_gemm_big(const CnerfSdGemm g) {
    constexpr int BM = SGB_BM, BN = SGB_BN, MT = 2, NT = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char sgb_lds[];          // the ONLY LDS object (a second one makes hipcc drain the DMA queue before every ds_read)
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, li = lane & 31;
    const uint32_t wm = wave >> 1, wn = wave & 1;
    // XCD-aware M-tile order: workgroup b runs on XCD b % 8; XCD x takes a contiguous run of M tiles
    uint32_t mt;
    {
        const uint32_t n = gridDim.x, q = n / CN_NXCD, r = n % CN_NXCD, xcd = blockIdx.x % CN_NXCD, k = blockIdx.x / CN_NXCD;
        mt = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const uint32_t m0 = mt * BM, n0 = blockIdx.y * BN;
    const uint32_t z = blockIdx.z;
    const uint32_t zo = z / g.batch_inner, zi = z - zo * g.batch_inner;
    const _Float16 *A = reinterpret_cast<const _Float16 *>(g.A) + g.sa_o * zo + g.sa_i * zi;
    const _Float16 *B = reinterpret_cast<const _Float16 *>(g.B) + g.sb_o * zo + g.sb_i * zi;
    const uint32_t n_k = (g.K + SG_BK - 1) / SG_BK;

    sd_f16v acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    constexpr int RA = BM / 64, RB = BN / 64;                                      // LDS-DMA rounds per K step (64 rows x 128 bytes per round)
    constexpr uint32_t OOB = 0xFFFFFF00u;
    const uint32_t grow = tid >> 3, gch = tid & 7u;
    const uint32_t a_batch = AMODE == 0 ? 1u : g.M / (g.H_out * g.W_out);
    const uint32_t a_bytes = AMODE == 0 ? (uint32_t)(((size_t)(g.M - 1) * g.lda + g.K) * 2) : (uint32_t)((size_t)a_batch * g.H_in * g.W_in * g.Cin * 2);
    const uint32_t b_bytes = (uint32_t)(((size_t)(g.N - 1) * g.ldb + g.K) * 2);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(A), 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(B), 0, b_bytes, 0x00020000);
    const uint32_t wave_u = __builtin_amdgcn_readfirstlane(wave);
    const uint32_t a_sc0 = sg_swz(grow, gch) * 8;                                   // (row >> 1) & 7 is the same for rows 64 r + grow
    uint32_t a_off[RA];
    int32_t a_oh[RA], a_ow[RA];
    bool a_ok[RA];
#pragma unroll
    for (int r = 0; r < RA; r++) {
        const uint32_t row = 64 * r + grow, m = m0 + row;
        a_ok[r] = m < g.M;
        a_oh[r] = a_ow[r] = 0;
        const uint32_t mm = a_ok[r] ? m : 0;
        if (AMODE == 0) a_off[r] = (mm * g.lda + a_sc0) * 2;
        else {
            const uint32_t hw = g.H_out * g.W_out;
            const uint32_t img = mm / hw, rem = mm - img * hw, oh = rem / g.W_out, ow = rem - oh * g.W_out;
            a_oh[r] = (int32_t)(oh * g.stride) - (int32_t)g.pad_t;
            a_ow[r] = (int32_t)(ow * g.stride) - (int32_t)g.pad_l;
            a_off[r] = ((img * g.H_in + (uint32_t)a_oh[r]) * g.W_in + (uint32_t)a_ow[r]) * g.Cin * 2 + a_sc0 * 2;      // wrapping 32-bit arithmetic (see k_sd_gemm)
            if (!SIMPLE) a_off[r] = img * g.H_in * g.W_in * g.Cin * 2 + a_sc0 * 2;
        }
    }
    uint32_t b_off[RB];
#pragma unroll
    for (int r = 0; r < RB; r++) {
        const uint32_t row = 64 * r + grow, n = n0 + row;
        b_off[r] = n < g.N ? (n * g.ldb + sg_swz(row, gch) * 8) * 2 : OOB;
    }
    uint32_t t_kh = 0, t_kw = 0, t_c0 = 0;                                         // uniform tap state (AMODE 2), advanced once per issued K step
    uint32_t i_k = 0, tap = 0;
    unsigned char *isA = sgb_lds, *isB = sgb_lds;
    bool kin = true;
    auto issue_begin = [&](uint32_t kt, uint32_t stage) __attribute__((always_inline)) {
        i_k = kt * SG_BK;
        isA = sgb_lds + (size_t)stage * SGB_STAGE;
        isB = isA + BM * SG_BK * 2;
        kin = i_k + a_sc0 < g.K;
        tap = AMODE == 2 ? ((t_kh * g.W_in + t_kw) * g.Cin + t_c0) * 2 : i_k * 2;
    };
    auto issue_round = [&](int q) __attribute__((always_inline)) {
        if (q < RA) {
            const int r = q;
            uint32_t voff;
            if (AMODE == 0) {
                voff = (a_off[r] + tap) | ((a_ok[r] && kin) ? 0u : OOB);
            } else if (SIMPLE) {
                const bool ok = a_ok[r] && (uint32_t)(a_oh[r] + (int32_t)t_kh) < g.H_in && (uint32_t)(a_ow[r] + (int32_t)t_kw) < g.W_in;
                voff = (a_off[r] + tap) | (ok ? 0u : OOB);
            } else {
                uint32_t ih = 0, iw = 0;
                const bool ok = a_ok[r] && sg_coord(a_oh[r] + (int32_t)t_kh, g.tstride, g.ups, g.H_in, ih) &&
                                sg_coord(a_ow[r] + (int32_t)t_kw, g.tstride, g.ups, g.W_in, iw);
                voff = ok ? a_off[r] + ((ih * g.W_in + iw) * g.Cin + t_c0) * 2 : OOB;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void *)(isA + (64 * r + 8 * wave_u) * 128), 16, voff, 0, 0, 0);
        } else {
            const int r = q - RA;
            const uint32_t voff = b_off[r] | (kin ? 0u : OOB);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void *)(isB + (64 * r + 8 * wave_u) * 128), 16, voff, i_k * 2, 0, 0);
        }
    };
    auto issue_end = [&]() __attribute__((always_inline)) {
        if (AMODE == 2) {
            t_c0 += SG_BK;
            if (t_c0 >= g.Cin) {
                t_c0 = 0;
                if (++t_kw == g.KW) { t_kw = 0; ++t_kh; }
            }
        }
    };
    auto issue = [&](uint32_t kt, uint32_t stage) __attribute__((always_inline)) {
        issue_begin(kt, stage);
#pragma unroll
        for (int q = 0; q < RA + RB; q++) issue_round(q);
        issue_end();
    };
    uint32_t a_rd[MT][4], b_rd[NT][4];
#pragma unroll
    for (int s2 = 0; s2 < 4; s2++) {
#pragma unroll
        for (int i = 0; i < MT; i++) {
            const uint32_t row = wm * 64 + i * 32 + li;
            a_rd[i][s2] = row * 128 + sg_swz(row, 2 * s2 + hi) * 16;
        }
#pragma unroll
        for (int j = 0; j < NT; j++) {
            const uint32_t row = wn * 64 + j * 32 + li;
            b_rd[j][s2] = row * 128 + sg_swz(row, 2 * s2 + hi) * 16;
        }
    }
    // ---- ring of three stages, two K steps in flight
    issue(0, 0);
    if (n_k > 1) issue(1, 1);
    uint32_t stage = 0;
    for (uint32_t it = 0; it < n_k; it++) {
        // this wave's share of step `it` has landed (the newer step's six DMA instructions may still be in flight) ...
        if (it + 1 < n_k) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ... and after the barrier everybody's has, and everybody is done reading the stage that step it + 2 overwrites (read in step it - 1)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const bool more = it + 2 < n_k;
        const uint32_t nstage = stage >= 1 ? stage - 1 : 2;                          // (it + 2) % 3
        const bool late = SCHED == 1 && wave_u >= 4;
        if (SCHED == 2) { if (more) issue_begin(it + 2, nstage); }
        else if (more && !late) issue(it + 2, nstage);
        const unsigned char *sA = sgb_lds + (size_t)stage * SGB_STAGE, *sB = sA + BM * SG_BK * 2;
        sd_h8 a[2][MT], b[2][NT];
#pragma unroll
        for (int i = 0; i < MT; i++) a[0][i] = *reinterpret_cast<const sd_h8 *>(sA + a_rd[i][0]);
#pragma unroll
        for (int j = 0; j < NT; j++) b[0][j] = *reinterpret_cast<const sd_h8 *>(sB + b_rd[j][0]);
#pragma unroll
        for (int s2 = 0; s2 < SG_BK / 16; s2++) {
            const int cur = s2 & 1, nxt = cur ^ 1;
            if (s2 + 1 < SG_BK / 16) {
#pragma unroll
                for (int i = 0; i < MT; i++) a[nxt][i] = *reinterpret_cast<const sd_h8 *>(sA + a_rd[i][s2 + 1]);
#pragma unroll
                for (int j = 0; j < NT; j++) b[nxt][j] = *reinterpret_cast<const sd_h8 *>(sB + b_rd[j][s2 + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < MT; i++)
#pragma unroll
                for (int j = 0; j < NT; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (SCHED == 2 && more) {
                if (s2 == 0) { issue_round(0); issue_round(1); }
                else if (s2 == 1) { issue_round(2); issue_round(3); }
                else if (s2 == 2) issue_round(4);
                else { issue_round(5); issue_end(); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (SCHED == 1 && more && late) issue(it + 2, nstage);
        stage = stage == 2 ? 0 : stage + 1;
    }
    __syncthreads();                                                               // every wave is done with the last stage: the tile may take its place
    // ---- epilogue phase 1: fp32 tile -> LDS [256][BN + 4]
    constexpr uint32_t LDC = BN + 4;
    float *ct = reinterpret_cast<float *>(sgb_lds);
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) ct[(wm * 64 + i * 32 + sg_rho(r, hi)) * LDC + wn * 64 + j * 32 + li] = acc[i][j][r];
    __syncthreads();
    constexpr uint32_t CT_BYTES = SGB_STAGES * SGB_STAGE > BM * LDC * 4 ? SGB_STAGES * SGB_STAGE : BM * LDC * 4;
    long long (*gn_lds)[36][2] = reinterpret_cast<long long (*)[36][2]>(sgb_lds + CT_BYTES);
    sg_epilogue_phase2<BM, BN, SGB_THREADS, false>(g, ct, gn_lds, tid, m0, n0, zo, zi, 0u, nullptr);
}
#endif  // CNERF_TUNING

// split-K tail: sum the partials, apply the epilogue.
// Round 6: a GroupNorm-statistics request (g.gn_sums) is served here too — rounds 2-5 left it to a separate k_gn_stats launch whenever the
// producing GEMM ran split-K, i.e. for almost every norm of the 8^2 / 16^2 / 32^2 UNet levels (59 launches of ~8 us per edit step).  A thread
// owns 4 consecutive columns of a row: at most two groups (N / groups >= 4); its two (sum, sum of squares) pairs of the ROUNDED outputs — what the
// consumer reads — go to a per-workgroup LDS table [images][groups][2] in 64-bit fixed point (sd_gn_fix.h), the table to the global sums by one
// atomic per touched entry.  SG_EPI_GN_MAX entries (images x groups): the host keeps larger requests off the split-K path.
#define SG_EPI_GN_MAX 512
__global__ void __launch_bounds__(256) k_sd_gemm_splitk_epilogue(const CnerfSdGemm g, const float *__restrict__ partial, uint32_t splits) {
    const size_t total = (size_t)g.M * g.N;
    __shared__ long long gn_tab[SG_EPI_GN_MAX][2];
    const bool do_gn = g.gn_sums != nullptr;
    const uint32_t gn_cg = do_gn ? g.N / g.gn_groups : 1u, gn_entries = do_gn ? (g.M / g.gn_rows) * g.gn_groups : 0u;
    if (do_gn) {
        for (uint32_t i = threadIdx.x; i < gn_entries * 2; i += 256) (&gn_tab[0][0])[i] = 0;
        __syncthreads();
    }
    auto gn_flush = [&]() {
        __syncthreads();
        long long *dst = reinterpret_cast<long long *>(g.gn_sums);
        for (uint32_t i = threadIdx.x; i < gn_entries * 2; i += 256) gn_add_fixed(dst + i, (&gn_tab[0][0])[i]);
    };
    if (g.act == 4) {                                       // GEGLU pairs: one thread per output element
        const size_t half = total / 2;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
            const uint32_t m = (uint32_t)(i / (g.N / 2)), c = (uint32_t)(i - (size_t)m * (g.N / 2)), n = 2 * c;
            float a = 0.0f, b = 0.0f;
            for (uint32_t s = 0; s < splits; s++) {
                const float2 p = *reinterpret_cast<const float2 *>(partial + (size_t)s * total + (size_t)m * g.N + n);
                a += p.x; b += p.y;
            }
            a = a * g.alpha + (g.bias ? g.bias[n] : 0.0f);
            b = b * g.alpha + (g.bias ? g.bias[n + 1] : 0.0f);
            reinterpret_cast<_Float16 *>(g.C)[(size_t)m * g.ldc + c] = (_Float16)(a * (0.5f * b * (1.0f + erff(b * 0.70710678118654752f))));
        }
        return;
    }
    // four consecutive columns per thread (16-byte partial loads, 8-byte stores, 32-bit index arithmetic) when the row pitches allow it;
    // the element-wise loop below is the general form
    if ((g.N & 3u) == 0 && (g.ldc & 3u) == 0 && (!g.residual || (g.ldr & 3u) == 0) && total < 0xFFFFFFFFull) {
        const uint32_t total4 = (uint32_t)(total / 4), n4 = g.N / 4;
        // With a statistics request a workgroup owns a TILE of the output — a slab of four whole groups x a range of rows (gridDim = slabs x row
        // chunks) — so that its table flush is four or five entries per image instead of every (image, group) pair.  (Grid-stride, 2048 workgroups x
        // 128 atomics on the same 128 addresses made this kernel 15 us instead of 6.5; contiguous row ranges under a 256-workgroup cap still 11:
        // profiles/r06_edit_step_kernels_gnstats_first.txt, r06_edit_step_kernels_gnstats_rows.txt.  Tiles: 9.2 us.  Also measured and dropped: a
        // fixed column chunk per thread walking four rows with its sums in registers — fewer LDS atomics, but a quarter of the workgroups and a
        // serial chain of split loads per thread: 12.9 us, profiles/r06_edit_step_kernels_gnstats_regs.txt.)
        const bool tiled = do_gn && (g.gn_groups & 3u) == 0;
        const uint32_t n_slabs = tiled ? g.gn_groups / 4 : 1u, slab_items = tiled ? gn_cg : n4;          // items (4 columns) per row of a slab
        const uint32_t row_chunks = do_gn ? gridDim.x / n_slabs : 1u, rpb = do_gn ? (g.M + row_chunks - 1) / row_chunks : g.M;
        const uint32_t slab = do_gn ? blockIdx.x % n_slabs : 0u, row0 = do_gn ? (blockIdx.x / n_slabs) * rpb : 0u;
        const uint32_t rows_in = do_gn ? (row0 < g.M ? min(rpb, g.M - row0) : 0u) : g.M;
        const uint32_t my_items = do_gn ? rows_in * slab_items : total4;
        const uint32_t j_lo = do_gn ? threadIdx.x : blockIdx.x * blockDim.x + threadIdx.x, j_step = do_gn ? blockDim.x : gridDim.x * blockDim.x;
        for (uint32_t j = j_lo; j < my_items; j += j_step) {
            uint32_t i = j;
            if (do_gn) {
                const uint32_t r = j / slab_items, c4 = j - r * slab_items;
                i = (row0 + r) * n4 + slab * slab_items + c4;
            }
            const uint32_t m = i / n4, n = (i - m * n4) * 4;
            sd_f4 a = {0.0f, 0.0f, 0.0f, 0.0f};
            for (uint32_t s = 0; s < splits; s++) a += *reinterpret_cast<const sd_f4 *>(partial + (size_t)s * total + (size_t)i * 4);
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = a[e] * g.alpha + (g.bias ? g.bias[n + e] : 0.0f);
            if (g.bias_rows) {
                const float *br = g.bias_rows + (size_t)(m / g.rows_per_bias_row) * (g.ld_bias_rows ? g.ld_bias_rows : g.N) + n;
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] += br[e];
            }
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = sg_act(v[e], g.act);
            if (g.residual) {
                typedef _Float16 sd_h4 __attribute__((ext_vector_type(4)));
                const _Float16 *rp = reinterpret_cast<const _Float16 *>(g.residual) + (size_t)m * g.ldr + n;
                if ((((uintptr_t)rp) & 7) == 0) {
                    const sd_h4 r = *reinterpret_cast<const sd_h4 *>(rp);
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] += (float)r[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] += (float)rp[e];
                }
            }
            if (do_gn) {
                const uint32_t img = m / g.gn_rows, g_lo = n / gn_cg, split_c = (g_lo + 1) * gn_cg - n;     // columns [0, split_c) of the four belong to g_lo
                float s0 = 0.0f, q0 = 0.0f, s1 = 0.0f, q1 = 0.0f;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float xf = (float)(_Float16)v[e];                         // what the consumer will read
                    if ((uint32_t)e < split_c) { s0 += xf; q0 += xf * xf; } else { s1 += xf; q1 += xf * xf; }
                }
                long long *t0 = gn_tab[img * g.gn_groups + g_lo];
                gn_add(t0, s0); gn_add(t0 + 1, q0);
                if (split_c < 4) { gn_add(t0 + 2, s1); gn_add(t0 + 3, q1); }        // (the next group of the same image: n + 3 < N)
            }
            if (g.C) {
                typedef _Float16 sd_h4 __attribute__((ext_vector_type(4)));
                sd_h4 o;
#pragma unroll
                for (int e = 0; e < 4; e++) o[e] = (_Float16)v[e];
                _Float16 *cp = reinterpret_cast<_Float16 *>(g.C) + (size_t)m * g.ldc + n;
                if ((((uintptr_t)cp) & 7) == 0) *reinterpret_cast<sd_h4 *>(cp) = o;
                else {
#pragma unroll
                    for (int e = 0; e < 4; e++) cp[e] = o[e];
                }
            }
            if (g.C32) {
#pragma unroll
                for (int e = 0; e < 4; e++) g.C32[(size_t)m * g.ldc + n + e] = v[e];
            }
        }
        if (do_gn) gn_flush();
        return;
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t m = (uint32_t)(i / g.N), n = (uint32_t)(i - (size_t)m * g.N);
        float a = 0.0f;
        for (uint32_t s = 0; s < splits; s++) a += partial[(size_t)s * total + i];
        float v = a * g.alpha + (g.bias ? g.bias[n] : 0.0f);
        if (g.bias_rows) v += g.bias_rows[(size_t)(m / g.rows_per_bias_row) * (g.ld_bias_rows ? g.ld_bias_rows : g.N) + n];
        v = sg_act(v, g.act);
        if (g.residual) v += (float)reinterpret_cast<const _Float16 *>(g.residual)[(size_t)m * g.ldr + n];
        if (do_gn) {
            const float xf = (float)(_Float16)v;
            long long *t0 = gn_tab[(m / g.gn_rows) * g.gn_groups + n / gn_cg];
            gn_add(t0, xf); gn_add(t0 + 1, xf * xf);
        }
        if (g.C) reinterpret_cast<_Float16 *>(g.C)[(size_t)m * g.ldc + n] = (_Float16)v;
        if (g.C32) g.C32[(size_t)m * g.ldc + n] = v;
    }
    if (do_gn) gn_flush();
}

// split-K tail with the LayerNorm of the output rows (round 6): one wave per row — the transformer blocks normalise the result of proj_in /
// attn1.to_out / attn2.to_out right away, a k_layernorm launch of ~5 us each (48 per UNet forward) that reads back what this kernel just wrote.
// Lane l owns the 8-column chunks l, l + 64, ...; the element arithmetic is the 4-column path's above (same partial order, alpha, bias, per-image
// bias, activation, residual), the LayerNorm is k_layernorm's (sd_ops.hip) on the half-rounded values: both outputs are bit-identical to the two launches.
template <int CPL>
__global__ void __launch_bounds__(256) k_sd_gemm_splitk_epilogue_ln(const CnerfSdGemm g, const float *__restrict__ partial, uint32_t splits) {
    const uint32_t lane = threadIdx.x & 63, m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= g.M) return;
    const uint32_t nchunks = g.N / 8;
    const size_t total = (size_t)g.M * g.N;
    float v[CPL][8];
    float sum = 0.0f;
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        const uint32_t col = lane + 64 * c, n = col * 8;
        if (col < nchunks) {
            sd_f4 a0 = {0.0f, 0.0f, 0.0f, 0.0f}, a1 = {0.0f, 0.0f, 0.0f, 0.0f};
            const float *pp = partial + (size_t)m * g.N + n;
            for (uint32_t s = 0; s < splits; s++) {
                a0 += *reinterpret_cast<const sd_f4 *>(pp + (size_t)s * total);
                a1 += *reinterpret_cast<const sd_f4 *>(pp + (size_t)s * total + 4);
            }
            float x[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
            for (int e = 0; e < 8; e++) x[e] = x[e] * g.alpha + (g.bias ? g.bias[n + e] : 0.0f);
            if (g.bias_rows) {
                const float *br = g.bias_rows + (size_t)(m / g.rows_per_bias_row) * (g.ld_bias_rows ? g.ld_bias_rows : g.N) + n;
#pragma unroll
                for (int e = 0; e < 8; e++) x[e] += br[e];
            }
#pragma unroll
            for (int e = 0; e < 8; e++) x[e] = sg_act(x[e], g.act);
            if (g.residual) {
                const _Float16 *rp = reinterpret_cast<const _Float16 *>(g.residual) + (size_t)m * g.ldr + n;
#pragma unroll
                for (int e = 0; e < 8; e++) x[e] += (float)rp[e];
            }
            sd_h8 o;
#pragma unroll
            for (int e = 0; e < 8; e++) { o[e] = (_Float16)x[e]; v[c][e] = (float)o[e]; sum += v[c][e]; }
            *reinterpret_cast<sd_h8 *>(reinterpret_cast<_Float16 *>(g.C) + (size_t)m * g.ldc + n) = o;
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++) v[c][e] = 0.0f;
        }
    }
    const float mean = cn_wave_sum(sum) / (float)g.N;
    float sq = 0.0f;
#pragma unroll
    for (int c = 0; c < CPL; c++)
        if (lane + 64 * c < nchunks) {
#pragma unroll
            for (int e = 0; e < 8; e++) { const float d = v[c][e] - mean; sq += d * d; }
        }
    const float rstd = rsqrtf(cn_wave_sum(sq) / (float)g.N + g.ln_eps);
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        const uint32_t col = lane + 64 * c;
        if (col < nchunks) {
            sd_h8 o;
#pragma unroll
            for (int e = 0; e < 8; e++) o[e] = (_Float16)((v[c][e] - mean) * rstd * g.ln_gamma[col * 8 + e] + g.ln_beta[col * 8 + e]);
            *reinterpret_cast<sd_h8 *>(reinterpret_cast<_Float16 *>(g.ln_out) + (size_t)m * g.N + col * 8) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side
// the LayerNorm request is served by the split-K tail only: dense half output rows of at most 2048 columns, 16-byte aligned chunks
static bool sg_ln_admissible(const CnerfSdGemm *g) {
    return g->ln_out && g->ln_gamma && g->ln_beta && g->C && !g->C32 && !g->gn_sums && g->act != 4 && (g->N & 7) == 0 && g->N <= 2048 && (g->ldc & 7) == 0 &&
           (!g->residual || (g->ldr & 7) == 0) && g->batch_outer * g->batch_inner == 1 &&
           ((((uintptr_t)g->C) | ((uintptr_t)g->ln_out) | ((uintptr_t)g->residual)) & 15) == 0;
}

static int sg_check(const CnerfSdGemm *g) {
    if (!g) return CNERF_ENULL;
    if (!g->A || !g->B || (!g->C && !g->C32)) return CNERF_ENULL;
    if (g->M == 0 || g->N == 0 || g->K == 0 || (g->K & 7) || (g->ldb & 7)) return CNERF_EINVAL;
    if (g->batch_outer == 0 || g->batch_inner == 0) return CNERF_EINVAL;
    if (g->act < 0 || g->act > 4) return CNERF_EINVAL;
    if (g->act == 4 && ((g->N & 7) || g->residual || g->C32 || !g->C || g->gn_sums)) return CNERF_EINVAL;
    if (g->bias_rows && g->rows_per_bias_row == 0) return CNERF_EINVAL;
    if (g->gn_sums && (g->gn_groups == 0 || g->N % g->gn_groups || g->N / g->gn_groups < 4 || g->gn_rows < 64 || g->M % g->gn_rows ||
                       g->batch_outer * g->batch_inner != 1))
        return CNERF_EINVAL;
    if ((((uintptr_t)g->A) | ((uintptr_t)g->B)) & 15) return CNERF_EINVAL;
    if (g->mode == 0) {
        if (g->lda & 7) return CNERF_EINVAL;
    } else if (g->mode == 1) {
        if (g->Cin == 0 || (g->Cin & 7) || g->KH == 0 || g->KW == 0 || g->K != g->KH * g->KW * g->Cin) return CNERF_EINVAL;
        if (g->H_out == 0 || g->W_out == 0 || g->M % (g->H_out * g->W_out)) return CNERF_EINVAL;
        if (g->ups < 1 || g->ups > 2 || g->tstride < 1 || g->tstride > 2 || g->stride == 0) return CNERF_EINVAL;
        if (g->batch_outer * g->batch_inner != 1) return CNERF_EINVAL;
    } else return CNERF_EINVAL;
    return CNERF_OK;
}

// Tile width (128 x 64 or 128 x 128) and split-K factor from a small cost model fitted to a sweep over the UNet / VAE shapes
// (scratch/gemm_sweep.py, MI355X): a workgroup costs (K steps + 8) x t_k microseconds (t_k = 1.0 for the 64-wide tile, 1.5 for the
// 128-wide one, two workgroups resident per CU); 512 workgroups run at once, more cost further rounds; a split adds the fp32
// partial round trip and one launch.  The old rule (ceil(512 / tiles) splits, only below 256 tiles) overshot 512 workgroups and paid a
// second, nearly empty round on the 32x32 / 16x16 UNet levels: 1.3-1.4x slower there.
struct SgPlan { int nt; uint32_t splits, kps; };
static int sg_env(const char *name) { return cn_tune_env(name, 0); }     // release build: always 0 (common.h)
static SgPlan sg_plan(const CnerfSdGemm *g) {
    const uint32_t n_ktiles = cn_div_up(g->K, SG_BK);
    const uint32_t batch = g->batch_outer * g->batch_inner;
    static const int force_nt = sg_env("CNERF_SG_NT"), force_sp = sg_env("CNERF_SG_SPLITS");  // tuning hooks (scratch/gemm_sweep.py)
    SgPlan best = {1, 1, n_ktiles};
    double best_cost = 1e30;
    for (int nt = 1; nt <= 2; nt++) {
        if (force_nt && nt != force_nt) continue;
        const uint32_t tiles = cn_div_up(g->M, SG_BM) * cn_div_up(g->N, 64 * nt) * batch;
        const double t_k = nt == 1 ? 1.0 : 1.5;
        const uint32_t max_sp = batch != 1 ? 1 : 32;
        for (uint32_t sp = 1; sp <= max_sp; sp++) {
            if (force_sp && sp != (uint32_t)force_sp && !(batch != 1 && sp == 1)) continue;
            const uint32_t kps = cn_div_up(n_ktiles, sp);
            if (sp > 1 && (kps < 8 || cn_div_up(n_ktiles, kps) != sp)) continue;           // at least 8 K steps per split; no empty splits
            const uint64_t blocks = (uint64_t)tiles * sp;
            const double rounds = (double)((blocks + 511) / 512);                             // measured: 640 workgroups cost two full rounds
            double cost = rounds * (kps + 8.0) * t_k;
            if (blocks < 256) cost *= 0.6 + 0.4 * (double)blocks / 256.0;                     // a lone workgroup on a CU runs its K steps faster
            if (sp > 1) cost += 3.0 + 2.0 * (double)g->M * g->N * 4.0 * sp / 4.0e6;
            if (cost < best_cost) { best_cost = cost; best = {nt, sp, kps}; }
        }
    }
    return best;
}
static int sg_pick_nt(const CnerfSdGemm *g) { return sg_plan(g).nt; }
static uint32_t sg_splits(const CnerfSdGemm *g, int nt, uint32_t &k_tiles_per_split) {
    (void)nt;
    const SgPlan p = sg_plan(g);
    k_tiles_per_split = p.kps;
    return p.splits;
}

// The LDS-DMA loader addresses A and B through buffer descriptors: 32-bit byte offsets, a 32-bit num_records and 0xFFFFFF00 as the
// "out of range" offset.  An operand whose extent (per batch entry for A) does not fit below that sentinel takes the register-staged
// loader, which uses 64-bit pointers.
static bool sg_glds_fits(const CnerfSdGemm *g) {
    const uint64_t lim = 0xFFFFFF00ull - 4096;
    const uint64_t a_bytes = g->mode == 0 ? ((uint64_t)(g->M - 1) * g->lda + g->K) * 2
                                          : (uint64_t)(g->M / (g->H_out * g->W_out)) * g->H_in * g->W_in * g->Cin * 2;
    const uint64_t b_bytes = ((uint64_t)(g->N - 1) * g->ldb + g->K) * 2;
    return a_bytes < lim && b_bytes < lim;
}
static bool sg_use_glds(const CnerfSdGemm *g) {
    static const int v = cn_tune_env("CNERF_SG_GLDS", 1);        // 0: register-staged loop (round 1)
    return v != 0 && sg_glds_fits(g);
}

#ifdef CNERF_TUNING
// Which problems take the 256 x 128 eight-wave kernel: LDS-DMA-able operands (dense, or a convolution with Cin % 64 == 0), no split-K, and enough
// tiles to give every CU a workgroup (below that the 128-row tiles fill the chip better); a GroupNorm-statistics request needs images of at
// least 128 rows (a 256-row tile then touches at most three: the epilogue's table holds four).
static bool sg_use_big(const CnerfSdGemm *g, const SgPlan &plan) {
    static const int on = cn_tune_env("CNERF_SG_BIG", 0);
    static const int min_tiles = cn_tune_env("CNERF_SG_BIG_TILES", 224), min_k = cn_tune_env("CNERF_SG_BIG_K", 512);
    if (!on || plan.splits > 1 || !sg_use_glds(g)) return false;
    if (g->mode != 0 && (g->Cin % SG_BK) != 0) return false;
    if (g->gn_sums && g->gn_rows < 128) return false;
    if (g->K < (uint32_t)min_k || (g->N % SGB_BN) > 0 && (g->N % SGB_BN) <= 64 && g->N < 1024) return false;      // (a ragged last N tile that is mostly padding)
    const uint64_t tiles = (uint64_t)cn_div_up(g->M, SGB_BM) * cn_div_up(g->N, SGB_BN) * g->batch_outer * g->batch_inner;
    return tiles >= (uint64_t)min_tiles;
}

template <int SCHED>
static void sgb_launch_s(const CnerfSdGemm *g, hipStream_t st) {
    const dim3 grid(cn_div_up(g->M, SGB_BM), cn_div_up(g->N, SGB_BN), g->batch_outer * g->batch_inner);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sd_gemm_big<0, false, SCHED>), hipFuncAttributeMaxDynamicSharedMemorySize, sgb_lds_bytes());
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sd_gemm_big<2, false, SCHED>), hipFuncAttributeMaxDynamicSharedMemorySize, sgb_lds_bytes());
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sd_gemm_big<2, true, SCHED>), hipFuncAttributeMaxDynamicSharedMemorySize, sgb_lds_bytes());
        attr_set = true;
    }
    if (g->mode == 0) hipLaunchKernelGGL((k_sd_gemm_big<0, false, SCHED>), grid, dim3(SGB_THREADS), sgb_lds_bytes(), st, *g);
    else if (g->tstride == 1 && g->ups == 1) hipLaunchKernelGGL((k_sd_gemm_big<2, true, SCHED>), grid, dim3(SGB_THREADS), sgb_lds_bytes(), st, *g);
    else hipLaunchKernelGGL((k_sd_gemm_big<2, false, SCHED>), grid, dim3(SGB_THREADS), sgb_lds_bytes(), st, *g);
}
static void sgb_launch(const CnerfSdGemm *g, hipStream_t st) {
    static const int sched = cn_tune_env("CNERF_SGB_SCHED", 2);
    if (sched == 0) { sgb_launch_s<0>(g, st); return; }
    if (sched == 1) { sgb_launch_s<1>(g, st); return; }
    sgb_launch_s<2>(g, st);
}
#endif  // CNERF_TUNING

template <int AMODE, int NT, bool SPLIT, bool GLDS, bool SIMPLE = false>
static void sg_launch_g(const CnerfSdGemm *g, dim3 grid, hipStream_t st, float *partial, uint32_t kps) {
    if constexpr (GLDS && AMODE == 2 && !SIMPLE) {
        if (g->tstride == 1 && g->ups == 1) { sg_launch_g<AMODE, NT, SPLIT, GLDS, true>(g, grid, st, partial, kps); return; }
    }
    static bool attr_set = false;
    auto kern = k_sd_gemm<AMODE, NT, SPLIT, GLDS, SIMPLE>;
    const uint32_t lds_bytes = GLDS ? sg_lds_bytes_glds(64 * NT) : sg_lds_bytes(64 * NT);
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(SG_THREADS), lds_bytes, st, *g, partial, kps);
}

template <int AMODE, int NT, bool SPLIT>
static void sg_launch(const CnerfSdGemm *g, dim3 grid, hipStream_t st, float *partial, uint32_t kps) {
    if constexpr (AMODE != 1) {
        if (sg_use_glds(g)) { sg_launch_g<AMODE, NT, SPLIT, true>(g, grid, st, partial, kps); return; }
    }
    static bool attr_set = false;
    auto kern = k_sd_gemm<AMODE, NT, SPLIT, false>;
    const uint32_t lds_bytes = sg_lds_bytes(64 * NT);
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(SG_THREADS), lds_bytes, st, *g, partial, kps);
}

template <int AMODE>
static void sg_launch_nt(const CnerfSdGemm *g, int nt, bool split, dim3 grid, hipStream_t st, float *partial, uint32_t kps) {
    if (nt == 2) {
        if (split) sg_launch<AMODE, 2, true>(g, grid, st, partial, kps); else sg_launch<AMODE, 2, false>(g, grid, st, partial, kps);
    } else {
        if (split) sg_launch<AMODE, 1, true>(g, grid, st, partial, kps); else sg_launch<AMODE, 1, false>(g, grid, st, partial, kps);
    }
}

extern "C" {

int cnerf_sd_gemm_workspace_bytes(const CnerfSdGemm *g, uint64_t *bytes) {
    if (!bytes) return CNERF_ENULL;
    int rc = sg_check(g);
    if (rc) return rc;
    uint32_t kps;
    const uint32_t splits = sg_splits(g, sg_pick_nt(g), kps);
    *bytes = splits > 1 ? (uint64_t)splits * g->M * g->N * sizeof(float) : 0;
    return CNERF_OK;
}

int cnerf_sd_gemm(const CnerfSdGemm *g, void *workspace, uint64_t workspace_bytes, void *stream) {
    int rc = sg_check(g);
    if (rc) return rc;
    hipStream_t st = CN_STREAM(stream);
    const SgPlan plan = sg_plan(g);
    const int nt = plan.nt;
    uint32_t kps = plan.kps;
    uint32_t splits = plan.splits;
    if (splits > 1 && (!workspace || workspace_bytes < (uint64_t)splits * g->M * g->N * sizeof(float))) {
        splits = 1;
        kps = cn_div_up(g->K, SG_BK);
    }
    if (splits > 1 && g->gn_sums && (uint64_t)(g->M / g->gn_rows) * g->gn_groups > SG_EPI_GN_MAX) {      // the split-K tail's statistics table (images x groups)
        splits = 1;
        kps = cn_div_up(g->K, SG_BK);
    }
    const bool split = splits > 1;
#ifdef CNERF_TUNING
    if (!split && sg_use_big(g, plan)) {
        sgb_launch(g, st);
        return cn_launch_status();
    }
#endif
    const dim3 grid(cn_div_up(g->M, SG_BM), cn_div_up(g->N, 64 * nt), split ? splits : g->batch_outer * g->batch_inner);
    const int amode = g->mode == 0 ? 0 : ((g->Cin % SG_BK) == 0 ? 2 : 1);
    float *partial = reinterpret_cast<float *>(workspace);
    if (amode == 0) sg_launch_nt<0>(g, nt, split, grid, st, partial, kps);
    else if (amode == 1) sg_launch_nt<1>(g, nt, split, grid, st, partial, kps);
    else sg_launch_nt<2>(g, nt, split, grid, st, partial, kps);
    if (split && sg_ln_admissible(g)) {
        const dim3 eg(cn_div_up(g->M, 4));
        const float *pw = reinterpret_cast<const float *>(workspace);
        const uint32_t cpl = cn_div_up(g->N / 8, 64);
        if (cpl == 1) hipLaunchKernelGGL(k_sd_gemm_splitk_epilogue_ln<1>, eg, dim3(256), 0, st, *g, pw, splits);
        else if (cpl == 2) hipLaunchKernelGGL(k_sd_gemm_splitk_epilogue_ln<2>, eg, dim3(256), 0, st, *g, pw, splits);
        else hipLaunchKernelGGL(k_sd_gemm_splitk_epilogue_ln<4>, eg, dim3(256), 0, st, *g, pw, splits);
    } else if (split) {
        const size_t total = (size_t)g->M * g->N;
        uint32_t eb = (uint32_t)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
        if (g->gn_sums) {                                                   // statistics: the grid is (slabs of four groups) x (row chunks), see the kernel
            const uint32_t n_slabs = (g->gn_groups & 3u) == 0 ? g->gn_groups / 4 : 1u;
            uint32_t row_chunks = cn_div_up(eb, n_slabs);
            if (row_chunks > g->M) row_chunks = g->M;
            eb = n_slabs * (row_chunks ? row_chunks : 1u);
        }
        hipLaunchKernelGGL(k_sd_gemm_splitk_epilogue, dim3(eb), dim3(256), 0, st, *g, reinterpret_cast<const float *>(workspace), splits);
    }
    return cn_launch_status();
}

int cnerf_sd_gemm_serves_ln(const CnerfSdGemm *g, int *yes) {
    if (!yes) return CNERF_ENULL;
    *yes = 0;
    int rc = sg_check(g);
    if (rc) return rc;
    uint32_t splits = sg_plan(g).splits;
    if (splits > 1 && g->gn_sums && (uint64_t)(g->M / g->gn_rows) * g->gn_groups > SG_EPI_GN_MAX) splits = 1;
    *yes = (splits > 1 && sg_ln_admissible(g)) ? 1 : 0;
    return CNERF_OK;
}

}  // extern "C"
