// Split-K MFMA GEMM for weight gradients dW = dz . a^T over [row][sample] operand matrices (shared by field_bwd.hip and mlp.hip).
#pragma once
#include "field_common.h"

// ------------------------------------------------------------------------------------------------ weight-gradient kernel
struct DwJob {
    uint32_t z_row, M, a_row, N;         // rows of dz / rows of the layer input inside the workspace, and their counts
    uint32_t dst_off, dst_stride, dst_col0, net;   // destination inside grad_params_{net 0, den 1, rgb 2}
    uint32_t tile0, nt_n;                // first global tile index of this job, tiles along N
};
#define FLD_MAX_JOBS 8
struct DwPlan {
    DwJob job[FLD_MAX_JOBS];
    uint32_t n_jobs, n_tiles, k_tiles_per_split;     // K split in units of 32-sample tiles
};

// One workgroup = one 32x32 tile of one layer's dW over one K split.  Per step it stages a [32 rows x 128 samples] block of dz
// and of the layer input into LDS with fully coalesced loads (a wave instruction reads 4 rows x 256 contiguous bytes), then
// each wave contracts its own 32 of the 128 samples; partial tiles are reduced through LDS and added with one float atomic per
// weight per split.  (Reading the MFMA fragments straight from HBM touches 32 rows x 32 B per instruction: 4x slower.)
#define DW_KB 128                       // samples per staged block (4 waves x 32)
template <bool H>
__global__ void __launch_bounds__(FLD_THREADS) k_field_bwd_dw(const void *__restrict__ ws_, size_t ld, uint32_t n_ktiles, DwPlan plan,
                                                              float *__restrict__ g_net, float *__restrict__ g_den, float *__restrict__ g_rgb) {
    using PR = Prec<H>;
    using elem_t = typename PR::elem_t;
    using frag_t = typename PR::frag_t;
    constexpr uint32_t ROWB = DW_KB * sizeof(elem_t);            // bytes per staged row: 256 (fp16) / 512 (fp32)
    constexpr uint32_t PAD = 16;                                  // row padding: keeps the per-lane 16-byte fragment reads off one bank group
    constexpr uint32_t RSTR = ROWB + PAD;
    __shared__ __attribute__((aligned(16))) unsigned char stage[2][32 * RSTR];      // [Z | A][row][samples]
    __shared__ float red[FLD_WAVES][32 * 32];
    const unsigned char *ws = reinterpret_cast<const unsigned char *>(ws_);
    const uint32_t tile = blockIdx.x, split = blockIdx.y;
    uint32_t jid = 0;
#pragma unroll
    for (uint32_t j = 1; j < FLD_MAX_JOBS; j++)
        if (j < plan.n_jobs && tile >= plan.job[j].tile0) jid = j;
    const DwJob jb = plan.job[jid];
    const uint32_t lt = tile - jb.tile0, mt = lt / jb.nt_n, nt = lt % jb.nt_n;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hi = lane >> 5, li = lane & 31;

    const uint32_t kt0 = split * plan.k_tiles_per_split, kt1 = min(kt0 + plan.k_tiles_per_split, n_ktiles);
    cn_f16v acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
    constexpr uint32_t CH_PER_ROW = ROWB / 16;                   // 16-byte chunks per staged row
    constexpr uint32_t CHUNKS = 32 * CH_PER_ROW;                 // per operand
    for (uint32_t kt = kt0; kt < kt1; kt += DW_KB / 32) {
        const size_t k0 = (size_t)kt * 32;
        const uint32_t k_valid = min((uint32_t)DW_KB, (kt1 - kt) * 32);
        // ---- coalesced staging of both operands
        for (uint32_t c = threadIdx.x; c < 2 * CHUNKS; c += FLD_THREADS) {
            const uint32_t op = c / CHUNKS, cc = c % CHUNKS, row = cc / CH_PER_ROW, ch = cc % CH_PER_ROW;
            const uint32_t grow = 32 * (op ? nt : mt) + row, nrows = op ? jb.N : jb.M;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (grow < nrows && ch * (16 / sizeof(elem_t)) < k_valid)
                v = *reinterpret_cast<const uint4 *>(ws + ((size_t)((op ? jb.a_row : jb.z_row) + grow) * ld + k0) * sizeof(elem_t) + ch * 16);
            *reinterpret_cast<uint4 *>(&stage[op][row * RSTR + ch * 16]) = v;
        }
        __syncthreads();
        // ---- this wave's 32 samples: 32 / KS MFMA K-steps
        const unsigned char *zrow = &stage[0][li * RSTR + wave * 32 * sizeof(elem_t)];
        const unsigned char *arow = &stage[1][li * RSTR + wave * 32 * sizeof(elem_t)];
#pragma unroll
        for (int s = 0; s < 32 / PR::KS; s++) {
            const uint32_t off = (PR::KS * s + PR::J * hi) * sizeof(elem_t);
            const frag_t a = *reinterpret_cast<const frag_t *>(zrow + off);
            const frag_t b = *reinterpret_cast<const frag_t *>(arow + off);
            acc = PR::mfma(a, b, acc);
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 16; r++) red[wave][fld_rho(r, hi) * 32 + li] = acc[r];
    __syncthreads();
    float *dst = (jb.net == 0 ? g_net : (jb.net == 1 ? g_den : g_rgb)) + jb.dst_off;
    for (uint32_t i = threadIdx.x; i < 32 * 32; i += FLD_THREADS) {
        const uint32_t row = 32 * mt + i / 32, col = 32 * nt + (i & 31);
        if (row < jb.M && col < jb.N) {
            const float v = red[0][i] + red[1][i] + red[2][i] + red[3][i];
            if (v != 0.0f) unsafeAtomicAdd(&dst[(size_t)row * jb.dst_stride + jb.dst_col0 + col], v);
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side
static inline void fb_add_job(DwPlan &pl, uint32_t z_row, uint32_t M, uint32_t a_row, uint32_t N, uint32_t net, uint32_t dst_off, uint32_t dst_stride,
                       uint32_t dst_col0) {
    DwJob &j = pl.job[pl.n_jobs++];
    j.z_row = z_row; j.M = M; j.a_row = a_row; j.N = N; j.net = net; j.dst_off = dst_off; j.dst_stride = dst_stride; j.dst_col0 = dst_col0;
    j.tile0 = pl.n_tiles;
    j.nt_n = cn_div_up(N, 32);
    pl.n_tiles += cn_div_up(M, 32) * j.nt_n;
}


static inline size_t fb_ld(uint32_t P_) { return ((size_t)P_ + 63) / 64 * 64; }
