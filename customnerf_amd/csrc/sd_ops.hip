// Normalisation / softmax / element-wise kernels of the SDS path (everything around the GEMMs of sd_gemm.hip).
// All of them are HBM-stream bound: 16-byte accesses (8 halfs) per lane, float32 arithmetic, one pass over the data wherever the
// row fits in registers.  Layout: NHWC / [tokens, channels] half.
#include "common.h"
#include "../../include/customnerf_sd.h"

typedef _Float16 so_h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ so_h8 so_ld8(const _Float16 *p) { return *reinterpret_cast<const so_h8 *>(p); }
__device__ __forceinline__ void so_st8(_Float16 *p, so_h8 v) { *reinterpret_cast<so_h8 *>(p) = v; }
__device__ __forceinline__ float so_sigmoid(float z) { return 1.0f / (1.0f + __expf(-z)); }
__device__ __forceinline__ float so_wave_max(float v) {
#pragma unroll
    for (int off = CN_WAVE / 2; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, CN_WAVE));
    return v;
}

// ------------------------------------------------------------------------------------------------ GroupNorm
// Thread t of a block owns one 16-byte channel chunk (8 channels, <= 2 groups when C/G >= 4) and walks down the rows of the
// block's slab, so its partial sums live in registers; one LDS + one global atomic per (thread, group) at the end.
// The statistics are kept as 64-BIT FIXED POINT (2^-20 units, include/customnerf_sd.h CNERF_SD_GN_FRAC_BITS): a thread's float32 partial
// sum is a deterministic function of its slab, it is rounded to the fixed-point grid once, and everything after that is integer
// addition — exact and order-independent, so the atomics no longer make the result depend on arrival order (round 4: the edit step is
// bit-reproducible; float atomics left it deterministic to ~1e-3 only).  Range +-8.8e12, resolution 1e-6 per partial.
#define GN_THREADS 256
#define GN_MAX_G 64
#include "sd_gn_fix.h"

struct GnGeom {
    uint32_t nchunks, cols_per_pass, rows_per_pass, cg;
};
__device__ __forceinline__ GnGeom gn_geom(uint32_t C, uint32_t G) {
    GnGeom q;
    q.nchunks = C / 8;
    q.cols_per_pass = q.nchunks < GN_THREADS ? q.nchunks : GN_THREADS;
    q.rows_per_pass = GN_THREADS / q.cols_per_pass;
    q.cg = C / G;
    return q;
}

// Measured and dropped (round 3): statistics + apply in ONE launch for the UNet's tensors — a workgroup per (image, lcm(8, C/G) channels) over all rows,
// slab kept in registers, so the statistics never leave the workgroup.  With 16-64 workgroups the slab moves at one CU's pace: 55 us for 2 x 4096 x 320
// against 2 x 8 us for the two launches; restricted to the tensors of at most 4 rows per thread (16 x 16 and 8 x 8 latents) the edit step gains 0.4 %.
// (scratch/gn_bench.py times the shapes.)
// MODE 0: forward statistics (sum x, sum x^2).  MODE 1: backward statistics (sum g, sum g xhat), g = dy * act'(z) * gamma.
template <int MODE>
__global__ void __launch_bounds__(GN_THREADS) k_gn_stats(const _Float16 *__restrict__ x, const _Float16 *__restrict__ dy, const float *__restrict__ gamma,
                                                         const float *__restrict__ beta, const long long *__restrict__ fsums, uint32_t HW, uint32_t C,
                                                         uint32_t G, float eps, int silu, uint32_t rows_per_block, long long *__restrict__ out) {
    __shared__ long long acc[GN_MAX_G][2];
    const GnGeom q = gn_geom(C, G);
    const uint32_t b = blockIdx.y, r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, HW);
    for (uint32_t i = threadIdx.x; i < G * 2; i += GN_THREADS) acc[i >> 1][i & 1] = 0;
    __syncthreads();
    const uint32_t tcol = threadIdx.x % q.cols_per_pass, trow = threadIdx.x / q.cols_per_pass;
    const float inv_n = 1.0f / ((float)HW * (float)q.cg);
    if (trow < q.rows_per_pass) {
        for (uint32_t col = tcol; col < q.nchunks; col += q.cols_per_pass) {
            const uint32_t c0 = col * 8, g_lo = c0 / q.cg, g_hi = (c0 + 7) / q.cg;
            const uint32_t split = (g_lo + 1) * q.cg - c0;              // channels [0, split) of the chunk belong to g_lo
            float s[2][2] = {{0, 0}, {0, 0}};
            float mean[2] = {0, 0}, rstd[2] = {0, 0}, gam[8], bet[8];
            if (MODE == 1) {
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const uint32_t g = k ? g_hi : g_lo;
                    const float m = gn_unfix(fsums[((size_t)b * G + g) * 2]) * inv_n;
                    const float v = gn_unfix(fsums[((size_t)b * G + g) * 2 + 1]) * inv_n - m * m;
                    mean[k] = m;
                    rstd[k] = (v != v) ? v : rsqrtf(fmaxf(v, 0.0f) + eps);              // (a poisoned sum of squares must not vanish in fmaxf)
                }
#pragma unroll
                for (int e = 0; e < 8; e++) { gam[e] = gamma[c0 + e]; bet[e] = beta[c0 + e]; }
            }
#pragma unroll 4
            for (uint32_t r = r0 + trow; r < r1; r += q.rows_per_pass) {      // unrolled: four independent 16-byte loads in flight per thread
                const size_t off = ((size_t)b * HW + r) * C + c0;
                const so_h8 xv = so_ld8(x + off);
                so_h8 dv;
                if (MODE == 1) dv = so_ld8(dy + off);
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int k = ((uint32_t)e < split) ? 0 : 1;
                    const float xf = (float)xv[e];
                    if (MODE == 0) {
                        s[k][0] += xf;
                        s[k][1] += xf * xf;
                    } else {
                        const float xh = (xf - mean[k]) * rstd[k];
                        float gq = (float)dv[e];
                        if (silu) {
                            const float z = xh * gam[e] + bet[e], sg = so_sigmoid(z);
                            gq *= sg * (1.0f + z * (1.0f - sg));
                        }
                        gq *= gam[e];
                        s[k][0] += gq;
                        s[k][1] += gq * xh;
                    }
                }
            }
            gn_add(&acc[g_lo][0], s[0][0]);
            gn_add(&acc[g_lo][1], s[0][1]);
            if (g_hi != g_lo) {
                gn_add(&acc[g_hi][0], s[1][0]);
                gn_add(&acc[g_hi][1], s[1][1]);
            }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < G * 2; i += GN_THREADS) gn_add_fixed(&out[(size_t)b * G * 2 + i], acc[i >> 1][i & 1]);
}

// MODE 0: y = act(xhat gamma + beta).  MODE 1: dx = rstd (g - S1/n - xhat S2/n).
// Same geometry as k_gn_stats: a thread owns one 16-byte channel chunk and walks down the rows of its slab, so gamma / beta and the
// (<= 2) groups' statistics are loaded once per thread instead of once per chunk (the flat-index version spent its time on index
// divisions and parameter reloads: 1.4 TB/s on the VAE's 67 MB tensors).
template <int MODE>
__global__ void __launch_bounds__(GN_THREADS) k_gn_apply(const _Float16 *__restrict__ x, const _Float16 *__restrict__ dy, const float *__restrict__ gamma,
                                                         const float *__restrict__ beta, const long long *__restrict__ fsums, const long long *__restrict__ bsums,
                                                         uint32_t HW, uint32_t C, uint32_t G, float eps, int silu, uint32_t rows_per_block,
                                                         _Float16 *__restrict__ out, const _Float16 *__restrict__ res = nullptr) {
    // res (MODE 1): a second gradient arriving at x — the skip connection of a residual block — added before the one rounding to half
    const GnGeom q = gn_geom(C, G);
    const uint32_t b = blockIdx.y, r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, HW);
    const uint32_t tcol = threadIdx.x % q.cols_per_pass, trow = threadIdx.x / q.cols_per_pass;
    if (trow >= q.rows_per_pass) return;
    const float inv_n = 1.0f / ((float)HW * (float)q.cg);
    for (uint32_t col = tcol; col < q.nchunks; col += q.cols_per_pass) {
        const uint32_t c0 = col * 8, g_lo = c0 / q.cg, g_hi = (c0 + 7) / q.cg, split = (g_lo + 1) * q.cg - c0;
        float mean[8], rstd[8], ga[8], be[8], s1[8], s2[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const uint32_t g = ((uint32_t)e < split) ? g_lo : g_hi;
            const float m = gn_unfix(fsums[((size_t)b * G + g) * 2]) * inv_n;
            const float v = gn_unfix(fsums[((size_t)b * G + g) * 2 + 1]) * inv_n - m * m;
            mean[e] = m;
            rstd[e] = (v != v) ? v : rsqrtf(fmaxf(v, 0.0f) + eps);              // (a poisoned sum of squares must not vanish in fmaxf)
            ga[e] = gamma[c0 + e];
            be[e] = beta[c0 + e];
            s1[e] = s2[e] = 0.0f;
            if (MODE == 1) {
                s1[e] = gn_unfix(bsums[((size_t)b * G + g) * 2]) * inv_n;
                s2[e] = gn_unfix(bsums[((size_t)b * G + g) * 2 + 1]) * inv_n;
            }
        }
#pragma unroll 4
        for (uint32_t r = r0 + trow; r < r1; r += q.rows_per_pass) {
            const size_t off = ((size_t)b * HW + r) * C + c0;
            const so_h8 xv = so_ld8(x + off);
            so_h8 dv, o, rv = {0, 0, 0, 0, 0, 0, 0, 0};
            if (MODE == 1) dv = so_ld8(dy + off);
            if (MODE == 1 && res) rv = so_ld8(res + off);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float xh = ((float)xv[e] - mean[e]) * rstd[e];
                const float z = xh * ga[e] + be[e];
                if (MODE == 0) {
                    o[e] = (_Float16)(silu ? z * so_sigmoid(z) : z);
                } else {
                    float gq = (float)dv[e];
                    if (silu) {
                        const float sg = so_sigmoid(z);
                        gq *= sg * (1.0f + z * (1.0f - sg));
                    }
                    gq *= ga[e];
                    o[e] = (_Float16)(rstd[e] * (gq - s1[e] - xh * s2[e]) + (float)rv[e]);
                }
            }
            so_st8(out + off, o);
        }
    }
}

// ------------------------------------------------------------------------------------------------ LayerNorm (one wave per row)
template <int CPL>
__global__ void __launch_bounds__(256) k_layernorm(const _Float16 *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ beta,
                                                   uint32_t rows, uint32_t C, float eps, _Float16 *__restrict__ y) {
    const uint32_t lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const uint32_t nchunks = C / 8;
    float v[CPL][8];
    float sum = 0.0f;
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        const uint32_t col = lane + 64 * c;
        if (col < nchunks) {
            const so_h8 h = so_ld8(x + (size_t)row * C + col * 8);
#pragma unroll
            for (int e = 0; e < 8; e++) { v[c][e] = (float)h[e]; sum += v[c][e]; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++) v[c][e] = 0.0f;
        }
    }
    const float mean = cn_wave_sum(sum) / (float)C;
    float sq = 0.0f;
#pragma unroll
    for (int c = 0; c < CPL; c++)
        if (lane + 64 * c < nchunks) {
#pragma unroll
            for (int e = 0; e < 8; e++) { const float d = v[c][e] - mean; sq += d * d; }
        }
    const float rstd = rsqrtf(cn_wave_sum(sq) / (float)C + eps);
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        const uint32_t col = lane + 64 * c;
        if (col < nchunks) {
            so_h8 o;
#pragma unroll
            for (int e = 0; e < 8; e++) o[e] = (_Float16)((v[c][e] - mean) * rstd * gamma[col * 8 + e] + beta[col * 8 + e]);
            so_st8(y + (size_t)row * C + col * 8, o);
        }
    }
}

// ------------------------------------------------------------------------------------------------ softmax (one wave per row, row in registers)
template <int CPL, bool BWD>
__global__ void __launch_bounds__(256) k_softmax(const _Float16 *__restrict__ P, _Float16 *__restrict__ S, uint64_t rows, uint32_t cols, uint32_t ld) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t row = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const uint32_t nchunks = ld / 8;
    _Float16 *s = S + row * ld;
    float v[CPL][8];
    if (!BWD) {
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            const uint32_t col = lane + 64 * c;
            so_h8 h;
            if (col < nchunks) h = so_ld8(s + col * 8);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                v[c][e] = (col < nchunks && col * 8 + e < cols) ? (float)h[e] : -INFINITY;
                mx = fmaxf(mx, v[c][e]);
            }
        }
        mx = so_wave_max(mx);
        float sum = 0.0f;
#pragma unroll
        for (int c = 0; c < CPL; c++)
#pragma unroll
            for (int e = 0; e < 8; e++) {
                v[c][e] = __expf(v[c][e] - mx);          // exp(-inf) = 0 for the padding
                sum += v[c][e];
            }
        const float inv = 1.0f / cn_wave_sum(sum);
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            const uint32_t col = lane + 64 * c;
            if (col < nchunks) {
                so_h8 o;
#pragma unroll
                for (int e = 0; e < 8; e++) o[e] = (_Float16)(v[c][e] * inv);
                so_st8(s + col * 8, o);
            }
        }
    } else {
        const _Float16 *p = P + row * ld;
        float pv[CPL][8];
        float dot = 0.0f;
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            const uint32_t col = lane + 64 * c;
            so_h8 h, hp;
            if (col < nchunks) { h = so_ld8(s + col * 8); hp = so_ld8(p + col * 8); }
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const bool ok = col < nchunks && col * 8 + e < cols;
                v[c][e] = ok ? (float)h[e] : 0.0f;
                pv[c][e] = ok ? (float)hp[e] : 0.0f;
                dot += v[c][e] * pv[c][e];
            }
        }
        dot = cn_wave_sum(dot);
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            const uint32_t col = lane + 64 * c;
            if (col < nchunks) {
                so_h8 o;
#pragma unroll
                for (int e = 0; e < 8; e++) o[e] = (_Float16)(pv[c][e] * (v[c][e] - dot));
                so_st8(s + col * 8, o);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ element-wise
__global__ void __launch_bounds__(256) k_geglu(const _Float16 *__restrict__ x, uint64_t rows, uint32_t C, _Float16 *__restrict__ y) {
    const uint32_t nchunks = C / 8;
    const uint32_t total = (uint32_t)(rows * nchunks);                 // < 2^32 (checked on the host): 32-bit index divisions
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint32_t row = i / nchunks;
        const uint32_t col = i - row * nchunks;
        const so_h8 a = so_ld8(x + (size_t)row * 2 * C + col * 8), g = so_ld8(x + (size_t)row * 2 * C + C + col * 8);
        so_h8 o;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float gf = (float)g[e];
            o[e] = (_Float16)((float)a[e] * (0.5f * gf * (1.0f + erff(gf * 0.70710678118654752f))));
        }
        so_st8(y + (size_t)row * C + col * 8, o);
    }
}

template <int OP>     // 0 add, 1 silu
__global__ void __launch_bounds__(256) k_ew(const _Float16 *__restrict__ a, const _Float16 *__restrict__ b, uint64_t n8, uint64_t n, _Float16 *__restrict__ y) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const so_h8 av = so_ld8(a + i * 8);
        so_h8 o;
        if (OP == 0) {
            const so_h8 bv = so_ld8(b + i * 8);
#pragma unroll
            for (int e = 0; e < 8; e++) o[e] = (_Float16)((float)av[e] + (float)bv[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++) { const float z = (float)av[e]; o[e] = (_Float16)(z * so_sigmoid(z)); }
        }
        so_st8(y + i * 8, o);
    }
    if (blockIdx.x == 0)
        for (size_t i = n8 * 8 + threadIdx.x; i < n; i += blockDim.x) {
            const float z = (float)a[i];
            y[i] = (_Float16)(OP == 0 ? z + (float)b[i] : z * so_sigmoid(z));
        }
}

// gn_sums != NULL: also the GroupNorm statistics of the OUTPUT (the UNet's up blocks normalise a channel concat first thing: round 6 — their
// twelve k_gn_stats launches per forward ride on this copy).  A thread owns 8 consecutive output channels = at most two groups (host: >= 8
// channels per group); per-workgroup LDS table [images x groups][2] in 64-bit fixed point (sd_gn_fix.h), one global atomic per touched entry.
#define SO_CONCAT_GN_MAX 512
__global__ void __launch_bounds__(256) k_concat(const _Float16 *__restrict__ a, const _Float16 *__restrict__ b, uint64_t rows, uint32_t C1, uint32_t C2,
                                                _Float16 *__restrict__ y, long long *__restrict__ gn_sums, uint32_t gn_groups, uint32_t gn_rows) {
    __shared__ long long gn_tab[SO_CONCAT_GN_MAX][2];
    const bool do_gn = gn_sums != nullptr;
    const uint32_t n1 = C1 / 8, nc = (C1 + C2) / 8;
    const uint32_t total = (uint32_t)(rows * nc);
    const uint32_t cg = do_gn ? (C1 + C2) / gn_groups : 1u, entries = do_gn ? (uint32_t)(rows / gn_rows) * gn_groups : 0u;
    if (do_gn) {
        for (uint32_t i = threadIdx.x; i < entries * 2; i += 256) (&gn_tab[0][0])[i] = 0;
        __syncthreads();
    }
    // statistics: a workgroup owns a TILE of the output — a slab of four whole groups x a range of rows (gridDim = slabs x row chunks) — so that
    // its table flush is a handful of entries (k_sd_gemm_splitk_epilogue, sd_gemm.hip, has the measurements behind this)
    const bool tiled = do_gn && (gn_groups & 3u) == 0 && (cg & 1u) == 0;
    const uint32_t n_slabs = tiled ? gn_groups / 4 : 1u, slab_chunks = tiled ? cg / 2 : nc;
    const uint32_t row_chunks = do_gn ? gridDim.x / n_slabs : 1u, rpb = do_gn ? (uint32_t)((rows + row_chunks - 1) / row_chunks) : (uint32_t)rows;
    const uint32_t slab = do_gn ? blockIdx.x % n_slabs : 0u, row0 = do_gn ? (blockIdx.x / n_slabs) * rpb : 0u;
    const uint32_t rows_in = do_gn ? (row0 < rows ? min(rpb, (uint32_t)rows - row0) : 0u) : (uint32_t)rows;
    const uint32_t my = do_gn ? rows_in * slab_chunks : total;
    const uint32_t j_lo = do_gn ? threadIdx.x : blockIdx.x * blockDim.x + threadIdx.x, j_step = do_gn ? blockDim.x : gridDim.x * blockDim.x;
    for (uint32_t j = j_lo; j < my; j += j_step) {
        uint32_t i = j;
        if (do_gn) {
            const uint32_t r = j / slab_chunks, c8 = j - r * slab_chunks;
            i = (row0 + r) * nc + slab * slab_chunks + c8;
        }
        const uint32_t row = i / nc;
        const uint32_t col = i - row * nc;
        const so_h8 v = col < n1 ? so_ld8(a + (size_t)row * C1 + col * 8) : so_ld8(b + (size_t)row * C2 + (col - n1) * 8);
        so_st8(y + (size_t)i * 8, v);
        if (do_gn) {
            const uint32_t n = col * 8, g_lo = n / cg, split_c = (g_lo + 1) * cg - n;      // channels [0, split_c) of the eight belong to g_lo
            float s0 = 0.0f, q0 = 0.0f, s1 = 0.0f, q1 = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float xf = (float)v[e];
                if ((uint32_t)e < split_c) { s0 += xf; q0 += xf * xf; } else { s1 += xf; q1 += xf * xf; }
            }
            long long *t0 = gn_tab[(row / gn_rows) * gn_groups + g_lo];
            gn_add(t0, s0); gn_add(t0 + 1, q0);
            if (split_c < 8) { gn_add(t0 + 2, s1); gn_add(t0 + 3, q1); }
        }
    }
    if (do_gn) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < entries * 2; i += 256) gn_add_fixed(gn_sums + i, (&gn_tab[0][0])[i]);
    }
}

// 64 x 64 tiles through LDS; dst[z][c][r] = src[z][r][c]
__global__ void __launch_bounds__(256) k_transpose(const _Float16 *__restrict__ src, _Float16 *__restrict__ dst, uint32_t rows, uint32_t cols, uint32_t lds_,
                                                   uint32_t ldd, uint64_t ss, uint64_t sd) {
    __shared__ _Float16 tile[64][66];
    const _Float16 *s = src + ss * blockIdx.z;
    _Float16 *d = dst + sd * blockIdx.z;
    const uint32_t r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const uint32_t tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (uint32_t r = ty; r < 64; r += 4) {
        const uint32_t rr = r0 + r, cc = c0 + tx;
        tile[r][tx] = (rr < rows && cc < cols) ? s[(size_t)rr * lds_ + cc] : (_Float16)0;
    }
    __syncthreads();
    for (uint32_t c = ty; c < 64; c += 4) {
        const uint32_t cc = c0 + c, rr = r0 + tx;
        if (cc < cols && rr < ldd) d[(size_t)cc * ldd + rr] = tile[tx][c];       // rr in [rows, ldd): zeros from the guarded load
    }
}

// ------------------------------------------------------------------------------------------------ image front-end
__device__ __forceinline__ void so_bilin(uint32_t o, uint32_t n_in, uint32_t n_out, uint32_t &i0, uint32_t &i1, float &lam) {
    const float scale = (float)n_in / (float)n_out;
    float src = ((float)o + 0.5f) * scale - 0.5f;           // torch upsample_bilinear2d, align_corners = False
    if (src < 0.0f) src = 0.0f;
    i0 = (uint32_t)src;
    if (i0 > n_in - 1) i0 = n_in - 1;
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    lam = src - (float)i0;
}

__global__ void __launch_bounds__(256) k_img_fwd(const float *__restrict__ img, uint32_t B, uint32_t Hi, uint32_t Wi, uint32_t Ho, uint32_t Wo,
                                                 _Float16 *__restrict__ out) {
    const size_t total = (size_t)B * Ho * Wo;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t ox = (uint32_t)(i % Wo), oy = (uint32_t)((i / Wo) % Ho), b = (uint32_t)(i / ((size_t)Wo * Ho));
        uint32_t y0, y1, x0, x1;
        float ly, lx;
        so_bilin(oy, Hi, Ho, y0, y1, ly);
        so_bilin(ox, Wi, Wo, x0, x1, lx);
        so_h8 o = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float *p = img + ((size_t)b * 3 + c) * Hi * Wi;
            const float top = p[(size_t)y0 * Wi + x0] * (1.0f - lx) + p[(size_t)y0 * Wi + x1] * lx;
            const float bot = p[(size_t)y1 * Wi + x0] * (1.0f - lx) + p[(size_t)y1 * Wi + x1] * lx;
            o[c] = (_Float16)(2.0f * (top * (1.0f - ly) + bot * ly) - 1.0f);
        }
        so_st8(out + i * 8, o);
    }
}

// ---- CLIP image front-end (nerf/clip.py:13-17: T.Resize(224, BICUBIC, antialias=None) -> T.CenterCrop(224) -> T.Normalize) -------------------
// Cubic convolution coefficients of torch's upsample_bicubic2d (A = -0.75), taps x0-1 .. x0+2, border indices clamped.
__device__ __forceinline__ void so_cubic(float t, float (&w)[4]) {
    const float A = -0.75f;
    const float x0 = t + 1.0f, x3 = 2.0f - t, x2 = 1.0f - t;
    w[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
    w[1] = ((A + 2.0f) * t - (A + 3.0f)) * t * t + 1.0f;
    w[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
    w[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
}

__global__ void __launch_bounds__(256) k_clip_preprocess(const float *__restrict__ img, uint32_t B, uint32_t Hi, uint32_t Wi, uint32_t Hr, uint32_t Wr,
                                                         uint32_t S, uint32_t top, uint32_t left, float3 mean, float3 inv_std, float *__restrict__ out) {
    const size_t total = (size_t)B * 3 * S * S;
    const float sy = (float)Hi / (float)Hr, sx = (float)Wi / (float)Wr;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t ox = (uint32_t)(i % S), oy = (uint32_t)((i / S) % S), c = (uint32_t)((i / ((size_t)S * S)) % 3), b = (uint32_t)(i / ((size_t)3 * S * S));
        const float fy = sy * ((float)(oy + top) + 0.5f) - 0.5f, fx = sx * ((float)(ox + left) + 0.5f) - 0.5f;
        const float yf = floorf(fy), xf = floorf(fx);
        float wy[4], wx[4];
        so_cubic(fy - yf, wy);
        so_cubic(fx - xf, wx);
        const int y0 = (int)yf, x0 = (int)xf;
        const float *p = img + ((size_t)b * 3 + c) * Hi * Wi;
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int yy = min(max(y0 - 1 + j, 0), (int)Hi - 1);
            float row = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int xx = min(max(x0 - 1 + k, 0), (int)Wi - 1);
                row += p[(size_t)yy * Wi + xx] * wx[k];
            }
            acc += row * wy[j];
        }
        const float m = c == 0 ? mean.x : (c == 1 ? mean.y : mean.z), is = c == 0 ? inv_std.x : (c == 1 ? inv_std.y : inv_std.z);
        out[i] = (acc - m) * is;
    }
}

// [B,3,S,S] float32 -> patch rows [B, (S/P)^2, P*P*3] half with the row laid out (kh, kw, c): the patch-embedding convolution
// (kernel = stride = P, no bias) then is one dense GEMM against weights packed [Cout][kh][kw][3].
__global__ void __launch_bounds__(256) k_patchify(const float *__restrict__ x, uint32_t B, uint32_t S, uint32_t P, _Float16 *__restrict__ out) {
    const uint32_t G = S / P;
    const size_t total = (size_t)B * 3 * S * S;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t c = (uint32_t)(i % 3);
        size_t r = i / 3;
        const uint32_t kw = (uint32_t)(r % P); r /= P;
        const uint32_t kh = (uint32_t)(r % P); r /= P;
        const uint32_t px = (uint32_t)(r % G); r /= G;
        const uint32_t py = (uint32_t)(r % G);
        const uint32_t b = (uint32_t)(r / G);
        out[i] = (_Float16)x[(((size_t)b * 3 + c) * S + (size_t)py * P + kh) * S + (size_t)px * P + kw];
    }
}

// Input-gradient of k_img_fwd as a GATHER: one thread per input pixel and channel sums, in a fixed order, the output pixels whose bilinear
// taps touch it (candidates from the inverse map, membership and weights from so_bilin itself, so forward and backward agree bit for bit on
// who reads whom).  The scatter form it replaces issued 12 float atomics per output pixel: 74 us for a 512 x 512 image and a gradient
// that changed in the last bits from run to run.
__device__ __forceinline__ void so_bilin_range(uint32_t i, uint32_t n_in, uint32_t n_out, uint32_t &lo, uint32_t &hi) {
    // outputs o with src(o) in (i - 1, i + 1) (+- one output of slack for rounding; the clamped borders reach to the ends)
    const float inv = (float)n_out / (float)n_in;
    const float a = ((float)i - 1.0f + 0.5f) * inv - 0.5f, b = ((float)i + 1.0f + 0.5f) * inv - 0.5f;
    lo = (i == 0 || a < 1.0f) ? 0u : (uint32_t)a - 1u;
    hi = (i >= n_in - 1) ? n_out - 1 : min((uint32_t)(b < 0.0f ? 0.0f : b) + 2u, n_out - 1);
}
__global__ void __launch_bounds__(256) k_img_bwd(const _Float16 *__restrict__ d_out, uint32_t B, uint32_t Hi, uint32_t Wi, uint32_t Ho, uint32_t Wo,
                                                 float *__restrict__ d_img) {
    const size_t total = (size_t)B * Hi * Wi;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t x = (uint32_t)(i % Wi), y = (uint32_t)((i / Wi) % Hi), b = (uint32_t)(i / ((size_t)Wi * Hi));
        uint32_t oy_lo, oy_hi, ox_lo, ox_hi;
        so_bilin_range(y, Hi, Ho, oy_lo, oy_hi);
        so_bilin_range(x, Wi, Wo, ox_lo, ox_hi);
        float acc[3] = {0.0f, 0.0f, 0.0f};
        for (uint32_t oy = oy_lo; oy <= oy_hi; oy++) {
            uint32_t y0, y1;
            float ly;
            so_bilin(oy, Hi, Ho, y0, y1, ly);
            const float wy = (y0 == y ? 1.0f - ly : 0.0f) + (y1 == y ? ly : 0.0f);
            if (wy == 0.0f) continue;
            for (uint32_t ox = ox_lo; ox <= ox_hi; ox++) {
                uint32_t x0, x1;
                float lx;
                so_bilin(ox, Wi, Wo, x0, x1, lx);
                const float wx = (x0 == x ? 1.0f - lx : 0.0f) + (x1 == x ? lx : 0.0f);
                if (wx == 0.0f) continue;
                const so_h8 g = so_ld8(d_out + (((size_t)b * Ho + oy) * Wo + ox) * 8);
                const float w = 2.0f * wy * wx;
#pragma unroll
                for (int c = 0; c < 3; c++) acc[c] += w * (float)g[c];
            }
        }
#pragma unroll
        for (int c = 0; c < 3; c++) d_img[(((size_t)b * 3 + c) * Hi + y) * Wi + x] = acc[c];
    }
}

__global__ void k_timestep_embedding(const float *__restrict__ t, uint32_t B, uint32_t dim, _Float16 *__restrict__ out) {
    const uint32_t half = dim / 2;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < B * half; i += gridDim.x * blockDim.x) {
        const uint32_t b = i / half, k = i - b * half;
        const float w = expf(-9.210340371976184f * (float)k / (float)half);        // ln(10000)
        const float a = t[b] * w;
        out[(size_t)b * dim + k] = (_Float16)cosf(a);
        out[(size_t)b * dim + half + k] = (_Float16)sinf(a);
    }
}

__global__ void k_add_noise(const float *__restrict__ latents, const float *__restrict__ noise, float sa, float sb, uint32_t pixels, _Float16 *__restrict__ out) {
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < pixels; p += gridDim.x * blockDim.x) {
        so_h8 o = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 4; c++) o[c] = (_Float16)(sa * latents[(size_t)c * pixels + p] + sb * noise[(size_t)c * pixels + p]);
        so_st8(out + (size_t)p * 8, o);
        so_st8(out + ((size_t)pixels + p) * 8, o);
    }
}

__global__ void k_sds_grad(const _Float16 *__restrict__ eps, uint32_t ld, const float *__restrict__ noise, float w, float guidance, uint32_t pixels,
                           float *__restrict__ grad) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < pixels * 4; i += gridDim.x * blockDim.x) {
        const uint32_t c = i / pixels, p = i - c * pixels;
        const float eu = (float)eps[(size_t)p * ld + c], et = (float)eps[((size_t)pixels + p) * ld + c];
        float g = w * ((et + guidance * (et - eu)) - noise[i]);
        if (g != g) g = 0.0f;                                                  // torch.nan_to_num defaults
        else if (g > 3.4028234663852886e38f) g = 3.4028234663852886e38f;
        else if (g < -3.4028234663852886e38f) g = -3.4028234663852886e38f;
        grad[i] = g;
    }
}

// zero fill as a kernel instead of a memset call: memset nodes inside captured HIP graphs were observed not to be ordered with
// the neighbouring kernel nodes on this ROCm, which corrupts the GroupNorm statistics on replay
__global__ void __launch_bounds__(256) k_zero_f32(float *__restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0.0f;
}

// ------------------------------------------------------------------------------------------------ host side
static inline uint32_t so_blocks(size_t work_items, uint32_t per_block = 256, uint32_t cap = 4096) {
    size_t b = (work_items + per_block - 1) / per_block;
    if (b > cap) b = cap;
    if (b == 0) b = 1;
    return (uint32_t)b;
}

static inline void so_zero(float *p, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_zero_f32, dim3(so_blocks(n, 256, 2048)), dim3(256), 0, st, p, n);
}

static int gn_check(uint32_t B, uint32_t HW, uint32_t C, uint32_t G) {
    if (B == 0 || HW == 0 || C == 0 || G == 0 || G > GN_MAX_G || (C % G) || (C & 7) || C / G < 4) return CNERF_EINVAL;
    if ((uint64_t)B * HW * (C / 8) >= (1ull << 32)) return CNERF_EINVAL;
    return CNERF_OK;
}
static uint32_t gn_rows_per_block(uint32_t B, uint32_t HW, uint32_t C) {
    // every block ends with 2G same-address float atomics (serialised in L2): ~16 KiB of rows per block, between 16 and 1024
    // blocks per image — enough blocks to stream a 512x512x128 tensor, few enough that the atomics stay off the critical path
    const uint32_t row_bytes = C * 2;
    uint32_t rpb = cn_div_up(16384, row_bytes);
    const uint32_t lo = cn_div_up(HW, 1024), hi = cn_div_up(HW, 16);
    if (rpb < lo) rpb = lo;
    if (rpb > hi) rpb = hi;
    (void)B;
    return rpb;
}

// rows per workgroup of the apply pass: aim at >= 2048 workgroups (small UNet tensors are latency-bound: parallelism first), at most
// 8 rows per thread and column chunk (large VAE tensors: amortise the hoisted parameters)
static uint32_t gn_apply_rows_per_block(uint32_t B, uint32_t HW, uint32_t C) {
    const uint32_t nchunks = C / 8, cpp = nchunks < GN_THREADS ? nchunks : GN_THREADS, rpp = GN_THREADS / cpp;
    uint32_t per_thread = (uint32_t)(((uint64_t)B * HW) / ((uint64_t)rpp * 2048));
    if (per_thread < 1) per_thread = 1;
    if (per_thread > 8) per_thread = 8;
    uint32_t rpb = rpp * per_thread;
    if (rpb > HW) rpb = HW;
    return rpb;
}

template <bool BWD>
static int so_softmax(const void *P, void *S, uint64_t rows, uint32_t cols, uint32_t ld, hipStream_t st) {
    if (cols == 0 || ld < cols || (ld & 7) || ld > 64 * 8 * 8) return CNERF_EINVAL;
    if (rows == 0) return CNERF_OK;
    if (!S || (BWD && !P)) return CNERF_ENULL;
    const dim3 grid((uint32_t)((rows + 3) / 4)), block(256);
    const uint32_t cpl = cn_div_up(ld / 8, 64);
    if (cpl == 1) hipLaunchKernelGGL((k_softmax<1, BWD>), grid, block, 0, st, (const _Float16 *)P, (_Float16 *)S, rows, cols, ld);
    else if (cpl == 2) hipLaunchKernelGGL((k_softmax<2, BWD>), grid, block, 0, st, (const _Float16 *)P, (_Float16 *)S, rows, cols, ld);
    else if (cpl <= 4) hipLaunchKernelGGL((k_softmax<4, BWD>), grid, block, 0, st, (const _Float16 *)P, (_Float16 *)S, rows, cols, ld);
    else hipLaunchKernelGGL((k_softmax<8, BWD>), grid, block, 0, st, (const _Float16 *)P, (_Float16 *)S, rows, cols, ld);
    return cn_launch_status();
}
extern "C" {

int cnerf_sd_groupnorm_forward(const void *x, const float *gamma, const float *beta, uint32_t B, uint32_t HW, uint32_t C, uint32_t G, float eps,
                               int silu, int64_t *sums, int zero_sums, void *y, void *stream) {
    int rc = gn_check(B, HW, C, G);
    if (rc) return rc;
    if (!x || !gamma || !beta || !sums || !y) return CNERF_ENULL;
    hipStream_t st = CN_STREAM(stream);
    if (zero_sums == 1) so_zero(reinterpret_cast<float *>(sums), (size_t)B * G * 4, st);
    const uint32_t rpb = gn_rows_per_block(B, HW, C);
    if (zero_sums != 2)
        hipLaunchKernelGGL((k_gn_stats<0>), dim3(cn_div_up(HW, rpb), B), dim3(GN_THREADS), 0, st, (const _Float16 *)x, (const _Float16 *)nullptr, gamma, beta,
                           (const long long *)nullptr, HW, C, G, eps, silu, rpb, reinterpret_cast<long long *>(sums));
    const uint32_t rpa = gn_apply_rows_per_block(B, HW, C);
    hipLaunchKernelGGL((k_gn_apply<0>), dim3(cn_div_up(HW, rpa), B), dim3(GN_THREADS), 0, st, (const _Float16 *)x, (const _Float16 *)nullptr, gamma, beta,
                       (const long long *)sums, (const long long *)nullptr, HW, C, G, eps, silu, rpa, (_Float16 *)y);
    return cn_launch_status();
}

static int gn_backward(const void *x, const void *dy, const float *gamma, const float *beta, uint32_t B, uint32_t HW, uint32_t C, uint32_t G, float eps,
                       int silu, const int64_t *sums, int64_t *scratch, int scratch_is_zero, const void *residual, void *dx, void *stream) {
    int rc = gn_check(B, HW, C, G);
    if (rc) return rc;
    if (!x || !dy || !gamma || !beta || !sums || !scratch || !dx) return CNERF_ENULL;
    hipStream_t st = CN_STREAM(stream);
    if (!scratch_is_zero) so_zero(reinterpret_cast<float *>(scratch), (size_t)B * G * 4, st);
    const uint32_t rpb = gn_rows_per_block(B, HW, C);
    hipLaunchKernelGGL((k_gn_stats<1>), dim3(cn_div_up(HW, rpb), B), dim3(GN_THREADS), 0, st, (const _Float16 *)x, (const _Float16 *)dy, gamma, beta,
                       (const long long *)sums, HW, C, G, eps, silu, rpb, reinterpret_cast<long long *>(scratch));
    const uint32_t rpa = gn_apply_rows_per_block(B, HW, C);
    hipLaunchKernelGGL((k_gn_apply<1>), dim3(cn_div_up(HW, rpa), B), dim3(GN_THREADS), 0, st, (const _Float16 *)x, (const _Float16 *)dy, gamma, beta,
                       (const long long *)sums, (const long long *)scratch, HW, C, G, eps, silu, rpa, (_Float16 *)dx, (const _Float16 *)residual);
    return cn_launch_status();
}

int cnerf_sd_groupnorm_backward(const void *x, const void *dy, const float *gamma, const float *beta, uint32_t B, uint32_t HW, uint32_t C, uint32_t G,
                                float eps, int silu, const int64_t *sums, int64_t *scratch, void *dx, void *stream) {
    return gn_backward(x, dy, gamma, beta, B, HW, C, G, eps, silu, sums, scratch, 0, nullptr, dx, stream);
}

int cnerf_sd_groupnorm_backward_ex(const void *x, const void *dy, const float *gamma, const float *beta, uint32_t B, uint32_t HW, uint32_t C, uint32_t G,
                                   float eps, int silu, const int64_t *sums, int64_t *scratch, int scratch_is_zero, const void *residual, void *dx,
                                   void *stream) {
    return gn_backward(x, dy, gamma, beta, B, HW, C, G, eps, silu, sums, scratch, scratch_is_zero, residual, dx, stream);
}

int cnerf_sd_layernorm_forward(const void *x, const float *gamma, const float *beta, uint32_t rows, uint32_t C, float eps, void *y, void *stream) {
    if (C == 0 || (C & 7) || C > 64 * 8 * 4) return CNERF_EINVAL;
    if (rows == 0) return CNERF_OK;
    if (!x || !gamma || !beta || !y) return CNERF_ENULL;
    hipStream_t st = CN_STREAM(stream);
    const dim3 grid(cn_div_up(rows, 4)), block(256);
    const uint32_t cpl = cn_div_up(C / 8, 64);
    if (cpl == 1) hipLaunchKernelGGL((k_layernorm<1>), grid, block, 0, st, (const _Float16 *)x, gamma, beta, rows, C, eps, (_Float16 *)y);
    else if (cpl == 2) hipLaunchKernelGGL((k_layernorm<2>), grid, block, 0, st, (const _Float16 *)x, gamma, beta, rows, C, eps, (_Float16 *)y);
    else hipLaunchKernelGGL((k_layernorm<4>), grid, block, 0, st, (const _Float16 *)x, gamma, beta, rows, C, eps, (_Float16 *)y);
    return cn_launch_status();
}

int cnerf_sd_softmax_forward(void *S, uint64_t rows, uint32_t cols, uint32_t ld, void *stream) {
    return so_softmax<false>(nullptr, S, rows, cols, ld, CN_STREAM(stream));
}
int cnerf_sd_softmax_backward(const void *P, void *dP, uint64_t rows, uint32_t cols, uint32_t ld, void *stream) {
    return so_softmax<true>(P, dP, rows, cols, ld, CN_STREAM(stream));
}

int cnerf_sd_geglu(const void *x, uint64_t rows, uint32_t C, void *y, void *stream) {
    if (C == 0 || (C & 7) || rows * (C / 8) >= (1ull << 32)) return CNERF_EINVAL;
    if (rows == 0) return CNERF_OK;
    if (!x || !y) return CNERF_ENULL;
    hipLaunchKernelGGL(k_geglu, dim3(so_blocks((size_t)rows * (C / 8), 256, 8192)), dim3(256), 0, CN_STREAM(stream), (const _Float16 *)x, rows, C, (_Float16 *)y);
    return cn_launch_status();
}

int cnerf_sd_transpose(const void *src, void *dst, uint32_t rows, uint32_t cols, uint32_t lds_, uint32_t ldd, uint32_t batch, uint64_t ss, uint64_t sd,
                       void *stream) {
    if (lds_ < cols || ldd < rows) return CNERF_EINVAL;
    if (rows == 0 || cols == 0 || batch == 0) return CNERF_OK;
    if (!src || !dst) return CNERF_ENULL;
    hipLaunchKernelGGL(k_transpose, dim3(cn_div_up(cols, 64), cn_div_up(ldd, 64), batch), dim3(256), 0, CN_STREAM(stream), (const _Float16 *)src,
                       (_Float16 *)dst, rows, cols, lds_, ldd, ss, sd);
    return cn_launch_status();
}

int cnerf_sd_image_to_vae_input(const float *img, uint32_t B, uint32_t Hi, uint32_t Wi, uint32_t Ho, uint32_t Wo, void *out, void *stream) {
    if (B == 0 || Hi == 0 || Wi == 0 || Ho == 0 || Wo == 0) return CNERF_EINVAL;
    if (!img || !out) return CNERF_ENULL;
    hipLaunchKernelGGL(k_img_fwd, dim3(so_blocks((size_t)B * Ho * Wo)), dim3(256), 0, CN_STREAM(stream), img, B, Hi, Wi, Ho, Wo, (_Float16 *)out);
    return cn_launch_status();
}

int cnerf_sd_image_to_vae_input_backward(const void *d_out, uint32_t B, uint32_t Hi, uint32_t Wi, uint32_t Ho, uint32_t Wo, float *d_img, void *stream) {
    if (B == 0 || Hi == 0 || Wi == 0 || Ho == 0 || Wo == 0) return CNERF_EINVAL;
    if (!d_out || !d_img) return CNERF_ENULL;
    hipStream_t st = CN_STREAM(stream);
    hipLaunchKernelGGL(k_img_bwd, dim3(so_blocks((size_t)B * Hi * Wi, 256, 8192)), dim3(256), 0, st, (const _Float16 *)d_out, B, Hi, Wi, Ho, Wo, d_img);
    return cn_launch_status();
}

int cnerf_sd_clip_preprocess(const float *img, uint32_t B, uint32_t Hi, uint32_t Wi, uint32_t S, const float *mean3, const float *std3, float *out, void *stream) {
    if (B == 0 || Hi == 0 || Wi == 0 || S == 0) return CNERF_EINVAL;
    if (!img || !out || !mean3 || !std3) return CNERF_ENULL;
    // torchvision Resize(int): the smaller edge becomes S, the other int(S * long / short); CenterCrop offsets are round((size - S) / 2)
    uint32_t Hr, Wr;
    if (Hi <= Wi) { Hr = S; Wr = (uint32_t)((uint64_t)S * Wi / Hi); } else { Wr = S; Hr = (uint32_t)((uint64_t)S * Hi / Wi); }
    const uint32_t top = (uint32_t)lrintf((float)(Hr - S) / 2.0f), left = (uint32_t)lrintf((float)(Wr - S) / 2.0f);
    for (int c = 0; c < 3; c++) if (!(std3[c] > 0.0f)) return CNERF_EINVAL;
    hipLaunchKernelGGL(k_clip_preprocess, dim3(so_blocks((size_t)B * 3 * S * S)), dim3(256), 0, CN_STREAM(stream), img, B, Hi, Wi, Hr, Wr, S, top, left,
                       make_float3(mean3[0], mean3[1], mean3[2]), make_float3(1.0f / std3[0], 1.0f / std3[1], 1.0f / std3[2]), out);
    return cn_launch_status();
}

int cnerf_sd_patchify(const float *x, uint32_t B, uint32_t S, uint32_t P, void *out, void *stream) {
    if (B == 0 || S == 0 || P == 0 || S % P) return CNERF_EINVAL;
    if (!x || !out) return CNERF_ENULL;
    hipLaunchKernelGGL(k_patchify, dim3(so_blocks((size_t)B * 3 * S * S)), dim3(256), 0, CN_STREAM(stream), x, B, S, P, (_Float16 *)out);
    return cn_launch_status();
}

int cnerf_sd_timestep_embedding(const float *t, uint32_t B, uint32_t dim, void *out, void *stream) {
    if (B == 0 || dim == 0 || (dim & 1)) return CNERF_EINVAL;
    if (!t || !out) return CNERF_ENULL;
    hipLaunchKernelGGL(k_timestep_embedding, dim3(so_blocks((size_t)B * dim / 2)), dim3(256), 0, CN_STREAM(stream), t, B, dim, (_Float16 *)out);
    return cn_launch_status();
}

int cnerf_sd_add_noise(const float *latents, const float *noise, float alpha_bar, uint32_t pixels, void *unet_in, void *stream) {
    if (pixels == 0 || !(alpha_bar >= 0.0f && alpha_bar <= 1.0f)) return CNERF_EINVAL;
    if (!latents || !noise || !unet_in) return CNERF_ENULL;
    hipLaunchKernelGGL(k_add_noise, dim3(so_blocks(pixels)), dim3(256), 0, CN_STREAM(stream), latents, noise, sqrtf(alpha_bar), sqrtf(1.0f - alpha_bar), pixels,
                       (_Float16 *)unet_in);
    return cn_launch_status();
}

int cnerf_sd_sds_grad(const void *eps, uint32_t ld_eps, const float *noise, float alpha_bar, float guidance, float lambda_sd, uint32_t pixels, float *grad,
                      void *stream) {
    if (pixels == 0 || ld_eps < 4) return CNERF_EINVAL;
    if (!eps || !noise || !grad) return CNERF_ENULL;
    hipLaunchKernelGGL(k_sds_grad, dim3(so_blocks((size_t)pixels * 4)), dim3(256), 0, CN_STREAM(stream), (const _Float16 *)eps, ld_eps, noise,
                       (1.0f - alpha_bar) * lambda_sd, guidance, pixels, grad);
    return cn_launch_status();
}

int cnerf_sd_add(const void *a, const void *b, uint64_t n, void *y, void *stream) {
    if (n == 0) return CNERF_OK;
    if (!a || !b || !y) return CNERF_ENULL;
    hipLaunchKernelGGL((k_ew<0>), dim3(so_blocks(n / 8, 256, 8192)), dim3(256), 0, CN_STREAM(stream), (const _Float16 *)a, (const _Float16 *)b, n / 8, n, (_Float16 *)y);
    return cn_launch_status();
}
int cnerf_sd_silu(const void *x, uint64_t n, void *y, void *stream) {
    if (n == 0) return CNERF_OK;
    if (!x || !y) return CNERF_ENULL;
    hipLaunchKernelGGL((k_ew<1>), dim3(so_blocks(n / 8, 256, 8192)), dim3(256), 0, CN_STREAM(stream), (const _Float16 *)x, (const _Float16 *)nullptr, n / 8, n, (_Float16 *)y);
    return cn_launch_status();
}
int cnerf_sd_concat_gn(const void *a, const void *b, uint64_t rows, uint32_t C1, uint32_t C2, void *y, int64_t *gn_sums, uint32_t gn_groups,
                       uint32_t gn_rows, void *stream) {
    if ((C1 & 7) || (C2 & 7) || C1 + C2 == 0 || rows * ((C1 + C2) / 8) >= (1ull << 32)) return CNERF_EINVAL;
    if (gn_sums && (gn_groups == 0 || (C1 + C2) % gn_groups || (C1 + C2) / gn_groups < 8 || gn_rows == 0 || rows % gn_rows ||
                    (rows / gn_rows) * gn_groups > SO_CONCAT_GN_MAX))
        return CNERF_EINVAL;
    if (rows == 0) return CNERF_OK;
    if (!a || !b || !y) return CNERF_ENULL;
    // (statistics: fewer, fatter workgroups — every workgroup flushes its table with one global atomic per touched entry)
    uint32_t blocks = so_blocks((size_t)rows * ((C1 + C2) / 8), 256, gn_sums ? 2048 : 8192);
    if (gn_sums) {                                                          // (slabs of four groups) x (row chunks), see the kernel
        const uint32_t cg = (C1 + C2) / gn_groups;
        const uint32_t n_slabs = ((gn_groups & 3u) == 0 && (cg & 1u) == 0) ? gn_groups / 4 : 1u;
        uint32_t row_chunks = cn_div_up(blocks, n_slabs);
        if (row_chunks > rows) row_chunks = (uint32_t)rows;
        blocks = n_slabs * (row_chunks ? row_chunks : 1u);
    }
    hipLaunchKernelGGL(k_concat, dim3(blocks), dim3(256), 0, CN_STREAM(stream), (const _Float16 *)a, (const _Float16 *)b, rows, C1, C2, (_Float16 *)y,
                       reinterpret_cast<long long *>(gn_sums), gn_groups, gn_rows);
    return cn_launch_status();
}

int cnerf_sd_concat(const void *a, const void *b, uint64_t rows, uint32_t C1, uint32_t C2, void *y, void *stream) {
    return cnerf_sd_concat_gn(a, b, rows, C1, C2, y, nullptr, 0, 0, stream);
}

}  // extern "C"
