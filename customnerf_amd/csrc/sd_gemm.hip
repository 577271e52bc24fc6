// Implicit-GEMM / GEMM kernel of the SDS path (UNet forward, VAE encoder forward + input-gradient) on the gfx950 matrix cores.
//   C[m][n] = epilogue(alpha * sum_k A(m,k) B[n][k]),  fp16 operands, fp32 accumulation (v_mfma_f32_32x32x16_f16).
// One workgroup (4 waves, 2 x 2) owns a 128 x 128 output tile; each wave 64 x 64 = 2 x 2 MFMA tiles (64 accumulator VGPRs).
// K advances in steps of 64 through a double-buffered LDS stage; rows are padded to 72 halfs (144 B) which makes the
// 16-byte fragment reads (ds_read_b128, 16-lane groups) conflict-free.  Both operands are K-contiguous in memory — NHWC
// activations make the im2col view K-contiguous per tap, weights are packed [Cout][kh][kw][Cin] — so every global access is a
// 16-byte load and nothing is transposed on the way to the MFMA fragments.
// The im2col address arithmetic (tap -> input pixel, zero padding, nearest 2x upsampling, transposed stride for input
// gradients) runs once per 16-byte chunk in the loader; there is no materialised im2col buffer.
// Small-M problems (the 8x8 / 16x16 UNet levels at batch 2) are split along K over blockIdx.z with fp32 partials in a
// workspace and a separate epilogue pass, so the launch still covers the 256 CUs.
#include "common.h"
#include "../../include/customnerf_sd.h"

typedef _Float16 sd_h8 __attribute__((ext_vector_type(8)));
typedef float sd_f16v __attribute__((ext_vector_type(16)));

#define SG_BM 128
#define SG_BN 128
#define SG_BK 64
#define SG_LDK (SG_BK + 8)          // padded LDS row, halfs
#define SG_THREADS 256
#define SG_STAGE_HALFS ((SG_BM + SG_BN) * SG_LDK)

__host__ __device__ __forceinline__ int sg_rho(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

__device__ __forceinline__ float sg_act(float v, int act) {
    if (act == 1) return v / (1.0f + __expf(-v));
    if (act == 2) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
    return v;
}

// per-thread description of the A row it stages (fixed for the whole K loop)
struct SgRowA {
    const _Float16 *base;     // dense: row pointer; conv: image base
    int32_t oh_s, ow_s;       // conv: oh*stride - pad_t, ow*stride - pad_l
    bool valid;
};

__device__ __forceinline__ uint4 sg_load_a(const CnerfSdGemm &g, const SgRowA &row, uint32_t k) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (!row.valid || k >= g.K) return v;
    if (g.mode == 0) return *reinterpret_cast<const uint4 *>(row.base + k);
    const uint32_t tap = k / g.Cin, c = k - tap * g.Cin;
    const uint32_t kh = tap / g.KW, kw = tap - kh * g.KW;
    int32_t nh = row.oh_s + (int32_t)kh, nw = row.ow_s + (int32_t)kw;
    if (nh < 0 || nw < 0) return v;
    if (g.tstride == 2) {
        if ((nh | nw) & 1) return v;
        nh >>= 1; nw >>= 1;
    }
    if (g.ups == 2) { nh >>= 1; nw >>= 1; }
    if ((uint32_t)nh >= g.H_in || (uint32_t)nw >= g.W_in) return v;
    return *reinterpret_cast<const uint4 *>(row.base + ((size_t)nh * g.W_in + nw) * g.Cin + c);
}

template <bool SPLIT>
__global__ void __launch_bounds__(SG_THREADS) k_sd_gemm(const CnerfSdGemm g, float *__restrict__ partial, uint32_t k_tiles_per_split) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sg_lds[];
    _Float16 *lds = reinterpret_cast<_Float16 *>(sg_lds);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, li = lane & 31;
    const uint32_t wm = wave >> 1, wn = wave & 1;
    // blockIdx.x walks M tiles fastest: neighbouring workgroups share the B (weight) tile through L2
    const uint32_t m0 = blockIdx.x * SG_BM, n0 = blockIdx.y * SG_BN;
    uint32_t z = 0, split = 0;
    if (SPLIT) split = blockIdx.z; else z = blockIdx.z;
    const uint32_t zo = z / g.batch_inner, zi = z - zo * g.batch_inner;
    const _Float16 *A = reinterpret_cast<const _Float16 *>(g.A) + g.sa_o * zo + g.sa_i * zi;
    const _Float16 *B = reinterpret_cast<const _Float16 *>(g.B) + g.sb_o * zo + g.sb_i * zi;

    // staging role: 2 threads per row, 4 consecutive 16-byte chunks each
    const uint32_t srow = tid >> 1, sk = (tid & 1) * 32;
    SgRowA ra;
    {
        const uint32_t m = m0 + srow;
        ra.valid = m < g.M;
        ra.oh_s = ra.ow_s = 0;
        if (g.mode == 0) ra.base = A + (size_t)m * g.lda;
        else {
            const uint32_t hw = g.H_out * g.W_out;
            const uint32_t img = m / hw, rem = m - img * hw, oh = rem / g.W_out, ow = rem - oh * g.W_out;
            ra.base = A + (size_t)img * g.H_in * g.W_in * g.Cin;
            ra.oh_s = (int32_t)(oh * g.stride) - (int32_t)g.pad_t;
            ra.ow_s = (int32_t)(ow * g.stride) - (int32_t)g.pad_l;
        }
    }
    const uint32_t nrow = n0 + srow;
    const bool b_valid = nrow < g.N;
    const _Float16 *brow = B + (size_t)nrow * g.ldb;

    const uint32_t n_ktiles = (g.K + SG_BK - 1) / SG_BK;
    uint32_t kt0 = 0, kt1 = n_ktiles;
    if (SPLIT) { kt0 = split * k_tiles_per_split; kt1 = min(kt0 + k_tiles_per_split, n_ktiles); }

    sd_f16v acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    uint4 sa[4], sb[4];
    auto fetch = [&](uint32_t kt) {
        const uint32_t k = kt * SG_BK + sk;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            sa[c] = sg_load_a(g, ra, k + 8 * c);
            sb[c] = (b_valid && k + 8 * c < g.K) ? *reinterpret_cast<const uint4 *>(brow + k + 8 * c) : make_uint4(0, 0, 0, 0);
        }
    };
    auto commit = [&](uint32_t stage) {
        _Float16 *sA = lds + stage * SG_STAGE_HALFS, *sB = sA + SG_BM * SG_LDK;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            *reinterpret_cast<uint4 *>(sA + srow * SG_LDK + sk + 8 * c) = sa[c];
            *reinterpret_cast<uint4 *>(sB + srow * SG_LDK + sk + 8 * c) = sb[c];
        }
    };
    if (kt0 < kt1) {
        fetch(kt0);
        commit(0);
    }
    __syncthreads();
    for (uint32_t kt = kt0; kt < kt1; kt++) {
        const uint32_t stage = (kt - kt0) & 1;
        if (kt + 1 < kt1) fetch(kt + 1);
        const _Float16 *sA = lds + stage * SG_STAGE_HALFS, *sB = sA + SG_BM * SG_LDK;
#pragma unroll
        for (int s = 0; s < SG_BK / 16; s++) {
            sd_h8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; i++) a[i] = *reinterpret_cast<const sd_h8 *>(sA + (wm * 64 + i * 32 + li) * SG_LDK + s * 16 + 8 * hi);
#pragma unroll
            for (int j = 0; j < 2; j++) b[j] = *reinterpret_cast<const sd_h8 *>(sB + (wn * 64 + j * 32 + li) * SG_LDK + s * 16 + 8 * hi);
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < kt1) commit(stage ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane owns column n = li (+32 j), rows rho(r, hi) (+32 i)
    if (SPLIT) {
        float *P = partial + (size_t)split * g.M * g.N;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const uint32_t n = n0 + wn * 64 + j * 32 + li;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const uint32_t m = m0 + wm * 64 + i * 32 + sg_rho(r, hi);
                    if (m < g.M && n < g.N) P[(size_t)m * g.N + n] = acc[i][j][r];
                }
            }
        return;
    }
    _Float16 *C = g.C ? reinterpret_cast<_Float16 *>(g.C) + g.sc_o * zo + g.sc_i * zi : nullptr;
    float *C32 = g.C32 ? g.C32 + g.sc_o * zo + g.sc_i * zi : nullptr;
    const _Float16 *R = g.residual ? reinterpret_cast<const _Float16 *>(g.residual) + g.sc_o * zo + g.sc_i * zi : nullptr;
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const uint32_t n = n0 + wn * 64 + j * 32 + li;
        if (n >= g.N) continue;
        const float bias = g.bias ? g.bias[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const uint32_t m = m0 + wm * 64 + i * 32 + sg_rho(r, hi);
                if (m >= g.M) continue;
                float v = acc[i][j][r] * g.alpha + bias;
                if (g.bias_rows) v += g.bias_rows[(size_t)(m / g.rows_per_bias_row) * g.N + n];
                v = sg_act(v, g.act);
                if (R) v += (float)R[(size_t)m * g.ldr + n];
                if (C) C[(size_t)m * g.ldc + n] = (_Float16)v;
                if (C32) C32[(size_t)m * g.ldc + n] = v;
            }
    }
}

// split-K tail: sum the partials, apply the epilogue
__global__ void __launch_bounds__(256) k_sd_gemm_splitk_epilogue(const CnerfSdGemm g, const float *__restrict__ partial, uint32_t splits) {
    const size_t total = (size_t)g.M * g.N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t m = (uint32_t)(i / g.N), n = (uint32_t)(i - (size_t)m * g.N);
        float a = 0.0f;
        for (uint32_t s = 0; s < splits; s++) a += partial[(size_t)s * total + i];
        float v = a * g.alpha + (g.bias ? g.bias[n] : 0.0f);
        if (g.bias_rows) v += g.bias_rows[(size_t)(m / g.rows_per_bias_row) * g.N + n];
        v = sg_act(v, g.act);
        if (g.residual) v += (float)reinterpret_cast<const _Float16 *>(g.residual)[(size_t)m * g.ldr + n];
        if (g.C) reinterpret_cast<_Float16 *>(g.C)[(size_t)m * g.ldc + n] = (_Float16)v;
        if (g.C32) g.C32[(size_t)m * g.ldc + n] = v;
    }
}

// ------------------------------------------------------------------------------------------------ host side
static int sg_check(const CnerfSdGemm *g) {
    if (!g) return CNERF_ENULL;
    if (!g->A || !g->B || (!g->C && !g->C32)) return CNERF_ENULL;
    if (g->M == 0 || g->N == 0 || g->K == 0 || (g->K & 7) || (g->ldb & 7)) return CNERF_EINVAL;
    if (g->batch_outer == 0 || g->batch_inner == 0) return CNERF_EINVAL;
    if (g->act < 0 || g->act > 2) return CNERF_EINVAL;
    if (g->bias_rows && g->rows_per_bias_row == 0) return CNERF_EINVAL;
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

static uint32_t sg_splits(const CnerfSdGemm *g, uint32_t &k_tiles_per_split) {
    const uint32_t tiles = cn_div_up(g->M, SG_BM) * cn_div_up(g->N, SG_BN);
    const uint32_t n_ktiles = cn_div_up(g->K, SG_BK);
    k_tiles_per_split = n_ktiles;
    if (g->batch_outer * g->batch_inner != 1 || tiles >= 256 || n_ktiles < 8) return 1;
    uint32_t want = cn_div_up(512, tiles);
    if (want > n_ktiles / 4) want = n_ktiles / 4;          // at least 4 K tiles (256 k) per split
    if (want > 32) want = 32;
    if (want <= 1) return 1;
    k_tiles_per_split = cn_div_up(n_ktiles, want);
    return cn_div_up(n_ktiles, k_tiles_per_split);
}

extern "C" {

int cnerf_sd_gemm_workspace_bytes(const CnerfSdGemm *g, uint64_t *bytes) {
    if (!bytes) return CNERF_ENULL;
    int rc = sg_check(g);
    if (rc) return rc;
    uint32_t kps;
    const uint32_t splits = sg_splits(g, kps);
    *bytes = splits > 1 ? (uint64_t)splits * g->M * g->N * sizeof(float) : 0;
    return CNERF_OK;
}

int cnerf_sd_gemm(const CnerfSdGemm *g, void *workspace, uint64_t workspace_bytes, void *stream) {
    int rc = sg_check(g);
    if (rc) return rc;
    hipStream_t st = CN_STREAM(stream);
    const uint32_t lds_bytes = 2 * SG_STAGE_HALFS * sizeof(_Float16);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_sd_gemm<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_sd_gemm<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        attr_set = true;
    }
    uint32_t kps;
    uint32_t splits = sg_splits(g, kps);
    if (splits > 1 && (!workspace || workspace_bytes < (uint64_t)splits * g->M * g->N * sizeof(float))) { splits = 1; }
    const dim3 block(SG_THREADS);
    if (splits > 1) {
        const dim3 grid(cn_div_up(g->M, SG_BM), cn_div_up(g->N, SG_BN), splits);
        hipLaunchKernelGGL((k_sd_gemm<true>), grid, block, lds_bytes, st, *g, reinterpret_cast<float *>(workspace), kps);
        const size_t total = (size_t)g->M * g->N;
        const uint32_t eb = (uint32_t)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
        hipLaunchKernelGGL(k_sd_gemm_splitk_epilogue, dim3(eb), dim3(256), 0, st, *g, reinterpret_cast<const float *>(workspace), splits);
    } else {
        const dim3 grid(cn_div_up(g->M, SG_BM), cn_div_up(g->N, SG_BN), g->batch_outer * g->batch_inner);
        hipLaunchKernelGGL((k_sd_gemm<false>), grid, block, lds_bytes, st, *g, (float *)nullptr, 0u);
    }
    return cn_launch_status();
}

}  // extern "C"
