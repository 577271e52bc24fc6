// Fused field backward, two-launch form (see field_bwd_common.h for the building blocks; field_bwd_fused.hip holds the
// single-launch fp16 form that keeps the weight-gradient GEMM on chip).
#include "field_bwd_common.h"
#include "field_dw.h"

// ------------------------------------------------------------------------------------------------ workspace
// [row][sample] matrices, row stride ld = samples rounded up to 64.  Row offsets:
struct FieldWs {
    uint32_t enc, h1, h2, fea, hd, hr, dir, z1, z2, z3, zd, zdo, zr, zro, rows;
};
__host__ __device__ __forceinline__ FieldWs fb_ws_layout(const FieldDims &d) {
    FieldWs w;
    uint32_t o = 0;
    w.enc = o; o += d.enc_pad;
    w.h1 = o; o += 64;
    w.h2 = o; o += (d.n_hidden_geo == 2) ? 64 : 0;
    w.fea = o; o += 64;
    w.hd = o; o += 64;
    w.hr = o; o += 64;
    w.dir = o; o += 32;
    w.z1 = o; o += 64;
    w.z2 = o; o += (d.n_hidden_geo == 2) ? 64 : 0;
    w.z3 = o; o += 64;
    w.zd = o; o += 64;
    w.zdo = o; o += 8;
    w.zr = o; o += 64;
    w.zro = o; o += 8;
    w.rows = o;
    return w;
}

// ------------------------------------------------------------------------------------------------ data-gradient kernel
template <bool H, int SENC, int NGEO, int TENC>
__global__ void __launch_bounds__(FLD_THREADS) k_field_bwd_data(const void *__restrict__ enc, const float *__restrict__ xyz, const float *__restrict__ dirs,
                                                                uint32_t dir_group, uint32_t P_, FieldDims dm, const float *__restrict__ pnet,
                                                                const float *__restrict__ pden, const float *__restrict__ prgb,
                                                                const float *__restrict__ g_sigma, const float *__restrict__ g_rgbc,
                                                                void *__restrict__ grad_enc, void *__restrict__ ws_, size_t ld) {
    using PR = Prec<H>;
    using frag_t = typename PR::frag_t;
    using elem_t = typename PR::elem_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char fld_lds[];
    elem_t *wl = reinterpret_cast<elem_t *>(fld_lds);
    const FieldLds lo = fld_lds_layout<H>(dm);
    const FieldLdsT lt = fb_ldsT_layout<H>(dm);
    elem_t *wt = wl + lo.off[7];
    const FieldWs wo = fb_ws_layout(dm);
    elem_t *ws = reinterpret_cast<elem_t *>(ws_);

    constexpr uint32_t S64 = FLD_HID / PR::KS, SDIR = FLD_DIR / PR::KS, S32 = 32 / PR::KS, SR0 = S64 + SDIR;
    const uint32_t in_r0 = FLD_HID + FLD_DIR;
    const float *n0 = pnet, *n1 = pnet + FLD_HID * dm.enc_pad;
    const float *n2 = n1 + (NGEO == 2 ? FLD_HID * FLD_HID : 0);
    const float *d0 = pden, *dO = pden + FLD_HID * FLD_HID;
    const float *r0 = prgb, *rO = prgb + FLD_HID * in_r0;

    // ---- stage weights
    fb_stage_layer<H, 0>(wl + lo.off[0], n0, FLD_HID, dm.enc_pad, 2, SENC, dm.enc_pad);
    if (NGEO == 2) fb_stage_layer<H, 1>(wl + lo.off[1], n1, FLD_HID, FLD_HID, 2, S64, FLD_HID);
    fb_stage_layer<H, 1>(wl + lo.off[2], n2, FLD_HID, FLD_HID, 2, S64, FLD_HID);
    fb_stage_layer<H, 1>(wl + lo.off[3], d0, FLD_HID, FLD_HID, 2, S64, FLD_HID);
    fb_stage_layer<H, 1>(wl + lo.off[4], dO, 16, FLD_HID, 1, S64, FLD_HID);
    fb_stage_layer<H, 2>(wl + lo.off[5], r0, FLD_HID, in_r0, 2, SR0, in_r0);
    fb_stage_layer<H, 1>(wl + lo.off[6], rO, 16, FLD_HID, 1, S64, FLD_HID);
    if constexpr (H) {
        fb_stage_layer_T<H>(wt + lt.off[0], n0, FLD_HID, dm.enc_pad, 0, dm.enc_pad, TENC, S64);
        if (NGEO == 2) fb_stage_layer_T<H>(wt + lt.off[1], n1, FLD_HID, FLD_HID, 0, FLD_HID, 2, S64);
        fb_stage_layer_T<H>(wt + lt.off[2], n2, FLD_HID, FLD_HID, 0, FLD_HID, 2, S64);
        fb_stage_layer_T<H>(wt + lt.off[3], d0, FLD_HID, FLD_HID, 0, FLD_HID, 2, S64);
        fb_stage_layer_T<H>(wt + lt.off[4], dO, 16, FLD_HID, 0, FLD_HID, 2, S32);
        fb_stage_layer_T<H>(wt + lt.off[5], r0, FLD_HID, in_r0, FLD_NDIR, FLD_HID, 2, S64);
        fb_stage_layer_T<H>(wt + lt.off[6], rO, 16, FLD_HID, 0, FLD_HID, 2, S32);
    }
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hi = lane >> 5;
    const uint32_t n_tiles = (P_ + FLD_TILE - 1) / FLD_TILE;
    for (uint32_t tile = blockIdx.x * FLD_WAVES + wave; tile < n_tiles; tile += gridDim.x * FLD_WAVES) {
        // keep the weight fragments in LDS: without this barrier the compiler hoists every fragment load out of the
        // persistent loop (hundreds of VGPRs, one wave per SIMD, spills in the backward)
        asm volatile("" ::: "memory");
        const uint32_t p = tile * FLD_TILE + (lane & 31);
        const bool valid = p < P_;

        // ================= forward recompute
        frag_t x0[SENC];
        fb_load_enc<H, SENC>(enc, P_, dm.L, p, valid, hi, x0);
        cn_f16v acc[2];
        frag_t h1[2 * PR::FR], h2[2 * PR::FR], fea[2 * PR::FR], hd[2 * PR::FR], hr[2 * PR::FR];
        fb_zero(acc);
        fb_gemm<H, 2, SENC>(wl + lo.off[0], SENC, 0, x0, lane, acc);
        fb_dump_natural<H, SENC>(ws + (size_t)wo.enc * ld, ld, p, hi, x0, dm.enc_pad);     // spills happen as soon as a value is final
        fb_c_to_b<H, true>(acc, h1);
        fb_dump_clayout<H>(ws + (size_t)wo.h1 * ld, ld, p, hi, h1);
        if (NGEO == 2) {
            fb_zero(acc);
            fb_gemm<H, 2, S64>(wl + lo.off[1], S64, 0, h1, lane, acc);
            fb_c_to_b<H, true>(acc, h2);
            fb_dump_clayout<H>(ws + (size_t)wo.h2 * ld, ld, p, hi, h2);
        }
        const frag_t *hlast = (NGEO == 2) ? h2 : h1;
        fb_zero(acc);
        fb_gemm<H, 2, S64>(wl + lo.off[2], S64, 0, hlast, lane, acc);
        fb_c_to_b<H, false>(acc, fea);
        fb_dump_clayout<H>(ws + (size_t)wo.fea * ld, ld, p, hi, fea);
        fb_zero(acc);
        fb_gemm<H, 2, S64>(wl + lo.off[3], S64, 0, fea, lane, acc);
        fb_c_to_b<H, true>(acc, hd);
        fb_dump_clayout<H>(ws + (size_t)wo.hd * ld, ld, p, hi, hd);
        cn_f16v out[1];
        fb_zero(out);
        fb_gemm<H, 1, S64>(wl + lo.off[4], S64, 0, hd, lane, out);
        float raw = out[0][0];
        if (H) raw = (float)(_Float16)raw;
        frag_t dfr[SDIR];
        fb_dir_frags<H>(dirs, dir_group, p, valid, hi, dfr);
        fb_zero(acc);
        fb_gemm<H, 2, S64>(wl + lo.off[5], SR0, 0, fea, lane, acc);
        fb_gemm<H, 2, SDIR>(wl + lo.off[5], SR0, S64, dfr, lane, acc);
        fb_dump_natural<H, SDIR>(ws + (size_t)wo.dir * ld, ld, p, hi, dfr, FLD_NDIR);
        fb_c_to_b<H, true>(acc, hr);
        fb_dump_clayout<H>(ws + (size_t)wo.hr * ld, ld, p, hi, hr);
        fb_zero(out);
        fb_gemm<H, 1, S64>(wl + lo.off[6], S64, 0, hr, lane, out);

        // ================= output-layer gradients (rows 0..3 live in registers 0..3 of the hi == 0 lanes)
        float dzro[4] = {0, 0, 0, 0}, dzdo = 0.0f;
        if (valid && hi == 0) {
            const float x = xyz[(size_t)p * 3], y = xyz[(size_t)p * 3 + 1], z = xyz[(size_t)p * 3 + 2];
            const float g = 5.0f * expf(-(x * x + y * y + z * z) / 0.08f);
            dzdo = g_sigma[p] * expf(fminf(fmaxf(raw + g, -15.0f), 15.0f));           // trunc_exp backward
            const float4 gc = *reinterpret_cast<const float4 *>(g_rgbc + (size_t)p * 4);
            const float gcv[4] = {gc.x, gc.y, gc.z, gc.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float sg = 1.0f / (1.0f + expf(-out[0][k]));
                if (H) sg = (float)(_Float16)sg;
                dzro[k] = (k < (int)dm.n_rgb_out) ? gcv[k] * sg * (1.0f - sg) : 0.0f;
            }
        }
        // as B fragments (K = 32 output rows in C order; only rows 0..3 are non-zero)
        frag_t bro[H ? 1 : 4], bdo[H ? 1 : 4];
        if constexpr (H) {
            cn_h8 f = PR::zero(), g = PR::zero();
#pragma unroll
            for (int k = 0; k < 4; k++) f[k] = (_Float16)dzro[k];
            g[0] = (_Float16)dzdo;
            bro[0] = f; bdo[0] = g;
#pragma unroll
            for (int k = 0; k < 4; k++) dzro[k] = (float)f[k];                          // what the weight-gradient GEMM will see
            dzdo = (float)g[0];
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) { bro[k] = dzro[k]; bdo[k] = (k == 0) ? dzdo : 0.0f; }
        }

        // ================= colour head: dz_r = (W_ro^T dz_ro) * [hr > 0] ; dfea  = W_r0[:, fea]^T dz_r
        frag_t zr[2 * PR::FR], zd[2 * PR::FR], z3[2 * PR::FR], z2[2 * PR::FR], z1[2 * PR::FR];
        fb_zero(acc);
        fb_gemm_T<H, 2, (H ? 1 : 4)>(wt + lt.off[6], rO, 16, FLD_HID, 0, FLD_HID, S32, bro, lane, acc);
        fb_c_to_b_masked<H>(acc, hr, zr);
        fb_dump_clayout<H>(ws + (size_t)wo.zr * ld, ld, p, hi, zr);
        cn_f16v dfea[2];
        fb_zero(dfea);
        fb_gemm_T<H, 2, S64>(wt + lt.off[5], r0, FLD_HID, in_r0, FLD_NDIR, FLD_HID, S64, zr, lane, dfea);
        // ================= density head
        fb_zero(acc);
        fb_gemm_T<H, 2, (H ? 1 : 4)>(wt + lt.off[4], dO, 16, FLD_HID, 0, FLD_HID, S32, bdo, lane, acc);
        fb_c_to_b_masked<H>(acc, hd, zd);
        fb_dump_clayout<H>(ws + (size_t)wo.zd * ld, ld, p, hi, zd);
        fb_gemm_T<H, 2, S64>(wt + lt.off[3], d0, FLD_HID, FLD_HID, 0, FLD_HID, S64, zd, lane, dfea);
        fb_c_to_b<H, false>(dfea, z3);                      // network output has no activation
        fb_dump_clayout<H>(ws + (size_t)wo.z3 * ld, ld, p, hi, z3);
        // ================= geometry network
        fb_zero(acc);
        fb_gemm_T<H, 2, S64>(wt + lt.off[2], n2, FLD_HID, FLD_HID, 0, FLD_HID, S64, z3, lane, acc);
        if (NGEO == 2) {
            fb_c_to_b_masked<H>(acc, h2, z2);
            fb_dump_clayout<H>(ws + (size_t)wo.z2 * ld, ld, p, hi, z2);
            fb_zero(acc);
            fb_gemm_T<H, 2, S64>(wt + lt.off[1], n1, FLD_HID, FLD_HID, 0, FLD_HID, S64, z2, lane, acc);
        }
        fb_c_to_b_masked<H>(acc, h1, z1);
        fb_dump_clayout<H>(ws + (size_t)wo.z1 * ld, ld, p, hi, z1);
        cn_f16v denc[TENC];
        fb_zero(denc);
        fb_gemm_T<H, TENC, S64>(wt + lt.off[0], n0, FLD_HID, dm.enc_pad, 0, dm.enc_pad, S64, z1, lane, denc);

        // ================= d(loss)/d(grid features) in the encoder's [L, P, 2] layout
        if (valid) {
#pragma unroll
            for (int t = 0; t < TENC; t++) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const uint32_t f0 = 32 * t + fld_rho(r, hi);          // even feature -> (level, c = 0), f0 + 1 -> c = 1
                    const uint32_t level = f0 >> 1;
                    if (level < dm.L) {
                        if constexpr (H) {
                            union { _Float16 h[2]; uint32_t u; } v;
                            v.h[0] = (_Float16)denc[t][r]; v.h[1] = (_Float16)denc[t][r + 1];
                            reinterpret_cast<uint32_t *>(grad_enc)[(size_t)level * P_ + p] = v.u;
                        } else {
                            reinterpret_cast<float2 *>(grad_enc)[(size_t)level * P_ + p] = make_float2(denc[t][r], denc[t][r + 1]);
                        }
                    }
                }
            }
        }

        // ================= the two single-row operands (padded samples spill zeros)
        if (hi == 0) {
            ws[(size_t)wo.zdo * ld + p] = (elem_t)dzdo;
#pragma unroll
            for (int k = 0; k < 4; k++) ws[(size_t)(wo.zro + k) * ld + p] = (elem_t)dzro[k];
        }
    }
}

static int fb_dims(uint32_t enc_dim, uint32_t n_hidden_geo, uint32_t n_rgb_out, FieldDims &dm) {
    if (enc_dim == 0 || enc_dim > 64 || (enc_dim & 1)) return CNERF_EINVAL;
    if (n_hidden_geo < 1 || n_hidden_geo > 2) return CNERF_EINVAL;
    if (n_rgb_out != 3 && n_rgb_out != 4) return CNERF_EINVAL;
    dm.enc_dim = enc_dim;
    dm.enc_pad = (enc_dim + 15) / 16 * 16;
    dm.n_hidden_geo = n_hidden_geo;
    dm.n_rgb_out = n_rgb_out;
    dm.L = enc_dim / 2;
    return CNERF_OK;
}


template <bool H>
static int fb_launch(const void *enc, const float *xyz, const float *dirs, uint32_t dir_group, uint32_t P_, const FieldDims &dm, const float *pnet,
                     const float *pden, const float *prgb, const float *g_sigma, const float *g_rgbc, void *grad_enc, float *g_net, float *g_den,
                     float *g_rgb, void *workspace, hipStream_t st) {
    using PR = Prec<H>;
    const FieldLds lo = fld_lds_layout<H>(dm);
    const FieldLdsT lt = fb_ldsT_layout<H>(dm);
    const uint32_t lds_bytes = (lo.off[7] + (H ? lt.off[7] : 0)) * sizeof(typename PR::elem_t);
    const size_t ld = fb_ld(P_);
    const uint32_t n_tiles = cn_div_up(P_, FLD_TILE);
    uint32_t blocks = cn_div_up(n_tiles, FLD_WAVES);
    if (blocks > 256) blocks = 256;                       // one persistent workgroup per CU (LDS: 93 KiB fp16 / 96 KiB fp32)
#define FLD_BWD_CASE(SE16, NG, TE)                                                                                                                \
    {                                                                                                                                            \
        auto kern = k_field_bwd_data<H, (SE16) * 16 / PR::KS, NG, TE>;                                                                            \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);                  \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(FLD_THREADS), lds_bytes, st, enc, xyz, dirs, dir_group, P_, dm, pnet, pden, prgb, g_sigma,   \
                           g_rgbc, grad_enc, workspace, ld);                                                                                     \
    }
    const uint32_t se16 = dm.enc_pad / 16;
    if (dm.n_hidden_geo == 1) {
        switch (se16) {
            case 1: FLD_BWD_CASE(1, 1, 1) break;
            case 2: FLD_BWD_CASE(2, 1, 1) break;
            case 3: FLD_BWD_CASE(3, 1, 2) break;
            case 4: FLD_BWD_CASE(4, 1, 2) break;
            default: return CNERF_EINVAL;
        }
    } else {
        switch (se16) {
            case 1: FLD_BWD_CASE(1, 2, 1) break;
            case 2: FLD_BWD_CASE(2, 2, 1) break;
            case 3: FLD_BWD_CASE(3, 2, 2) break;
            case 4: FLD_BWD_CASE(4, 2, 2) break;
            default: return CNERF_EINVAL;
        }
    }
    int rc = cn_launch_status();
    if (rc) return rc;

    // weight gradients
    const FieldWs wo = fb_ws_layout(dm);
    DwPlan pl;
    pl.n_jobs = 0; pl.n_tiles = 0;
    const uint32_t in_r0 = FLD_HID + FLD_DIR;
    fb_add_job(pl, wo.z1, 64, wo.enc, dm.enc_pad, 0, 0, dm.enc_pad, 0);                                        // n0
    uint32_t off = FLD_HID * dm.enc_pad;
    if (dm.n_hidden_geo == 2) { fb_add_job(pl, wo.z2, 64, wo.h1, 64, 0, off, 64, 0); off += 4096; }            // n1
    fb_add_job(pl, wo.z3, 64, dm.n_hidden_geo == 2 ? wo.h2 : wo.h1, 64, 0, off, 64, 0);                        // n2
    fb_add_job(pl, wo.zd, 64, wo.fea, 64, 1, 0, 64, 0);                                                        // d0
    fb_add_job(pl, wo.zdo, 1, wo.hd, 64, 1, 4096, 64, 0);                                                      // do (row 0)
    fb_add_job(pl, wo.zr, 64, wo.dir, FLD_NDIR, 2, 0, in_r0, 0);                                               // r0, direction columns
    fb_add_job(pl, wo.zr, 64, wo.fea, 64, 2, 0, in_r0, FLD_NDIR);                                              // r0, feature columns
    fb_add_job(pl, wo.zro, dm.n_rgb_out, wo.hr, 64, 2, FLD_HID * in_r0, 64, 0);                                // ro
    uint32_t splits = n_tiles < 64 ? 1 : (n_tiles < 4096 ? 8 : 64);
    pl.k_tiles_per_split = cn_div_up(cn_div_up(n_tiles, splits), DW_KB / 32) * (DW_KB / 32);     // whole 128-sample blocks
    splits = cn_div_up(n_tiles, pl.k_tiles_per_split);
    hipLaunchKernelGGL((k_field_bwd_dw<H>), dim3(pl.n_tiles, splits), dim3(FLD_THREADS), 0, st, workspace, ld, n_tiles, pl, g_net, g_den, g_rgb);
    return cn_launch_status();
}

// single-launch fp16 form (field_bwd_fused.hip)
uint64_t ff_workspace_bytes(const FieldDims &dm);
int ff_launch(const void *enc, const float *xyz, const float *dirs, uint32_t dir_group, uint32_t P_, const FieldDims &dm, const float *pnet,
              const float *pden, const float *prgb, const float *g_sigma, const float *g_rgbc, void *grad_enc, float *g_net, float *g_den,
              float *g_rgb, void *workspace, const uint8_t *tile_live, hipStream_t st, const void *wimg);
static bool fb_use_fused(const FieldDims &dm, int dtype) {
    static const int env = cn_tune_env("CNERF_FIELD_FUSED_BWD", 1);
    return env && dtype == CNERF_F16 && dm.enc_pad <= 32;           // 92 KiB weight fragments + 64 KiB staging must fit the 160 KiB LDS
}

extern "C" {

int cnerf_field_backward_workspace_bytes(uint32_t P_, uint32_t enc_dim, uint32_t n_hidden_geo, uint32_t n_rgb_out, int dtype, uint64_t *bytes) {
    if (!bytes) return CNERF_ENULL;
    FieldDims dm;
    int rc = fb_dims(enc_dim, n_hidden_geo, n_rgb_out, dm);
    if (rc) return rc;
    if (dtype != CNERF_F32 && dtype != CNERF_F16) return CNERF_EINVAL;
    const FieldWs wo = fb_ws_layout(dm);
    if (fb_use_fused(dm, dtype)) *bytes = ff_workspace_bytes(dm);
    else *bytes = (uint64_t)wo.rows * fb_ld(P_) * (dtype == CNERF_F16 ? 2 : 4) + 256;
    return CNERF_OK;
}

int cnerf_field_backward(const void *enc, const float *xyz, const float *dirs, uint32_t dir_group, uint32_t P_, uint32_t enc_dim,
                         uint32_t n_hidden_geo, uint32_t n_rgb_out, const float *params_net, const float *params_den, const float *params_rgb,
                         const float *grad_sigma, const float *grad_rgbc, void *grad_enc, float *grad_params_net, float *grad_params_den,
                         float *grad_params_rgb, void *workspace, uint64_t workspace_bytes, int dtype, void *stream) {
    return cnerf_field_backward_ex(enc, xyz, dirs, dir_group, P_, enc_dim, n_hidden_geo, n_rgb_out, params_net, params_den, params_rgb, grad_sigma, grad_rgbc,
                                   grad_enc, grad_params_net, grad_params_den, grad_params_rgb, workspace, workspace_bytes, dtype, nullptr, stream);
}

int cnerf_field_backward_ex(const void *enc, const float *xyz, const float *dirs, uint32_t dir_group, uint32_t P_, uint32_t enc_dim,
                            uint32_t n_hidden_geo, uint32_t n_rgb_out, const float *params_net, const float *params_den, const float *params_rgb,
                            const float *grad_sigma, const float *grad_rgbc, void *grad_enc, float *grad_params_net, float *grad_params_den,
                            float *grad_params_rgb, void *workspace, uint64_t workspace_bytes, int dtype, const uint8_t *tile_live, void *stream) {
    return cnerf_field_backward_img(enc, xyz, dirs, dir_group, P_, enc_dim, n_hidden_geo, n_rgb_out, params_net, params_den, params_rgb, grad_sigma, grad_rgbc,
                                    grad_enc, grad_params_net, grad_params_den, grad_params_rgb, workspace, workspace_bytes, dtype, tile_live, nullptr, stream);
}

int cnerf_field_backward_img(const void *enc, const float *xyz, const float *dirs, uint32_t dir_group, uint32_t P_, uint32_t enc_dim,
                             uint32_t n_hidden_geo, uint32_t n_rgb_out, const float *params_net, const float *params_den, const float *params_rgb,
                             const float *grad_sigma, const float *grad_rgbc, void *grad_enc, float *grad_params_net, float *grad_params_den,
                             float *grad_params_rgb, void *workspace, uint64_t workspace_bytes, int dtype, const uint8_t *tile_live,
                             const void *weight_image, void *stream) {
    FieldDims dm;
    int rc = fb_dims(enc_dim, n_hidden_geo, n_rgb_out, dm);
    if (rc) return rc;
    if (dtype != CNERF_F32 && dtype != CNERF_F16) return CNERF_EINVAL;
    if (P_ == 0) return CNERF_OK;
    if (!enc || !xyz || !dirs || !params_net || !params_den || !params_rgb || !grad_sigma || !grad_rgbc || !grad_enc || !grad_params_net ||
        !grad_params_den || !grad_params_rgb || !workspace)
        return CNERF_ENULL;
    if (dir_group == 0 || (((uintptr_t)grad_rgbc) & 15) || (((uintptr_t)workspace) & 15) || (((uintptr_t)weight_image) & 15)) return CNERF_EINVAL;
    uint64_t need = 0;
    cnerf_field_backward_workspace_bytes(P_, enc_dim, n_hidden_geo, n_rgb_out, dtype, &need);
    if (workspace_bytes < need) return CNERF_EINVAL;
    if (fb_use_fused(dm, dtype))
        return ff_launch(enc, xyz, dirs, dir_group, P_, dm, params_net, params_den, params_rgb, grad_sigma, grad_rgbc, grad_enc, grad_params_net,
                         grad_params_den, grad_params_rgb, workspace, tile_live, CN_STREAM(stream), weight_image);
    if (dtype == CNERF_F16)
        return fb_launch<true>(enc, xyz, dirs, dir_group, P_, dm, params_net, params_den, params_rgb, grad_sigma, grad_rgbc, grad_enc,
                               grad_params_net, grad_params_den, grad_params_rgb, workspace, CN_STREAM(stream));
    return fb_launch<false>(enc, xyz, dirs, dir_group, P_, dm, params_net, params_den, params_rgb, grad_sigma, grad_rgbc, grad_enc, grad_params_net,
                            grad_params_den, grad_params_rgb, workspace, CN_STREAM(stream));
}

}  // extern "C"
