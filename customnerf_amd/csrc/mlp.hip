// Generic tinycudann.Network replacement (FullyFusedMLP contract) on the gfx950 matrix cores:
//   cnerf_mlp_forward / cnerf_mlp_backward for  n_in (<= 128) -> 64 x {1,2} hidden (ReLU) -> n_out (<= 64, none | sigmoid).
// Same data flow as the fused field (field_common.h): A = weight fragments staged once per workgroup into LDS in fragment
// order, B = activations, the C registers of a layer are the B fragments of the next one.  The backward recomputes the
// forward, runs the dz chain with transposed fragments, writes dL/dx, spills dz_l / layer inputs as [row][sample] matrices
// and finishes with the split-K weight-gradient GEMM of field_dw.h.
// Reference call sites: nerf/network_grid.py:18-54 (RGB_network heads), 98-139 (the three field MLPs, which normally take the
// fused field kernels instead).
#include "field_bwd_common.h"
#include "field_dw.h"

struct MlpDims {
    uint32_t n_in, in_pad16, kp;        // real inputs, tcnn's padded width (parameter row stride), kernel K width (32/64/96/128)
    uint32_t n_out, out_pad16, to;      // outputs (<= 64), rows of the output matrix, 32-row output tiles (1 | 2)
    uint32_t nh, act;                   // hidden layers (1 | 2), output activation (0 none, 1 sigmoid)
};

struct MlpLds {
    uint32_t f0, f1, fo, t0, t1, to, end;      // element offsets: forward L0, hidden, out; transposed L0, hidden, out
};
template <bool H>
__host__ __device__ __forceinline__ MlpLds mlp_lds(const MlpDims &d, bool with_T) {
    MlpLds l;
    uint32_t o = 0;
    l.f0 = o; o += 64 * d.kp;
    l.f1 = o; o += (d.nh == 2) ? 4096 : 0;
    l.fo = o; o += d.to * 32 * 64;
    l.t0 = o; o += with_T ? ((d.kp + 31) / 32) * 32 * 64 : 0;
    l.t1 = o; o += (with_T && d.nh == 2) ? 4096 : 0;
    l.to = o; o += with_T ? d.to * 64 * 32 : 0;
    l.end = o;
    return l;
}

// natural-order input fragments from a row-major [P, ld] matrix (element-wise guarded: n_in need not be a multiple of 8)
template <bool H, int SIN>
__device__ __forceinline__ void mlp_load_x(const void *__restrict__ x, uint32_t ld, uint32_t n_in, uint32_t p, bool valid, uint32_t hi,
                                           typename Prec<H>::frag_t (&b)[SIN]) {
    using PR = Prec<H>;
    const typename PR::elem_t *row = reinterpret_cast<const typename PR::elem_t *>(x) + (size_t)p * ld;
#pragma unroll
    for (int s = 0; s < SIN; s++) {
        if constexpr (H) {
            cn_h8 f;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t col = 16 * s + 8 * hi + j;
                f[j] = (valid && col < n_in) ? row[col] : (_Float16)0;
            }
            b[s] = f;
        } else {
            const uint32_t col = 2 * s + hi;
            b[s] = (valid && col < n_in) ? row[col] : 0.0f;
        }
    }
}

__device__ __forceinline__ float mlp_act(float v, uint32_t act) { return act == 1 ? 1.0f / (1.0f + expf(-v)) : v; }

template <bool H, int SIN, int NH, int TO>
__global__ void __launch_bounds__(FLD_THREADS) k_mlp_fwd(const void *__restrict__ x, uint32_t ldx, const float *__restrict__ params, uint32_t P_,
                                                         MlpDims dm, void *__restrict__ y, uint32_t ldy) {
    using PR = Prec<H>;
    using frag_t = typename PR::frag_t;
    using elem_t = typename PR::elem_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char fld_lds[];
    elem_t *wl = reinterpret_cast<elem_t *>(fld_lds);
    const MlpLds lo = mlp_lds<H>(dm, false);
    constexpr uint32_t S64 = FLD_HID / PR::KS;
    const float *w0 = params, *w1 = params + 64 * dm.in_pad16, *wo = w1 + (NH == 2 ? 4096 : 0);
    fb_stage_layer<H, 0>(wl + lo.f0, w0, 64, dm.in_pad16, 2, SIN, dm.in_pad16);
    if (NH == 2) fb_stage_layer<H, 1>(wl + lo.f1, w1, 64, 64, 2, S64, 64);
    fb_stage_layer<H, 1>(wl + lo.fo, wo, dm.out_pad16, 64, TO, S64, 64);
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hi = lane >> 5;
    const uint32_t n_tiles = (P_ + FLD_TILE - 1) / FLD_TILE;
    for (uint32_t tile = blockIdx.x * FLD_WAVES + wave; tile < n_tiles; tile += gridDim.x * FLD_WAVES) {
        asm volatile("" ::: "memory");
        const uint32_t p = tile * FLD_TILE + (lane & 31);
        const bool valid = p < P_;
        frag_t x0[SIN];
        mlp_load_x<H, SIN>(x, ldx, dm.n_in, p, valid, hi, x0);
        cn_f16v acc[2];
        frag_t h[2 * PR::FR];
        fb_zero(acc);
        fb_gemm<H, 2, SIN>(wl + lo.f0, SIN, 0, x0, lane, acc);
        fb_c_to_b<H, true>(acc, h);
        if (NH == 2) {
            fb_zero(acc);
            fb_gemm<H, 2, S64>(wl + lo.f1, S64, 0, h, lane, acc);
            fb_c_to_b<H, true>(acc, h);
        }
        cn_f16v out[TO];
        fb_zero(out);
        fb_gemm<H, TO, S64>(wl + lo.fo, S64, 0, h, lane, out);
        if (valid) {
            elem_t *yr = reinterpret_cast<elem_t *>(y) + (size_t)p * ldy;
#pragma unroll
            for (int t = 0; t < TO; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const uint32_t row = 32 * t + (uint32_t)fld_rho(r, hi);
                    if (row < dm.n_out) yr[row] = (elem_t)mlp_act(out[t][r], dm.act);
                }
        }
    }
}

// workspace rows: x (kp) | h1 (64) | h2 (64, NH == 2) | z0 (64) | z1 (64, NH == 2) | zo (32 to)
struct MlpWs {
    uint32_t x, h1, h2, z0, z1, zo, rows;
};
__host__ __device__ __forceinline__ MlpWs mlp_ws(const MlpDims &d) {
    MlpWs w;
    uint32_t o = 0;
    w.x = o; o += d.kp;
    w.h1 = o; o += 64;
    w.h2 = o; o += (d.nh == 2) ? 64 : 0;
    w.z0 = o; o += 64;
    w.z1 = o; o += (d.nh == 2) ? 64 : 0;
    w.zo = o; o += 32 * d.to;
    w.rows = o;
    return w;
}

template <bool H, int SIN, int NH, int TIN, int TO>
__global__ void __launch_bounds__(FLD_THREADS) k_mlp_bwd_data(const void *__restrict__ x, uint32_t ldx, const float *__restrict__ params,
                                                              const void *__restrict__ gy, uint32_t ldgy, uint32_t P_, MlpDims dm,
                                                              void *__restrict__ gx, uint32_t ldgx, void *__restrict__ ws_, size_t ld) {
    using PR = Prec<H>;
    using frag_t = typename PR::frag_t;
    using elem_t = typename PR::elem_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char fld_lds[];
    elem_t *wl = reinterpret_cast<elem_t *>(fld_lds);
    const MlpLds lo = mlp_lds<H>(dm, H);
    const MlpWs wo_ = mlp_ws(dm);
    elem_t *ws = reinterpret_cast<elem_t *>(ws_);
    constexpr uint32_t S64 = FLD_HID / PR::KS, S32 = 32 / PR::KS;
    const float *w0 = params, *w1 = params + 64 * dm.in_pad16, *wo = w1 + (NH == 2 ? 4096 : 0);
    fb_stage_layer<H, 0>(wl + lo.f0, w0, 64, dm.in_pad16, 2, SIN, dm.in_pad16);
    if (NH == 2) fb_stage_layer<H, 1>(wl + lo.f1, w1, 64, 64, 2, S64, 64);
    fb_stage_layer<H, 1>(wl + lo.fo, wo, dm.out_pad16, 64, TO, S64, 64);
    if constexpr (H) {
        fb_stage_layer_T<H>(wl + lo.t0, w0, 64, dm.in_pad16, 0, dm.in_pad16, TIN, S64);
        if (NH == 2) fb_stage_layer_T<H>(wl + lo.t1, w1, 64, 64, 0, 64, 2, S64);
        fb_stage_layer_T<H>(wl + lo.to, wo, dm.out_pad16, 64, 0, 64, 2, TO * S32);
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hi = lane >> 5;
    const uint32_t n_tiles = (P_ + FLD_TILE - 1) / FLD_TILE;
    for (uint32_t tile = blockIdx.x * FLD_WAVES + wave; tile < n_tiles; tile += gridDim.x * FLD_WAVES) {
        asm volatile("" ::: "memory");
        const uint32_t p = tile * FLD_TILE + (lane & 31);
        const bool valid = p < P_;
        // ---- forward recompute
        frag_t x0[SIN];
        mlp_load_x<H, SIN>(x, ldx, dm.n_in, p, valid, hi, x0);
        fb_dump_natural<H, SIN>(ws + (size_t)wo_.x * ld, ld, p, hi, x0, dm.kp);
        cn_f16v acc[2];
        frag_t h1[2 * PR::FR], h2[2 * PR::FR];
        fb_zero(acc);
        fb_gemm<H, 2, SIN>(wl + lo.f0, SIN, 0, x0, lane, acc);
        fb_c_to_b<H, true>(acc, h1);
        fb_dump_clayout<H>(ws + (size_t)wo_.h1 * ld, ld, p, hi, h1);
        if (NH == 2) {
            fb_zero(acc);
            fb_gemm<H, 2, S64>(wl + lo.f1, S64, 0, h1, lane, acc);
            fb_c_to_b<H, true>(acc, h2);
            fb_dump_clayout<H>(ws + (size_t)wo_.h2 * ld, ld, p, hi, h2);
        }
        const frag_t *hlast = (NH == 2) ? h2 : h1;
        cn_f16v out[TO];
        fb_zero(out);
        fb_gemm<H, TO, S64>(wl + lo.fo, S64, 0, hlast, lane, out);
        // ---- output gradient: the rows this lane owns, turned in place into the B fragments of the transposed output layer
        {
            const elem_t *gr = reinterpret_cast<const elem_t *>(gy) + (size_t)p * ldgy;
#pragma unroll
            for (int t = 0; t < TO; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const uint32_t row = 32 * t + (uint32_t)fld_rho(r, hi);
                    float g = (valid && row < dm.n_out) ? (float)gr[row] : 0.0f;
                    if (dm.act == 1) {
                        float sg = 1.0f / (1.0f + expf(-out[t][r]));
                        if (H) sg = (float)(_Float16)sg;
                        g *= sg * (1.0f - sg);
                    }
                    if (H) g = (float)(_Float16)g;
                    out[t][r] = g;
                    ws[(size_t)(wo_.zo + row) * ld + p] = (elem_t)g;
                }
        }
        frag_t bo[TO * PR::FR];
#pragma unroll
        for (int t = 0; t < TO; t++)
#pragma unroll
            for (int sub = 0; sub < PR::FR; sub++) {
                if constexpr (H) {
                    cn_h8 f;
#pragma unroll
                    for (int j = 0; j < 8; j++) f[j] = (_Float16)out[t][8 * sub + j];
                    bo[t * PR::FR + sub] = f;
                } else {
                    bo[t * PR::FR + sub] = out[t][sub];
                }
            }
        // ---- chain
        frag_t z1[2 * PR::FR], z0[2 * PR::FR];
        fb_zero(acc);
        fb_gemm_T<H, 2, TO * PR::FR>(wl + lo.to, wo, dm.out_pad16, 64, 0, 64, TO * S32, bo, lane, acc);
        if (NH == 2) {
            fb_c_to_b_masked<H>(acc, h2, z1);
            fb_dump_clayout<H>(ws + (size_t)wo_.z1 * ld, ld, p, hi, z1);
            fb_zero(acc);
            fb_gemm_T<H, 2, S64>(wl + lo.t1, w1, 64, 64, 0, 64, S64, z1, lane, acc);
        }
        fb_c_to_b_masked<H>(acc, h1, z0);
        fb_dump_clayout<H>(ws + (size_t)wo_.z0 * ld, ld, p, hi, z0);
        if (gx) {
            cn_f16v dx[TIN];
            fb_zero(dx);
            fb_gemm_T<H, TIN, S64>(wl + lo.t0, w0, 64, dm.in_pad16, 0, dm.in_pad16, S64, z0, lane, dx);
            if (valid) {
                elem_t *gxr = reinterpret_cast<elem_t *>(gx) + (size_t)p * ldgx;
#pragma unroll
                for (int t = 0; t < TIN; t++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const uint32_t f = 32 * t + (uint32_t)fld_rho(r, hi);
                        if (f < dm.n_in) gxr[f] = (elem_t)dx[t][r];
                    }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side
static int mlp_dims(uint32_t n_in, uint32_t n_out, uint32_t n_neurons, uint32_t n_hidden, int act, MlpDims &d) {
    if (n_in == 0 || n_in > 128 || n_out == 0 || n_out > 64) return CNERF_EINVAL;
    if (n_neurons != 64 || n_hidden < 1 || n_hidden > 2 || act < 0 || act > 1) return CNERF_EINVAL;
    d.n_in = n_in;
    d.in_pad16 = (n_in + 15) / 16 * 16;
    d.kp = (n_in + 31) / 32 * 32;
    d.n_out = n_out;
    d.out_pad16 = (n_out + 15) / 16 * 16;
    d.to = (n_out + 31) / 32;
    d.nh = n_hidden;
    d.act = (uint32_t)act;
    return CNERF_OK;
}

template <bool H>
static int mlp_launch_fwd(const void *x, uint32_t ldx, const float *params, uint32_t P_, const MlpDims &dm, void *y, uint32_t ldy, hipStream_t st) {
    using PR = Prec<H>;
    const MlpLds lo = mlp_lds<H>(dm, false);
    const uint32_t lds_bytes = lo.end * sizeof(typename PR::elem_t);
    uint32_t blocks = cn_div_up(cn_div_up(P_, FLD_TILE), FLD_WAVES);
    if (blocks > 512) blocks = 512;
#define MLP_FWD(KP, NHV, TOV)                                                                                                        \
    {                                                                                                                             \
        auto kern = k_mlp_fwd<H, (KP) / PR::KS, NHV, TOV>;                                                                              \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);   \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(FLD_THREADS), lds_bytes, st, x, ldx, params, P_, dm, y, ldy);                 \
    }
#define MLP_FWD_KP(NHV, TOV)                                                                                  \
    switch (dm.kp) {                                                                                         \
    case 32: MLP_FWD(32, NHV, TOV) break;                                                                  \
    case 64: MLP_FWD(64, NHV, TOV) break;                                                                  \
    case 96: MLP_FWD(96, NHV, TOV) break;                                                                  \
    case 128: MLP_FWD(128, NHV, TOV) break;                                                                  \
    default: return CNERF_EINVAL;                                                                            \
    }
    if (dm.nh == 1 && dm.to == 1) { MLP_FWD_KP(1, 1) }
    else if (dm.nh == 1) { MLP_FWD_KP(1, 2) }
    else if (dm.to == 1) { MLP_FWD_KP(2, 1) }
    else { MLP_FWD_KP(2, 2) }
    return cn_launch_status();
}

template <bool H>
static int mlp_launch_bwd(const void *x, uint32_t ldx, const float *params, const void *gy, uint32_t ldgy, uint32_t P_, const MlpDims &dm, void *gx,
                          uint32_t ldgx, float *gparams, void *workspace, hipStream_t st) {
    using PR = Prec<H>;
    const MlpLds lo = mlp_lds<H>(dm, H);
    const uint32_t lds_bytes = lo.end * sizeof(typename PR::elem_t);
    const size_t ld = fb_ld(P_);
    const uint32_t n_tiles = cn_div_up(P_, FLD_TILE);
    uint32_t blocks = cn_div_up(n_tiles, FLD_WAVES);
    if (blocks > 256) blocks = 256;
#define MLP_BWD(KP, NHV, TOV)                                                                                                                      \
    {                                                                                                                                           \
        auto kern = k_mlp_bwd_data<H, (KP) / PR::KS, NHV, (KP) / 32, TOV>;                                                                           \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);                 \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(FLD_THREADS), lds_bytes, st, x, ldx, params, gy, ldgy, P_, dm, gx, ldgx, workspace, ld);    \
    }
#define MLP_BWD_KP(NHV, TOV)                                                                                  \
    switch (dm.kp) {                                                                                         \
    case 32: MLP_BWD(32, NHV, TOV) break;                                                                  \
    case 64: MLP_BWD(64, NHV, TOV) break;                                                                  \
    case 96: MLP_BWD(96, NHV, TOV) break;                                                                  \
    case 128: MLP_BWD(128, NHV, TOV) break;                                                                  \
    default: return CNERF_EINVAL;                                                                            \
    }
    if (dm.nh == 1 && dm.to == 1) { MLP_BWD_KP(1, 1) }
    else if (dm.nh == 1) { MLP_BWD_KP(1, 2) }
    else if (dm.to == 1) { MLP_BWD_KP(2, 1) }
    else { MLP_BWD_KP(2, 2) }
    int rc = cn_launch_status();
    if (rc) return rc;
    const MlpWs wo = mlp_ws(dm);
    DwPlan pl;
    pl.n_jobs = 0; pl.n_tiles = 0;
    fb_add_job(pl, wo.z0, 64, wo.x, dm.in_pad16, 0, 0, dm.in_pad16, 0);
    uint32_t off = 64 * dm.in_pad16;
    if (dm.nh == 2) { fb_add_job(pl, wo.z1, 64, wo.h1, 64, 0, off, 64, 0); off += 4096; }
    fb_add_job(pl, wo.zo, dm.n_out, dm.nh == 2 ? wo.h2 : wo.h1, 64, 0, off, 64, 0);
    uint32_t splits = n_tiles < 64 ? 1 : (n_tiles < 4096 ? 8 : 64);
    pl.k_tiles_per_split = cn_div_up(cn_div_up(n_tiles, splits), DW_KB / 32) * (DW_KB / 32);
    splits = cn_div_up(n_tiles, pl.k_tiles_per_split);
    hipLaunchKernelGGL((k_field_bwd_dw<H>), dim3(pl.n_tiles, splits), dim3(FLD_THREADS), 0, st, workspace, ld, n_tiles, pl, gparams, gparams, gparams);
    return cn_launch_status();
}

extern "C" {

int cnerf_mlp_forward(const void *x, uint32_t ldx, const float *params, uint32_t P_, uint32_t n_in, uint32_t n_out, uint32_t n_neurons,
                      uint32_t n_hidden_layers, int output_activation, void *y, uint32_t ldy, int dtype, void *stream) {
    MlpDims dm;
    int rc = mlp_dims(n_in, n_out, n_neurons, n_hidden_layers, output_activation, dm);
    if (rc) return rc;
    if (dtype != CNERF_F32 && dtype != CNERF_F16) return CNERF_EINVAL;
    if (ldx < n_in || ldy < n_out) return CNERF_EINVAL;
    if (P_ == 0) return CNERF_OK;
    if (!x || !params || !y) return CNERF_ENULL;
    if (dtype == CNERF_F16) return mlp_launch_fwd<true>(x, ldx, params, P_, dm, y, ldy, CN_STREAM(stream));
    return mlp_launch_fwd<false>(x, ldx, params, P_, dm, y, ldy, CN_STREAM(stream));
}

int cnerf_mlp_backward_workspace_bytes(uint32_t P_, uint32_t n_in, uint32_t n_out, uint32_t n_neurons, uint32_t n_hidden_layers, int dtype,
                                       uint64_t *bytes) {
    if (!bytes) return CNERF_ENULL;
    MlpDims dm;
    int rc = mlp_dims(n_in, n_out, n_neurons, n_hidden_layers, 0, dm);
    if (rc) return rc;
    if (dtype != CNERF_F32 && dtype != CNERF_F16) return CNERF_EINVAL;
    *bytes = (uint64_t)mlp_ws(dm).rows * fb_ld(P_) * (dtype == CNERF_F16 ? 2 : 4) + 256;
    return CNERF_OK;
}

int cnerf_mlp_backward(const void *x, uint32_t ldx, const float *params, const void *grad_y, uint32_t ldgy, uint32_t P_, uint32_t n_in,
                       uint32_t n_out, uint32_t n_neurons, uint32_t n_hidden_layers, int output_activation, void *grad_x, uint32_t ldgx,
                       float *grad_params, void *workspace, uint64_t workspace_bytes, int dtype, void *stream) {
    MlpDims dm;
    int rc = mlp_dims(n_in, n_out, n_neurons, n_hidden_layers, output_activation, dm);
    if (rc) return rc;
    if (dtype != CNERF_F32 && dtype != CNERF_F16) return CNERF_EINVAL;
    if (ldx < n_in || ldgy < n_out || (grad_x && ldgx < n_in)) return CNERF_EINVAL;
    if (P_ == 0) return CNERF_OK;
    if (!x || !params || !grad_y || !grad_params || !workspace) return CNERF_ENULL;
    uint64_t need = 0;
    cnerf_mlp_backward_workspace_bytes(P_, n_in, n_out, n_neurons, n_hidden_layers, dtype, &need);
    if (workspace_bytes < need || (((uintptr_t)workspace) & 15)) return CNERF_EINVAL;
    if (dtype == CNERF_F16)
        return mlp_launch_bwd<true>(x, ldx, params, grad_y, ldgy, P_, dm, grad_x, ldgx, grad_params, workspace, CN_STREAM(stream));
    return mlp_launch_bwd<false>(x, ldx, params, grad_y, ldgy, P_, dm, grad_x, ldgx, grad_params, workspace, CN_STREAM(stream));
}

}  // extern "C"
