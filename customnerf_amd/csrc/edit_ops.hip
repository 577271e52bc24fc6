// The small tensors between the kernels of one editing step (train_step_editing, utils_init_nerf.py:353-394; encode_imgs' posterior sample,
// sd.py:97-105; the SDS loss, sd.py:150-152).  Each of these was a chain of 3-10 framework element-wise launches on 50-200 KB of data (67 launches,
// 0.29 ms per step in profiles/r05_edit_step_kernels_before.txt); here each is one launch.  All of them are latency-bound: one pass, float32,
// and every reduction is a single workgroup adding in a fixed order (the editing step stays bit-reproducible).
#include "common.h"
#include "../../include/customnerf_sd.h"

#define EO_THREADS 256
static inline uint32_t eo_blocks(size_t n) { return (uint32_t)((n + EO_THREADS - 1) / EO_THREADS); }

// ------------------------------------------------------------------------------------------------ ray buffer <-> images
// out_ray [3][B*HW][6] (cnerf_composite_run: all / fg / bg; channels rgb, depth, weights_sum, mask) -> three NCHW images [B][3][HW]
__global__ void __launch_bounds__(EO_THREADS) k_ray_images(const float *__restrict__ out_ray, uint32_t B, uint32_t HW, float *__restrict__ img_all,
                                                           float *__restrict__ img_fg, float *__restrict__ img_bg) {
    const uint32_t N = B * HW;
    const uint32_t i = blockIdx.x * EO_THREADS + threadIdx.x;
    if (i >= 3 * N) return;
    const uint32_t v = i / N, n = i - v * N, b = n / HW, p = n - b * HW;
    const float *src = out_ray + (size_t)i * 6;
    float *dst = (v == 0 ? img_all : v == 1 ? img_fg : img_bg) + (size_t)b * 3 * HW + p;
    dst[0] = src[0]; dst[HW] = src[1]; dst[2 * (size_t)HW] = src[2];
}

// the adjoint: d(out_ray) [3][B*HW][6], written in full (a missing image gradient and the three other channels: zeros)
__global__ void __launch_bounds__(EO_THREADS) k_ray_images_bwd(const float *__restrict__ d_all, const float *__restrict__ d_fg, const float *__restrict__ d_bg,
                                                               uint32_t B, uint32_t HW, float *__restrict__ d_out_ray) {
    const uint32_t N = B * HW;
    const uint32_t i = blockIdx.x * EO_THREADS + threadIdx.x;
    if (i >= 3 * N) return;
    const uint32_t v = i / N, n = i - v * N, b = n / HW, p = n - b * HW;
    const float *src = v == 0 ? d_all : v == 1 ? d_fg : d_bg;
    float g[3] = {0.0f, 0.0f, 0.0f};
    if (src) {
        src += (size_t)b * 3 * HW + p;
        g[0] = src[0]; g[1] = src[HW]; g[2] = src[2 * (size_t)HW];
    }
    float2 *dst = reinterpret_cast<float2 *>(d_out_ray + (size_t)i * 6);
    dst[0] = make_float2(g[0], g[1]); dst[1] = make_float2(g[2], 0.0f); dst[2] = make_float2(0.0f, 0.0f);
}

// ------------------------------------------------------------------------------------------------ one-workgroup sums
__device__ __forceinline__ float eo_block_sum(float v, float *s_part) {
#pragma unroll
    for (int off = CN_WAVE / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, CN_WAVE);
    const uint32_t wave = threadIdx.x / CN_WAVE, lane = threadIdx.x % CN_WAVE;
    if (lane == 0) s_part[wave] = v;
    __syncthreads();
    float t = 0.0f;
    if (threadIdx.x == 0)
        for (uint32_t w = 0; w < blockDim.x / CN_WAVE; w++) t += s_part[w];
    return t;                                                                       // (valid on thread 0)
}

// loss = scale * mean |a - b| ; dsign[i] = (scale / n) sgn(a[i] - b[i])  (F.l1_loss: d/da = +dsign, d/db = -dsign)
__global__ void __launch_bounds__(1024) k_l1_loss(const float *__restrict__ a, const float *__restrict__ b, uint32_t n, float scale_over_n,
                                                  float *__restrict__ loss, float *__restrict__ dsign) {
    __shared__ float s_part[1024 / CN_WAVE];
    float s = 0.0f;
    for (uint32_t i = threadIdx.x; i < n; i += 1024) {
        const float d = a[i] - b[i];
        s += fabsf(d);
        dsign[i] = d > 0.0f ? scale_over_n : d < 0.0f ? -scale_over_n : 0.0f;
    }
    const float t = eo_block_sum(s, s_part);
    if (threadIdx.x == 0) loss[0] = t * scale_over_n;
}

// the SDS loss (sd.py:150-152): target = (latents - grad).detach(); loss = 0.5 * sum (latents - target)^2 — its value is a log entry, its
// gradient d(loss)/d(latents) = latents - target is `grad` up to the rounding of the subtraction, which is kept: d = lat - (lat - grad).
// What is stored is 2 d — torch's mse backward forms 2 (x - t) first and multiplies by the upstream 0.5 g after, so a |d| above half the
// float32 range (a non-finite epsilon that nan_to_num mapped to +-3.4e38) comes out infinite there, and here (tests/golden/sds.npz "nonfinite").
__global__ void __launch_bounds__(1024) k_sds_loss(const float *__restrict__ lat, const float *__restrict__ grad, uint32_t n, float *__restrict__ loss,
                                                   float *__restrict__ diff) {
    __shared__ float s_part[1024 / CN_WAVE];
    float s = 0.0f;
    for (uint32_t i = threadIdx.x; i < n; i += 1024) {
        const float x = lat[i];
        const float d = x - (x - grad[i]);
        diff[i] = 2.0f * d;
        s += d * d;
    }
    const float t = eo_block_sum(s, s_part);
    if (threadIdx.x == 0) loss[0] = 0.5f * t;
}

// dst = src * (scalar[0] * mult): the backward of the two losses above (scalar = the upstream gradient, a device scalar)
__global__ void __launch_bounds__(EO_THREADS) k_scale_by_scalar(const float *__restrict__ src, const float *__restrict__ scalar, float mult, uint32_t n,
                                                                float *__restrict__ dst) {
    const uint32_t i = blockIdx.x * EO_THREADS + threadIdx.x;
    if (i < n) dst[i] = src[i] * (scalar[0] * mult);
}

// ------------------------------------------------------------------------------------------------ posterior sample of the VAE
// moments [B][hw][8] half NHWC (mean 0..3, logvar 4..7) -> latents [B][4][hw] float32 = (mean + exp(0.5 clamp(logvar, -30, 20)) noise) sf
// (diffusers DiagonalGaussianDistribution.sample, `* 0.18215`: sd.py:102-104)
__global__ void __launch_bounds__(EO_THREADS) k_sample_latents(const _Float16 *__restrict__ moments, const float *__restrict__ noise, uint32_t B, uint32_t hw,
                                                               float sf, float *__restrict__ latents) {
    const uint32_t i = blockIdx.x * EO_THREADS + threadIdx.x;                     // one latent element, NCHW order
    if (i >= B * 4 * hw) return;
    const uint32_t b = i / (4 * hw), r = i - b * 4 * hw, c = r / hw, p = r - c * hw;
    const _Float16 *m = moments + ((size_t)b * hw + p) * 8;
    const float mean = (float)m[c], logvar = fminf(fmaxf((float)m[4 + c], -30.0f), 20.0f);
    latents[i] = (mean + expf(0.5f * logvar) * noise[i]) * sf;
}

__global__ void __launch_bounds__(EO_THREADS) k_sample_latents_bwd(const _Float16 *__restrict__ moments, const float *__restrict__ noise,
                                                                   const float *__restrict__ d_lat, uint32_t B, uint32_t hw, float sf,
                                                                   _Float16 *__restrict__ d_moments) {
    const uint32_t i = blockIdx.x * EO_THREADS + threadIdx.x;                     // one pixel: eight halfs out
    if (i >= B * hw) return;
    const uint32_t b = i / hw, p = i - b * hw;
    const _Float16 *m = moments + (size_t)i * 8;
    _Float16 out[8];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const size_t j = ((size_t)b * 4 + c) * hw + p;
        const float g = d_lat[j] * sf;
        const float lv = (float)m[4 + c];
        const bool inside = lv >= -30.0f && lv <= 20.0f;                           // clamp passes the gradient on the closed interval
        const float std = expf(0.5f * fminf(fmaxf(lv, -30.0f), 20.0f));
        out[c] = (_Float16)g;
        out[4 + c] = (_Float16)(inside ? g * noise[j] * std * 0.5f : 0.0f);
    }
    *reinterpret_cast<uint4 *>(d_moments + (size_t)i * 8) = *reinterpret_cast<const uint4 *>(out);
}

// ------------------------------------------------------------------------------------------------ a few floats from the host
struct EoFloats { float v[16]; };
__global__ void k_set_floats(float *__restrict__ dst, EoFloats vals, uint32_t n) {
    if (threadIdx.x < n) dst[threadIdx.x] = vals.v[threadIdx.x];
}

// ================================================================================================ C ABI
int cnerf_edit_ray_images(const float *out_ray, uint32_t B, uint32_t HW, float *img_all, float *img_fg, float *img_bg, void *stream) {
    if (B == 0 || HW == 0 || (uint64_t)B * HW * 3 >= (1ull << 31)) return CNERF_EINVAL;
    if (!out_ray || !img_all || !img_fg || !img_bg) return CNERF_ENULL;
    hipLaunchKernelGGL(k_ray_images, dim3(eo_blocks((size_t)3 * B * HW)), dim3(EO_THREADS), 0, CN_STREAM(stream), out_ray, B, HW, img_all, img_fg, img_bg);
    return cn_launch_status();
}

int cnerf_edit_ray_images_backward(const float *d_all, const float *d_fg, const float *d_bg, uint32_t B, uint32_t HW, float *d_out_ray, void *stream) {
    if (B == 0 || HW == 0 || (uint64_t)B * HW * 3 >= (1ull << 31)) return CNERF_EINVAL;
    if (!d_out_ray) return CNERF_ENULL;
    hipLaunchKernelGGL(k_ray_images_bwd, dim3(eo_blocks((size_t)3 * B * HW)), dim3(EO_THREADS), 0, CN_STREAM(stream), d_all, d_fg, d_bg, B, HW, d_out_ray);
    return cn_launch_status();
}

int cnerf_edit_l1_loss(const float *a, const float *b, uint32_t n, float scale, float *loss, float *dsign, void *stream) {
    if (n == 0) return CNERF_EINVAL;
    if (!a || !b || !loss || !dsign) return CNERF_ENULL;
    hipLaunchKernelGGL(k_l1_loss, dim3(1), dim3(1024), 0, CN_STREAM(stream), a, b, n, scale / (float)n, loss, dsign);
    return cn_launch_status();
}

int cnerf_edit_sds_loss(const float *latents, const float *grad, uint32_t n, float *loss, float *diff, void *stream) {  // (diff = 2 d)
    if (n == 0) return CNERF_EINVAL;
    if (!latents || !grad || !loss || !diff) return CNERF_ENULL;
    hipLaunchKernelGGL(k_sds_loss, dim3(1), dim3(1024), 0, CN_STREAM(stream), latents, grad, n, loss, diff);
    return cn_launch_status();
}

int cnerf_edit_scale_by_scalar(const float *src, const float *scalar, float mult, uint32_t n, float *dst, void *stream) {
    if (n == 0) return CNERF_OK;
    if (!src || !scalar || !dst) return CNERF_ENULL;
    hipLaunchKernelGGL(k_scale_by_scalar, dim3(eo_blocks(n)), dim3(EO_THREADS), 0, CN_STREAM(stream), src, scalar, mult, n, dst);
    return cn_launch_status();
}

int cnerf_sd_sample_latents(const void *moments, const float *noise, uint32_t B, uint32_t hw, float scaling_factor, float *latents, void *stream) {
    if (B == 0 || hw == 0 || (uint64_t)B * hw * 4 >= (1ull << 31)) return CNERF_EINVAL;
    if (!moments || !noise || !latents) return CNERF_ENULL;
    hipLaunchKernelGGL(k_sample_latents, dim3(eo_blocks((size_t)B * 4 * hw)), dim3(EO_THREADS), 0, CN_STREAM(stream), (const _Float16 *)moments, noise, B, hw,
                       scaling_factor, latents);
    return cn_launch_status();
}

int cnerf_sd_sample_latents_backward(const void *moments, const float *noise, const float *d_latents, uint32_t B, uint32_t hw, float scaling_factor,
                                     void *d_moments, void *stream) {
    if (B == 0 || hw == 0 || (uint64_t)B * hw * 4 >= (1ull << 31)) return CNERF_EINVAL;
    if (!moments || !noise || !d_latents || !d_moments) return CNERF_ENULL;
    hipLaunchKernelGGL(k_sample_latents_bwd, dim3(eo_blocks((size_t)B * hw)), dim3(EO_THREADS), 0, CN_STREAM(stream), (const _Float16 *)moments, noise, d_latents,
                       B, hw, scaling_factor, (_Float16 *)d_moments);
    return cn_launch_status();
}

int cnerf_set_floats(float *dst, const float *host_values, uint32_t n, void *stream) {
    if (n == 0 || n > 16) return CNERF_EINVAL;
    if (!dst || !host_values) return CNERF_ENULL;
    EoFloats f;
    for (uint32_t i = 0; i < 16; i++) f.v[i] = i < n ? host_values[i] : 0.0f;
    hipLaunchKernelGGL(k_set_floats, dim3(1), dim3(64), 0, CN_STREAM(stream), dst, f, n);
    return cn_launch_status();
}
