// Ray generation, fused Adam step and library identification for gfx950.
#include "common.h"

// ------------------------------------------------------------------------------------------------ ray generation
// One thread per pixel of one view.  convention 0: nerfstudio / OpenGL pinhole (reference nerf/provider.py:402-464,
// output already in [H,W] order, i.e. the [W,H]->[H,W] permute of :460-464 is folded into the index);
// convention 1: torch-ngp get_rays (reference nerf/provider_utils.py:239-302);
// convention 2: nerfstudio OPENCV_FISHEYE (provider.py:421-433): the normalised pixel coordinate is un-distorted by ten Newton steps on the
// OpenCV radial + tangential model (provider_utils.py:128-234, operation for operation), then theta = |coord| (clipped to [0, pi]) gives
// dir = (x sin(theta)/theta, y sin(theta)/theta, -cos(theta)) before the rotation.
struct RayDistortion { float k1, k2, k3, k4, p1, p2; };

// provider_utils.py:128-194 (_compute_residual_and_jacobian) + :221-232 (one Newton step), float32, the reference's operation order
__device__ __forceinline__ void rg_undistort_step(float &x, float &y, float xd, float yd, const RayDistortion &k, float eps) {
    const float r = x * x + y * y;
    const float d = 1.0f + r * (k.k1 + r * (k.k2 + r * (k.k3 + r * k.k4)));
    const float fx = d * x + 2 * k.p1 * x * y + k.p2 * (r + 2 * x * x) - xd;
    const float fy = d * y + 2 * k.p2 * x * y + k.p1 * (r + 2 * y * y) - yd;
    const float d_r = k.k1 + r * (2.0f * k.k2 + r * (3.0f * k.k3 + r * 4.0f * k.k4));
    const float d_x = 2.0f * x * d_r, d_y = 2.0f * y * d_r;
    const float fx_x = d + d_x * x + 2.0f * k.p1 * y + 6.0f * k.p2 * x;
    const float fx_y = d_y * x + 2.0f * k.p1 * x + 2.0f * k.p2 * y;
    const float fy_x = d_x * y + 2.0f * k.p2 * y + 2.0f * k.p1 * x;
    const float fy_y = d + d_y * y + 2.0f * k.p2 * x + 6.0f * k.p1 * y;
    const float den = fy_x * fx_y - fx_x * fy_y;
    const float xn = fx * fy_y - fy * fx_y, yn = fy * fx_x - fx * fy_x;
    const bool ok = fabsf(den) > eps;
    x = x + (ok ? xn / den : 0.0f);
    y = y + (ok ? yn / den : 0.0f);
}

__global__ void __launch_bounds__(256) k_generate_rays(const float *__restrict__ c2w, uint32_t V, uint32_t H, uint32_t W, float fx, float fy,
                                                       float cx, float cy, float level, int convention, RayDistortion dist,
                                                       float *__restrict__ origins, float *__restrict__ directions) {
    const uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t v = blockIdx.y;
    if (pix >= H * W) return;
    const uint32_t iy = pix / W, ix = pix - iy * W;
    const float *m = c2w + (size_t)v * 12;   // row-major [3,4]
    float dx, dy, dz;
    if (convention == 0 || convention == 2) {
        // torch.linspace(0, W*l-1, W)[ix]: start + ix*step for the lower half, end - (W-1-ix)*step for the upper half
        const float endx = W * level - 1.0f, endy = H * level - 1.0f;
        const float stepx = W > 1 ? endx / (float)(W - 1) : 0.0f, stepy = H > 1 ? endy / (float)(H - 1) : 0.0f;
        const float x = ((ix < W / 2) ? (float)ix * stepx : endx - (float)(W - 1 - ix) * stepx) + 0.5f;
        const float y = ((iy < H / 2) ? (float)iy * stepy : endy - (float)(H - 1 - iy) * stepy) + 0.5f;
        float px = (x - cx) / fx, py = -(y - cy) / fy, pz = -1.0f;
        if (convention == 2) {
            const float xd = px, yd = py;
            for (int it = 0; it < 10; it++) rg_undistort_step(px, py, xd, yd, dist, 1e-3f);      // max_iterations = 10, eps = 1e-3 (provider_utils.py:200-201)
            float theta = sqrtf(px * px + py * py);
            theta = fminf(fmaxf(theta, 0.0f), 3.14159265358979323846f);                      // torch.clip(theta, 0, math.pi)
            const float st = sinf(theta);
            px = px * st / theta;                                                               // (0/0 at the exact principal point, as in the reference)
            py = py * st / theta;
            pz = -cosf(theta);
        }
        dx = m[0] * px + m[1] * py + m[2] * pz;
        dy = m[4] * px + m[5] * py + m[6] * pz;
        dz = m[8] * px + m[9] * py + m[10] * pz;
        const float nrm = fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-12f);   // F.normalize eps
        dx /= nrm; dy /= nrm; dz /= nrm;
    } else {
        const float i = (float)ix + 0.5f, j = (float)iy + 0.5f;
        float px = (i - cx) / fx, py = (j - cy) / fy, pz = 1.0f;
        const float nrm = sqrtf(fmaxf(px * px + py * py + pz * pz, 1e-20f));      // safe_normalize
        px /= nrm; py /= nrm; pz /= nrm;
        dx = px * m[0] + py * m[1] + pz * m[2];
        dy = px * m[4] + py * m[5] + pz * m[6];
        dz = px * m[8] + py * m[9] + pz * m[10];
    }
    const size_t o = ((size_t)v * H * W + pix) * 3;
    origins[o] = m[3]; origins[o + 1] = m[7]; origins[o + 2] = m[11];
    directions[o] = dx; directions[o + 1] = dy; directions[o + 2] = dz;
}

// ------------------------------------------------------------------------------------------------ Adam
// torch.optim.Adam semantics (no weight decay, no amsgrad):  m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
// p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps).  One pass: 16 B/lane vector loads, optional fp16 shadow store,
// optional gradient zeroing (saves the separate 4 B/param memset pass of zero_grad).
__global__ void __launch_bounds__(256) k_adam(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
                                              __half *__restrict__ ph, uint64_t n, float step_size, float beta1, float beta2, float eps,
                                              float rsqrt_bc2, float gscale, int zero_grad) {
    const uint64_t n4 = n / 4;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pp = reinterpret_cast<float4 *>(p)[i], gg = reinterpret_cast<float4 *>(g)[i];
        float4 mm = reinterpret_cast<float4 *>(m)[i], vv = reinterpret_cast<float4 *>(v)[i];
        float *pa = &pp.x, *ga = &gg.x, *ma = &mm.x, *va = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float gk = ga[k] * gscale;
            ma[k] = beta1 * ma[k] + (1.0f - beta1) * gk;
            va[k] = beta2 * va[k] + (1.0f - beta2) * gk * gk;
            pa[k] -= step_size * ma[k] / (sqrtf(va[k]) * rsqrt_bc2 + eps);
        }
        reinterpret_cast<float4 *>(p)[i] = pp;
        reinterpret_cast<float4 *>(m)[i] = mm;
        reinterpret_cast<float4 *>(v)[i] = vv;
        if (zero_grad) reinterpret_cast<float4 *>(g)[i] = make_float4(0, 0, 0, 0);
        if (ph) {
            union { __half2 h[2]; uint2 u; } o;
            o.h[0] = __floats2half2_rn(pp.x, pp.y);
            o.h[1] = __floats2half2_rn(pp.z, pp.w);
            reinterpret_cast<uint2 *>(ph)[i] = o.u;
        }
    }
    for (uint64_t i = n4 * 4 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float gk = g[i] * gscale;
        const float mk = beta1 * m[i] + (1.0f - beta1) * gk;
        const float vk = beta2 * v[i] + (1.0f - beta2) * gk * gk;
        const float pk = p[i] - step_size * mk / (sqrtf(vk) * rsqrt_bc2 + eps);
        m[i] = mk; v[i] = vk; p[i] = pk;
        if (zero_grad) g[i] = 0;
        if (ph) ph[i] = __float2half_rn(pk);
    }
}

// ---- dynamic loss scaling (torch.cuda.amp.GradScaler semantics, utils_init_nerf.py: scaler.scale(loss).backward(); scaler.step(); scaler.update())
// entirely on the device: state = float[4] {scale, growth_tracker, found_inf, good_steps}; no host read anywhere in the step.
__global__ void __launch_bounds__(256) k_scaler_check(const float *__restrict__ g, uint64_t n, float *__restrict__ state) {
    const uint64_t n4 = n / 4, stride = (uint64_t)gridDim.x * blockDim.x;
    bool bad = false;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4 *>(g)[i];
        // x - x is 0 for every finite x and NaN for +-inf / NaN
        const float t = (v.x - v.x) + (v.y - v.y) + (v.z - v.z) + (v.w - v.w);
        bad = bad || (t != 0.0f);
    }
    for (uint64_t i = n4 * 4 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) bad = bad || ((g[i] - g[i]) != 0.0f);
    if (__any(bad) && (threadIdx.x & 63) == 0) state[2] = 1.0f;          // benign race: every writer stores the same value
}

__global__ void k_scaler_update(float *state, float growth, float backoff, float interval) {
    if (state[2] != 0.0f) { state[0] *= backoff; state[1] = 0.0f; }
    else {
        state[3] += 1.0f;
        const float t = state[1] + 1.0f;
        if (t >= interval) { state[0] *= growth; state[1] = 0.0f; } else state[1] = t;
    }
    state[2] = 0.0f;
}

// ---- data-parallel exchange (customnerf_amd/dp.py).  pack: float32 gradient * scale -> float16 payload, the float32 source zeroed in the same pass
// (the next step's scatter accumulates into it); reduce: the `world` float16 slices a rank received, summed in float32, + the scaler's found-inf flag.
__global__ void __launch_bounds__(256) k_dp_pack(float *__restrict__ g, __half *__restrict__ out, uint64_t n, float scale) {
    const uint64_t n4 = n / 4, stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4 *>(g)[i];
        union { __half2 h[2]; uint2 u; } o;
        o.h[0] = __floats2half2_rn(v.x * scale, v.y * scale);
        o.h[1] = __floats2half2_rn(v.z * scale, v.w * scale);
        reinterpret_cast<uint2 *>(out)[i] = o.u;
        reinterpret_cast<float4 *>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (uint64_t i = n4 * 4 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { out[i] = __float2half_rn(g[i] * scale); g[i] = 0.f; }
}

__global__ void __launch_bounds__(256) k_dp_reduce(const __half *__restrict__ recv, uint32_t world, uint64_t shard, float *__restrict__ out,
                                                   float *__restrict__ state) {
    const uint64_t n8 = shard / 8, stride = (uint64_t)gridDim.x * blockDim.x;          // shard is a multiple of 64 elements (dp.ALIGN)
    bool bad = false;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (uint32_t r = 0; r < world; ++r) {
            union { uint4 u; __half2 h[4]; } v;
            v.u = reinterpret_cast<const uint4 *>(recv + (uint64_t)r * shard)[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float2 f = __half22float2(v.h[k]); acc[2 * k] += f.x; acc[2 * k + 1] += f.y; }
        }
        reinterpret_cast<float4 *>(out)[2 * i] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        reinterpret_cast<float4 *>(out)[2 * i + 1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += acc[k] - acc[k];
        bad = bad || (t != 0.0f);
    }
    if (state && __any(bad) && (threadIdx.x & 63) == 0) state[2] = 1.0f;
}

// k_adam with the gradient scale, the skip decision and the step count taken from the scaler state (see cnerf_adam_step_scaled)
__global__ void __launch_bounds__(256) k_adam_scaled(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
                                                     __half *__restrict__ ph, uint64_t n, float lr, float beta1, float beta2, float eps,
                                                     const float *__restrict__ state, float extra_inv, int zero_grad) {
    const bool skip = state[2] != 0.0f;
    const float gscale = extra_inv / state[0];
    const double step = (double)state[3] + 1.0;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    const float step_size = (float)((double)lr / bc1), rsqrt_bc2 = (float)(1.0 / sqrt(bc2));
    const uint64_t n4 = n / 4;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    if (skip) {                                                           // non-finite gradients: optimizer.step() is skipped, gradients are still cleared
        if (zero_grad) {
            for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) reinterpret_cast<float4 *>(g)[i] = make_float4(0, 0, 0, 0);
            for (uint64_t i = n4 * 4 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) g[i] = 0;
        }
        return;
    }
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pp = reinterpret_cast<float4 *>(p)[i], gg = reinterpret_cast<float4 *>(g)[i];
        float4 mm = reinterpret_cast<float4 *>(m)[i], vv = reinterpret_cast<float4 *>(v)[i];
        float *pa = &pp.x, *ga = &gg.x, *ma = &mm.x, *va = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float gk = ga[k] * gscale;
            ma[k] = beta1 * ma[k] + (1.0f - beta1) * gk;
            va[k] = beta2 * va[k] + (1.0f - beta2) * gk * gk;
            pa[k] -= step_size * ma[k] / (sqrtf(va[k]) * rsqrt_bc2 + eps);
        }
        reinterpret_cast<float4 *>(p)[i] = pp;
        reinterpret_cast<float4 *>(m)[i] = mm;
        reinterpret_cast<float4 *>(v)[i] = vv;
        if (zero_grad) reinterpret_cast<float4 *>(g)[i] = make_float4(0, 0, 0, 0);
        if (ph) {
            union { __half2 h[2]; uint2 u; } o;
            o.h[0] = __floats2half2_rn(pp.x, pp.y);
            o.h[1] = __floats2half2_rn(pp.z, pp.w);
            reinterpret_cast<uint2 *>(ph)[i] = o.u;
        }
    }
    for (uint64_t i = n4 * 4 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float gk = g[i] * gscale;
        const float mk = beta1 * m[i] + (1.0f - beta1) * gk;
        const float vk = beta2 * v[i] + (1.0f - beta2) * gk * gk;
        const float pk = p[i] - step_size * mk / (sqrtf(vk) * rsqrt_bc2 + eps);
        m[i] = mk; v[i] = vk; p[i] = pk;
        if (zero_grad) g[i] = 0;
        if (ph) ph[i] = __float2half_rn(pk);
    }
}

// The small parameter tensors of a step (the three MLPs: 22.5 k floats) in ONE launch — three launches of ~6 us each before round 4 — which,
// being the last Adam launch of the step, also applies GradScaler.update() (one more launch saved).  Round 6: several workgroups (one element
// per thread and job; a single workgroup walked 22 trips of dependent loads: 21 us for 90 KB), the scaler update by whichever workgroup
// finishes LAST (a device-global ticket: every workgroup has read the state — scale, found_inf, step count — before it takes its ticket).
// Same arithmetic as k_adam_scaled, element by element.
__device__ unsigned int g_adam_multi_ticket = 0;
__global__ void __launch_bounds__(1024) k_adam_scaled_multi(CnerfAdamJobs jobs, float beta1, float beta2, float eps, float *__restrict__ state,
                                                            float extra_inv, int zero_grad, int update_scaler, float growth, float backoff, float interval) {
    __shared__ double s_bc1;
    __shared__ float s_r2;
    __shared__ unsigned int s_last;
    const bool skip = state[2] != 0.0f;
    const float gscale = extra_inv / state[0];
    if (threadIdx.x == 0) {                                               // the double-precision pow()s once per workgroup, not once per thread
        const double step = (double)state[3] + 1.0;
        s_bc1 = 1.0 - pow((double)beta1, step);
        s_r2 = (float)(1.0 / sqrt(1.0 - pow((double)beta2, step)));
    }
    __syncthreads();
    const double bc1 = s_bc1;
    const float rsqrt_bc2 = s_r2;
    for (uint32_t j = 0; j < jobs.n_jobs; j++) {
        float *__restrict__ p = jobs.p[j], *__restrict__ g = jobs.g[j], *__restrict__ m = jobs.m[j], *__restrict__ v = jobs.v[j];
        __half *__restrict__ ph = reinterpret_cast<__half *>(jobs.p_half[j]);
        const float step_size = (float)((double)jobs.lr[j] / bc1);             // (the same expression as k_adam_scaled: bit-identical updates)
        const uint64_t n = jobs.n[j];
        for (uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 1024) {
            if (!skip) {
                const float gs = g[i] * gscale;
                const float mn = beta1 * m[i] + (1.0f - beta1) * gs;
                const float vn = beta2 * v[i] + (1.0f - beta2) * gs * gs;
                const float pn = p[i] - step_size * mn / (sqrtf(vn) * rsqrt_bc2 + eps);
                m[i] = mn; v[i] = vn; p[i] = pn;
                if (ph) ph[i] = __float2half_rn(pn);
            }
            if (zero_grad) g[i] = 0;
        }
    }
    if (update_scaler) {
        __syncthreads();                                                      // every thread of this workgroup has read the state
        if (threadIdx.x == 0) {
            __threadfence();
            s_last = atomicAdd(&g_adam_multi_ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
        }
        __syncthreads();
        if (s_last && threadIdx.x == 0) {                                     // ... and so has every other workgroup: they took their tickets before
            g_adam_multi_ticket = 0;
            if (skip) { state[0] *= backoff; state[1] = 0.0f; }
            else {
                state[3] += 1.0f;
                const float t = state[1] + 1.0f;
                if (t >= interval) { state[0] *= growth; state[1] = 0.0f; } else state[1] = t;
            }
            state[2] = 0.0f;
        }
    }
}

void *g_cn_stage_events[CNERF_STAGE_EVENTS] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
float *g_cn_found_inf = nullptr;

extern "C" {

int cnerf_abi_version(void) { return CNERF_ABI_VERSION; }
const char *cnerf_target_arch(void) { return "gfx950"; }
int cnerf_profile_stage_events(void *const *events, uint32_t n) {
    if (n > CNERF_STAGE_EVENTS || (n && !events)) return CNERF_EINVAL;
    for (uint32_t i = 0; i < CNERF_STAGE_EVENTS; i++) g_cn_stage_events[i] = i < n ? events[i] : nullptr;
    return CNERF_OK;
}

int cnerf_generate_rays(const float *c2w, uint32_t V, uint32_t H, uint32_t W, float fx, float fy, float cx, float cy, float level,
                        int convention, float *origins, float *directions, void *stream) {
    if (!c2w || !origins || !directions) return CNERF_ENULL;
    if (convention < 0 || convention > 1 || fx == 0.0f || fy == 0.0f) return CNERF_EINVAL;
    if (V == 0 || H == 0 || W == 0) return CNERF_OK;
    const RayDistortion none = {0, 0, 0, 0, 0, 0};
    hipLaunchKernelGGL(k_generate_rays, dim3(cn_div_up(H * W, 256), V), dim3(256), 0, CN_STREAM(stream), c2w, V, H, W, fx, fy, cx, cy, level,
                       convention, none, origins, directions);
    return cn_launch_status();
}

int cnerf_generate_rays_fisheye(const float *c2w, uint32_t V, uint32_t H, uint32_t W, float fx, float fy, float cx, float cy, float level,
                                const float *distortion_host, float *origins, float *directions, void *stream) {
    if (!c2w || !origins || !directions || !distortion_host) return CNERF_ENULL;
    if (fx == 0.0f || fy == 0.0f) return CNERF_EINVAL;
    if (V == 0 || H == 0 || W == 0) return CNERF_OK;
    const RayDistortion k = {distortion_host[0], distortion_host[1], distortion_host[2], distortion_host[3], distortion_host[4], distortion_host[5]};
    hipLaunchKernelGGL(k_generate_rays, dim3(cn_div_up(H * W, 256), V), dim3(256), 0, CN_STREAM(stream), c2w, V, H, W, fx, fy, cx, cy, level, 2, k,
                       origins, directions);
    return cn_launch_status();
}

int cnerf_adam_step(float *p, float *g, float *m, float *v, void *p_half, uint64_t n, float lr, float beta1, float beta2, float eps,
                    uint32_t step, float grad_scale_inv, int zero_grad, void *stream) {
    if (!p || !g || !m || !v) return CNERF_ENULL;
    if (step == 0) return CNERF_EINVAL;
    if (n == 0) return CNERF_OK;
    if ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) return CNERF_EINVAL;
    if (p_half && (((uintptr_t)p_half) & 7)) return CNERF_EINVAL;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float rsqrt_bc2 = (float)(1.0 / sqrt(bc2));
    const uint64_t want = cn_div_up64(cn_div_up64(n, 4), 256);
    const uint32_t blocks = (uint32_t)(want < 4096 ? (want ? want : 1) : 4096);
    hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, CN_STREAM(stream), p, g, m, v, (__half *)p_half, n, step_size, beta1, beta2, eps,
                       rsqrt_bc2, grad_scale_inv, zero_grad);
    return cn_launch_status();
}

int cnerf_scaler_check(const float *g, uint64_t n, float *state, void *stream) {
    if (!state || (!g && n)) return CNERF_ENULL;
    if (n == 0) return CNERF_OK;
    if (((uintptr_t)g) & 15) return CNERF_EINVAL;
    const uint64_t want = cn_div_up64(cn_div_up64(n, 4), 256);
    const uint32_t blocks = (uint32_t)(want < 2048 ? (want ? want : 1) : 2048);
    hipLaunchKernelGGL(k_scaler_check, dim3(blocks), dim3(256), 0, CN_STREAM(stream), g, n, state);
    return cn_launch_status();
}

int cnerf_stream_capture_id(void *stream, uint64_t *id) {
    if (!id) return CNERF_ENULL;
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    unsigned long long cid = 0;
    const hipError_t e = hipStreamGetCaptureInfo(CN_STREAM(stream), &status, &cid);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    *id = status == hipStreamCaptureStatusActive ? (uint64_t)cid : 0;
    return CNERF_OK;
}

int cnerf_scaler_watch(float *state) {
    g_cn_found_inf = state ? state + 2 : nullptr;
    return CNERF_OK;
}

int cnerf_scaler_update(float *state, float growth_factor, float backoff_factor, uint32_t growth_interval, void *stream) {
    if (!state) return CNERF_ENULL;
    if (!(growth_factor >= 1.0f) || !(backoff_factor > 0.0f && backoff_factor <= 1.0f) || growth_interval == 0) return CNERF_EINVAL;
    hipLaunchKernelGGL(k_scaler_update, dim3(1), dim3(1), 0, CN_STREAM(stream), state, growth_factor, backoff_factor, (float)growth_interval);
    return cn_launch_status();
}

int cnerf_adam_step_scaled(float *p, float *g, float *m, float *v, void *p_half, uint64_t n, float lr, float beta1, float beta2, float eps,
                           const float *state, float extra_inv, int zero_grad, void *stream) {
    if (!p || !g || !m || !v || !state) return CNERF_ENULL;
    if (n == 0) return CNERF_OK;
    if ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) return CNERF_EINVAL;
    if (p_half && (((uintptr_t)p_half) & 7)) return CNERF_EINVAL;
    const uint64_t want = cn_div_up64(cn_div_up64(n, 4), 256);
    const uint32_t blocks = (uint32_t)(want < 4096 ? (want ? want : 1) : 4096);
    hipLaunchKernelGGL(k_adam_scaled, dim3(blocks), dim3(256), 0, CN_STREAM(stream), p, g, m, v, (__half *)p_half, n, lr, beta1, beta2, eps, state,
                       extra_inv, zero_grad);
    return cn_launch_status();
}

int cnerf_adam_step_scaled_multi(const CnerfAdamJobs *jobs, float beta1, float beta2, float eps, float *state, float extra_inv, int zero_grad,
                                 int update_scaler, float growth_factor, float backoff_factor, uint32_t growth_interval, void *stream) {
    if (!jobs || !state) return CNERF_ENULL;
    if (jobs->n_jobs > CNERF_ADAM_MAX_JOBS) return CNERF_EINVAL;
    if (update_scaler && (!(growth_factor >= 1.0f) || !(backoff_factor > 0.0f && backoff_factor <= 1.0f) || growth_interval == 0)) return CNERF_EINVAL;
    for (uint32_t j = 0; j < jobs->n_jobs; j++)
        if (!jobs->p[j] || !jobs->g[j] || !jobs->m[j] || !jobs->v[j]) return CNERF_ENULL;
    if (jobs->n_jobs == 0 && !update_scaler) return CNERF_OK;
    uint64_t max_n = 1;
    for (uint32_t j = 0; j < jobs->n_jobs; j++) max_n = max_n > jobs->n[j] ? max_n : jobs->n[j];
    const uint32_t blocks = (uint32_t)(cn_div_up64(max_n, 1024) < 64 ? cn_div_up64(max_n, 1024) : 64);
    hipLaunchKernelGGL(k_adam_scaled_multi, dim3(blocks), dim3(1024), 0, CN_STREAM(stream), *jobs, beta1, beta2, eps, state, extra_inv, zero_grad, update_scaler,
                       growth_factor, backoff_factor, (float)growth_interval);
    return cn_launch_status();
}

int cnerf_dp_pack(float *grad, void *payload_half, uint64_t n, float scale, void *stream) {
    if (!grad || !payload_half) return CNERF_ENULL;
    if (n == 0) return CNERF_OK;
    if ((((uintptr_t)grad) & 15) || (((uintptr_t)payload_half) & 7)) return CNERF_EINVAL;
    const uint64_t want = cn_div_up64(cn_div_up64(n, 4), 256);
    const uint32_t blocks = (uint32_t)(want < 4096 ? (want ? want : 1) : 4096);
    hipLaunchKernelGGL(k_dp_pack, dim3(blocks), dim3(256), 0, CN_STREAM(stream), grad, (__half *)payload_half, n, scale);
    return cn_launch_status();
}

int cnerf_dp_reduce(const void *recv_half, uint32_t world, uint64_t shard, float *out, float *scaler_state, void *stream) {
    if (!recv_half || !out) return CNERF_ENULL;
    if (world == 0 || (shard & 63)) return CNERF_EINVAL;
    if (shard == 0) return CNERF_OK;
    if ((((uintptr_t)recv_half) | ((uintptr_t)out)) & 15) return CNERF_EINVAL;
    const uint64_t want = cn_div_up64(shard / 8, 256);
    const uint32_t blocks = (uint32_t)(want < 4096 ? (want ? want : 1) : 4096);
    hipLaunchKernelGGL(k_dp_reduce, dim3(blocks), dim3(256), 0, CN_STREAM(stream), (const __half *)recv_half, world, shard, out, scaler_state);
    return cn_launch_status();
}

}  // extern "C"
