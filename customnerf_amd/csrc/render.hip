// The pure-PyTorch renderer `NeRFRenderer.run` (nerf/renderer.py:278-405) as four gfx950 kernels:
//   k_sample_coarse        stratified samples + jitter + aabb clip                      (renderer.py:310-322)
//   k_sample_fine_merge    coarse weights -> inverse-CDF resampling -> merge           (renderer.py:334-363 + sample_pdf :21-55)
//   k_composite_run_fwd    the three weights_sum_i composites (all / fg / bg) at once  (renderer.py:384-402, 407-474)
//   k_composite_run_bwd    their gradient w.r.t. sigma and rgb+confidence
// One wavefront owns one ray: its samples sit on the 64 lanes (chunks of 64), transmittance is a multiplicative wave scan
// (DPP shuffles), the CDF inversion a binary search in LDS, the merge a rank computation — no sort, no global scratch.
// The reference spends ~100 elementwise/scan/sort/gather launches and three cumprod passes per step on the same work.
#include "common.h"
#include <float.h>

#define RN_WAVES 4
#define RN_THREADS (RN_WAVES * 64)
#define RN_MAXS 128                  // max coarse / fine samples per ray

__device__ __forceinline__ float rn_linspace01(uint32_t i, uint32_t n) {      // torch.linspace(0, 1, n)[i]
    const float step = 1.0f / (float)(n - 1);
    return (i < n / 2) ? step * (float)i : 1.0f - step * (float)(n - 1 - i);
}

// torch.clamp((z - near) / (far - near), 0, 1): NaN (a ray that misses the box has far == near) propagates, as in torch
__device__ __forceinline__ float rn_norm_depth(float z, float near, float far) {
    const float x = (z - near) / (far - near);
    return (x != x) ? x : fminf(fmaxf(x, 0.0f), 1.0f);
}

__device__ __forceinline__ void rn_point(const float *o, const float *d, float z, const float *aabb, float *out) {
#pragma unroll
    for (int c = 0; c < 3; c++) out[c] = fminf(fmaxf(o[c] + d[c] * z, aabb[c]), aabb[3 + c]);
}

__global__ void __launch_bounds__(256) k_sample_coarse(const float *__restrict__ rays_o, const float *__restrict__ rays_d, const float *__restrict__ nears,
                                                       const float *__restrict__ fars, const float *__restrict__ aabb, const float *__restrict__ noise,
                                                       uint32_t N, uint32_t T, float *__restrict__ z_vals, float *__restrict__ xyzs,
                                                       float *__restrict__ unit, float bound, float *__restrict__ nears_out, float *__restrict__ fars_out,
                                                       float min_near) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * T) return;
    const uint32_t n = idx / T, i = idx - n * T;
    float near, far;
    if (nears_out) {                                          // near_far_from_aabb folded in (renderer.py:297): every thread of a ray repeats the slab test, sample 0 stores it
        cn_near_far(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, aabb, min_near, near, far);
        if (i == 0) { nears_out[n] = near; fars_out[n] = far; }
    } else {
        near = nears[n]; far = fars[n];
    }
    float z = near + (far - near) * rn_linspace01(i, T);
    if (noise) z = z + (noise[idx] - 0.5f) * ((far - near) / (float)T);
    float p[3];
    rn_point(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, z, aabb, p);
    z_vals[idx] = z;
    xyzs[(size_t)idx * 3] = p[0]; xyzs[(size_t)idx * 3 + 1] = p[1]; xyzs[(size_t)idx * 3 + 2] = p[2];
    if (unit) {                                               // the grid's [0,1] coordinates, (x + bound) / (2 bound) as gridencoder/grid.py:156 computes them
        // torch divides a tensor by a host scalar as a multiplication by its float reciprocal (BinaryDivTrueKernel.cu): so does this
        const float inv = 1.0f / (2.0f * bound);
        unit[(size_t)idx * 3] = (p[0] + bound) * inv; unit[(size_t)idx * 3 + 1] = (p[1] + bound) * inv; unit[(size_t)idx * 3 + 2] = (p[2] + bound) * inv;
    }
}

// Wave scans on the DPP data path (row_shr inside the 16-lane rows, row_bcast:15 / :31 across them): no LDS-crossbar round trips
// (ds_bpermute) in the per-ray dependency chains.  `old` is what a lane keeps when its DPP source lane does not exist (the identity).
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float rn_dpp(float old, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL, ROW_MASK, 0xF, false));
}
#define RN_ROW_SHR(n) (0x110 + (n))
#define RN_ROW_BCAST15 0x142
#define RN_ROW_BCAST31 0x143
#define RN_WAVE_SHR1 0x138

__device__ __forceinline__ float rn_wave_incl_prod(float x) {
    x *= rn_dpp<RN_ROW_SHR(1)>(1.0f, x);
    x *= rn_dpp<RN_ROW_SHR(2)>(1.0f, x);
    x *= rn_dpp<RN_ROW_SHR(4)>(1.0f, x);
    x *= rn_dpp<RN_ROW_SHR(8)>(1.0f, x);
    x *= rn_dpp<RN_ROW_BCAST15, 0xA>(1.0f, x);
    x *= rn_dpp<RN_ROW_BCAST31, 0xC>(1.0f, x);
    return x;
}
__device__ __forceinline__ float rn_wave_incl_sum(float x) {
    x += rn_dpp<RN_ROW_SHR(1)>(0.0f, x);
    x += rn_dpp<RN_ROW_SHR(2)>(0.0f, x);
    x += rn_dpp<RN_ROW_SHR(4)>(0.0f, x);
    x += rn_dpp<RN_ROW_SHR(8)>(0.0f, x);
    x += rn_dpp<RN_ROW_BCAST15, 0xA>(0.0f, x);
    x += rn_dpp<RN_ROW_BCAST31, 0xC>(0.0f, x);
    return x;
}
__device__ __forceinline__ float rn_lane63(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}

// exclusive multiplicative scan of v over the 64 lanes, times carry; returns the wave total (times carry) in `carry`
__device__ __forceinline__ float rn_excl_prod_scan(float v, float &carry, uint32_t lane) {
    (void)lane;
    const float incl = rn_wave_incl_prod(v);
    const float excl = rn_dpp<RN_WAVE_SHR1>(1.0f, incl);
    const float res = excl * carry;
    carry = carry * rn_lane63(incl);
    return res;
}

__device__ __forceinline__ float rn_incl_sum_scan(float v, float &carry, uint32_t lane) {
    (void)lane;
    const float incl = rn_wave_incl_sum(v);
    const float res = incl + carry;
    carry = carry + rn_lane63(incl);
    return res;
}

__device__ __forceinline__ float rn_wave_sum(float v) { return rn_lane63(rn_wave_incl_sum(v)); }

__global__ void __launch_bounds__(RN_THREADS) k_sample_fine_merge(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                                  const float *__restrict__ nears, const float *__restrict__ fars,
                                                                  const float *__restrict__ aabb, const float *__restrict__ z_vals,
                                                                  const float *__restrict__ sigmas, const float *__restrict__ u_rand, uint32_t N,
                                                                  uint32_t T, uint32_t t, float *__restrict__ z_all, float *__restrict__ xyz_all,
                                                                  float *__restrict__ xyz_fine, uint32_t *__restrict__ src_index, float *__restrict__ unit_fine,
                                                                  float bound) {
    __shared__ float s_z[RN_WAVES][RN_MAXS], s_w[RN_WAVES][RN_MAXS], s_cdf[RN_WAVES][RN_MAXS], s_bin[RN_WAVES][RN_MAXS];
    __shared__ __attribute__((aligned(16))) float s_nz[RN_WAVES][RN_MAXS];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n = blockIdx.x * RN_WAVES + wave;
    if (n >= N) return;                                   // whole wave exits together; no workgroup barrier below
    float *zz = s_z[wave], *ww = s_w[wave], *cdf = s_cdf[wave], *bins = s_bin[wave], *nz = s_nz[wave];
    const float near = nears[n], far = fars[n];
    const float sd = (far - near) / (float)T;
    const float *zr = z_vals + (size_t)n * T, *sr = sigmas + (size_t)n * T;

    for (uint32_t i = lane; i < T; i += 64) zz[i] = zr[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // weights = alpha * cumprod([1, 1 - alpha + 1e-15])[:-1]          (renderer.py:336-341)
    float carry = 1.0f;
    for (uint32_t base = 0; base < T; base += 64) {
        const uint32_t i = base + lane;
        float alpha = 0.0f, delta = 0.0f;
        if (i < T) {
            delta = (i + 1 < T) ? zz[i + 1] - zz[i] : sd;
            alpha = 1.0f - expf(-delta * sr[i]);
        }
        const float tr = rn_excl_prod_scan((i < T) ? (1.0f - alpha + 1e-15f) : 1.0f, carry, lane);
        if (i < T) {
            ww[i] = alpha * tr;
            bins[i] = zz[i] + 0.5f * delta;               // z_vals_mid (only i < T-1 is used)
        }
    }
    __builtin_amdgcn_wave_barrier();
    // pdf over weights[1:-1] + 1e-5, cdf = [0, cumsum(pdf)]           (sample_pdf :28-31)
    const uint32_t nb = T - 2;                            // number of pdf bins; cdf has nb + 1 = T - 1 entries
    float tot = 0.0f;
    for (uint32_t k = lane; k < nb; k += 64) tot += ww[k + 1] + 1e-5f;
    tot = rn_wave_sum(tot);
    float csum = 0.0f;
    for (uint32_t base = 0; base < nb; base += 64) {
        const uint32_t k = base + lane;
        const float pdf = (k < nb) ? (ww[k + 1] + 1e-5f) / tot : 0.0f;
        const float c = rn_incl_sum_scan(pdf, csum, lane);
        if (k < nb) cdf[k + 1] = c;
    }
    if (lane == 0) cdf[0] = 0.0f;
    __builtin_amdgcn_wave_barrier();
    // invert the cdf                                                   (sample_pdf :33-53)
    const uint32_t ncdf = nb + 1;
    for (uint32_t m = lane; m < t; m += 64) {
        float u;
        if (u_rand) u = u_rand[(size_t)n * t + m];
        else {                                            // torch.linspace(0.5/t, 1 - 0.5/t, t)[m]
            const float a = 0.0f + 0.5f / (float)t, b = 1.0f - 0.5f / (float)t;
            const float step = (b - a) / (float)(t - 1);
            u = (m < t / 2) ? a + step * (float)m : b - step * (float)(t - 1 - m);
        }
        uint32_t lo = 0, hi = ncdf;                       // searchsorted(right=True): first index with cdf > u
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
        }
        const uint32_t below = lo > 0 ? lo - 1 : 0, above = lo < ncdf - 1 ? lo : ncdf - 1;
        const float cb = cdf[below], ca = cdf[above];
        float denom = ca - cb;
        if (denom < 1e-5f) denom = 1.0f;
        const float tt = (u - cb) / denom;
        nz[m] = bins[below] + tt * (bins[above] - bins[below]);
    }
    // (the rank loops below read the fine samples four at a time: pad to a multiple of four with +inf, which no comparison counts)
    for (uint32_t m = t + lane; m < ((t + 3u) & ~3u); m += 64) nz[m] = __builtin_inff();
    __builtin_amdgcn_wave_barrier();
    // merge by rank: position of a coarse sample = its index + #fine < it; of a fine sample = its rank among the fine ones
    // (ties by index) + #coarse <= it.  Equal values give equal samples, so any tie order reproduces torch.sort's output.
    // Round 6: 16-byte broadcast reads of the fine samples (a scalar LDS read per comparison was a 192-step latency chain per lane) and a
    // binary search in the coarse samples, which are sorted (stratified draw; the rank rule above already relies on it).
    const float4 *nz4 = reinterpret_cast<const float4 *>(nz);
    const uint32_t t4 = (t + 3u) >> 2;
    const float *o = rays_o + (size_t)n * 3, *d = rays_d + (size_t)n * 3;
    // Two output forms.  Merged (xyz_all): positions in sorted order, what the reference evaluates.  Split (xyz_fine + src_index): the
    // new samples stay in their own block [N, t, 3] behind the coarse block [N, T, 3] (whose grid features are already computed),
    // and src_index[n][pos] = row of the sorted sample `pos` in that [coarse | fine] sample list — the compositing kernels read through it.
    float *za = z_all + (size_t)n * (T + t), *xa = xyz_all ? xyz_all + (size_t)n * (T + t) * 3 : nullptr;
    uint32_t *si = src_index ? src_index + (size_t)n * (T + t) : nullptr;
    for (uint32_t i = lane; i < T; i += 64) {
        const float v = zz[i];
        uint32_t pos = i;
        for (uint32_t m = 0; m < t4; m++) {
            const float4 w = nz4[m];
            pos += ((w.x < v) ? 1u : 0u) + ((w.y < v) ? 1u : 0u) + ((w.z < v) ? 1u : 0u) + ((w.w < v) ? 1u : 0u);
        }
        za[pos] = v;
        if (xa) {
            float p[3];
            rn_point(o, d, v, aabb, p);
            xa[pos * 3] = p[0]; xa[pos * 3 + 1] = p[1]; xa[pos * 3 + 2] = p[2];
        }
        if (si) si[pos] = n * T + i;
    }
    for (uint32_t m = lane; m < t; m += 64) {
        const float v = nz[m];
        uint32_t pos = 0;
        for (uint32_t k = 0; k < t4; k++) {
            const float4 w = nz4[k];
            pos += ((w.x < v || (w.x == v && 4 * k < m)) ? 1u : 0u) + ((w.y < v || (w.y == v && 4 * k + 1 < m)) ? 1u : 0u) +
                   ((w.z < v || (w.z == v && 4 * k + 2 < m)) ? 1u : 0u) + ((w.w < v || (w.w == v && 4 * k + 3 < m)) ? 1u : 0u);
        }
        {
            uint32_t lo = 0, hi = T;                          // #coarse <= v = first index with zz > v
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (zz[mid] <= v) lo = mid + 1; else hi = mid;
            }
            pos += lo;
        }
        float p[3];
        rn_point(o, d, v, aabb, p);
        za[pos] = v;
        if (xa) { xa[pos * 3] = p[0]; xa[pos * 3 + 1] = p[1]; xa[pos * 3 + 2] = p[2]; }
        if (xyz_fine) {
            // draw order, NOT sorted along the ray: measured 1 % faster end to end (sorted neighbours collide in the scatter's LDS atomics)
            float *xf = xyz_fine + ((size_t)n * t + m) * 3;
            xf[0] = p[0]; xf[1] = p[1]; xf[2] = p[2];
            if (unit_fine) {                                  // grid coordinates of the new samples, as k_sample_coarse writes them
                float *uf = unit_fine + ((size_t)n * t + m) * 3;
                const float inv = 1.0f / (2.0f * bound);
                uf[0] = (p[0] + bound) * inv; uf[1] = (p[1] + bound) * inv; uf[2] = (p[2] + bound) * inv;
            }
        }
        if (si) si[pos] = N * T + n * t + m;
    }
}

// ------------------------------------------------------------------------------------------------ sample_pdf as its own entry point
// `sample_pdf(bins, weights, n_samples, det)` (renderer.py:21-55) for arbitrary bins / weights (run() itself goes through the fused
// k_sample_fine_merge above): one wave per row, pdf / cdf as a wave scan, inverse-CDF search in LDS.  bins [B, n_bins], weights [B, n_bins - 1]
// -> samples [B, n_samples]; u [B, n_samples] replays the draw of :37, NULL = det (the midpoint linspace of :33-35).
#define RN_PDF_MAXB 256
__global__ void __launch_bounds__(RN_THREADS) k_sample_pdf(const float *__restrict__ bins, const float *__restrict__ weights, const float *__restrict__ u_rand,
                                                           uint32_t B, uint32_t n_bins, uint32_t n_samples, float *__restrict__ samples) {
    __shared__ float s_cdf[RN_WAVES][RN_PDF_MAXB], s_bin[RN_WAVES][RN_PDF_MAXB];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n = blockIdx.x * RN_WAVES + wave;
    if (n >= B) return;                                   // whole wave exits together; no workgroup barrier below
    float *cdf = s_cdf[wave], *bn = s_bin[wave];
    const uint32_t nw = n_bins - 1;
    const float *wr = weights + (size_t)n * nw, *br = bins + (size_t)n * n_bins;
    for (uint32_t i = lane; i < n_bins; i += 64) bn[i] = br[i];
    float tot = 0.0f;
    for (uint32_t k = lane; k < nw; k += 64) tot += wr[k] + 1e-5f;
    tot = rn_wave_sum(tot);
    float csum = 0.0f;
    for (uint32_t base = 0; base < nw; base += 64) {
        const uint32_t k = base + lane;
        const float pdf = (k < nw) ? (wr[k] + 1e-5f) / tot : 0.0f;
        const float c = rn_incl_sum_scan(pdf, csum, lane);
        if (k < nw) cdf[k + 1] = c;
    }
    if (lane == 0) cdf[0] = 0.0f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t ncdf = n_bins;
    for (uint32_t m = lane; m < n_samples; m += 64) {
        float u;
        if (u_rand) u = u_rand[(size_t)n * n_samples + m];
        else {                                            // torch.linspace(0.5/t, 1 - 0.5/t, t)[m]
            const float a = 0.0f + 0.5f / (float)n_samples, b = 1.0f - 0.5f / (float)n_samples;
            const float step = n_samples > 1 ? (b - a) / (float)(n_samples - 1) : 0.0f;
            u = (m < n_samples / 2) ? a + step * (float)m : b - step * (float)(n_samples - 1 - m);
        }
        uint32_t lo = 0, hi = ncdf;                       // searchsorted(right=True): first index with cdf > u
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
        }
        const uint32_t below = lo > 0 ? lo - 1 : 0, above = lo < ncdf - 1 ? lo : ncdf - 1;
        const float cb = cdf[below], ca = cdf[above];
        float denom = ca - cb;
        if (denom < 1e-5f) denom = 1.0f;
        const float tt = (u - cb) / denom;
        samples[(size_t)n * n_samples + m] = bn[below] + tt * (bn[above] - bn[below]);
    }
}

// ------------------------------------------------------------------------------------------------ composites
// per-sample quantities of one variant (0 all, 1 fg, 2 bg)
__device__ __forceinline__ float rn_edit(float conf, int soft, float thr) {
    return soft ? 1.0f / (1.0f + expf(-(conf - thr) * 100.0f)) : (conf > 0.5f ? 1.0f : 0.0f);
}
__device__ __forceinline__ float rn_variant_scale(int v, float e) { return v == 0 ? 1.0f : (v == 1 ? e : 1.0f - e); }

#define RN_MAXCH 4                   // up to 256 samples per ray (4 chunks of 64)

__global__ void __launch_bounds__(RN_THREADS) k_composite_run_fwd(const float *__restrict__ sigmas, const float *__restrict__ rgbc,
                                                                  const float *__restrict__ z_vals, const float *__restrict__ nears,
                                                                  const float *__restrict__ fars, uint32_t N, uint32_t S, uint32_t num_steps, int soft,
                                                                  float thr, float *__restrict__ out_ray, float *__restrict__ out_w,
                                                                  const uint32_t *__restrict__ src_index, float *__restrict__ sigma_m,
                                                                  float *__restrict__ rgbc_m, uint32_t vmask) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n = blockIdx.x * RN_WAVES + wave;
    if (n >= N) return;
    const float near = nears[n], far = fars[n];
    const float sd = (far - near) / (float)num_steps;
    const float *zr = z_vals + (size_t)n * S;
    const float4 *c4 = reinterpret_cast<const float4 *>(rgbc);
    // vmask: bit v set = variant v wanted (cnerf_composite_run_indexed_variants); the others are written as zeros without their scans / exponentials
    float carry[3] = {1.0f, 1.0f, 1.0f};
    float acc[3][6];
#pragma unroll
    for (int v = 0; v < 3; v++)
#pragma unroll
        for (int k = 0; k < 6; k++) acc[v][k] = 0.0f;
    for (uint32_t base = 0; base < S; base += 64) {
        const uint32_t i = base + lane;
        const bool ok = i < S;
        float z = 0, delta = 0, sigma = 0;
        float4 c = make_float4(0, 0, 0, 0);
        if (ok) {
            z = zr[i];
            delta = (i + 1 < S) ? zr[i + 1] - z : sd;
            const size_t row = src_index ? (size_t)src_index[(size_t)n * S + i] : (size_t)n * S + i;     // sorted position -> sample row
            sigma = sigmas[row];
            c = c4[row];
            if (sigma_m) sigma_m[(size_t)n * S + i] = sigma;                                               // sorted-order copies for the result dict
            if (rgbc_m) reinterpret_cast<float4 *>(rgbc_m)[(size_t)n * S + i] = c;
        }
        const float zn = rn_norm_depth(z, near, far);
        const float e = rn_edit(c.w, soft, thr);
#pragma unroll
        for (int v = 0; v < 3; v++) {
            if (!((vmask >> v) & 1u)) {
                if (ok && out_w) out_w[((size_t)v * N + n) * S + i] = 0.0f;
                continue;
            }
            const float alpha = ok ? 1.0f - expf(-delta * (sigma * rn_variant_scale(v, e))) : 0.0f;
            const float tr = rn_excl_prod_scan(ok ? (1.0f - alpha + 1e-15f) : 1.0f, carry[v], lane);
            const float w = alpha * tr;
            if (ok) {
                acc[v][0] += w * c.x; acc[v][1] += w * c.y; acc[v][2] += w * c.z;
                acc[v][3] += w * zn; acc[v][4] += w; acc[v][5] += w * c.w;
                if (out_w) out_w[((size_t)v * N + n) * S + i] = w;
            }
        }
    }
#pragma unroll
    for (int v = 0; v < 3; v++) {
        if (!((vmask >> v) & 1u)) {
            if (lane < 6) out_ray[((size_t)v * N + n) * 6 + lane] = 0.0f;
            continue;
        }
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const float s = rn_wave_sum(acc[v][k]);
            if (lane == 0) out_ray[((size_t)v * N + n) * 6 + k] = s;
        }
    }
}

// Backward.  For one variant: w_i = a_i T_i, T_i = prod_{j<i} q_j, q = 1 - a + 1e-15, a = 1 - exp(-delta * s_v).
//   G_i = dL/dw_i = g_img . rgb_i + g_depth zn_i + g_ws + g_mask conf_i
//   dL/da_i = G_i T_i - (sum_{k>i} G_k w_k) / q_i ;  da/ds_v = delta (1 - a)
template <int NCH>                                  // 64-sample chunks per ray: ceil(S / 64)
__global__ void __launch_bounds__(RN_THREADS) k_composite_run_bwd(const float *__restrict__ g_out, const float *__restrict__ sigmas,
                                                                  const float *__restrict__ rgbc, const float *__restrict__ z_vals,
                                                                  const float *__restrict__ nears, const float *__restrict__ fars, uint32_t N, uint32_t S,
                                                                  uint32_t num_steps, int soft, float thr, int detach_bg, int detach_mask,
                                                                  float *__restrict__ g_sigma, float *__restrict__ g_rgbc,
                                                                  const uint32_t *__restrict__ src_index, float flush_thr, uint8_t *__restrict__ tile_live,
                                                                  uint32_t T_coarse) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n = blockIdx.x * RN_WAVES + wave;
    if (n >= N) return;
    const float near = nears[n], far = fars[n];
    const float sd = (far - near) / (float)num_steps;
    const float *zr = z_vals + (size_t)n * S;
    const float4 *c4 = reinterpret_cast<const float4 *>(rgbc);
    // early termination (flush_thr > 0: the consumer is the half-precision fused field): a sample whose gradients, as k_field_bwd_x2 will
    // consume them — half(g_sigma * exp'), half(g_c * sigmoid') — are all zero contributes exactly nothing to any weight or table entry;
    // behind an opaque surface (T underflows) and in empty space (sigma -> 0) that is most of a fitted scene's samples.  Such rows are
    // written as exact zeros here, and the ray's 32-sample tiles of the sample list that hold nothing else are reported dead (tile_live),
    // so that the field backward skips them and the scatter emits no records for them — bit-identical gradients, less work.
    const float e_lo = __expf(-15.0f), e_hi = __expf(15.0f);
    uint32_t live_bits = 0;                                 // bit k: local tile k of this ray (T_coarse / 32 coarse tiles, then the fine ones) has a live row
    float go[3][6];
#pragma unroll
    for (int v = 0; v < 3; v++)
#pragma unroll
        for (int k = 0; k < 6; k++) go[v][k] = g_out[((size_t)v * N + n) * 6 + k];
    // A variant none of whose six outputs received a gradient adds exact zeros to every sample gradient of the ray: skip its transmittance scans and
    // exponentials (wave-uniform: the wave owns the ray).  The reconstruction loss touches the first variant only (k_recon_loss writes zeros for the
    // other two), the editing losses all three.  (`!(x == 0)`: a NaN gradient keeps its variant.)
    bool use[3];
#pragma unroll
    for (int v = 0; v < 3; v++) {
        bool any = false;
#pragma unroll
        for (int k = 0; k < 6; k++) any = any || !(go[v][k] == 0.0f);
        use[v] = __builtin_amdgcn_readfirstlane(any ? 1 : 0) != 0;
    }

    // pass 1: the ray's samples into registers (one per lane and chunk), alpha / transmittance per variant, totals sum_k G_k w_k.  Pass 2 reuses all of
    // it (round 6: it used to load and recompute everything a second time — 31 -> 25.5 us for 2.1 M samples)
    float z_[NCH], delta_[NCH], sigma_[NCH], alpha_[3][NCH], tr_[3][NCH];
    float4 c_[NCH];
    uint32_t row_[NCH];
    float tot[3] = {0, 0, 0};
    {
        float carry[3] = {1.0f, 1.0f, 1.0f};
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) {
            const uint32_t i = ch * 64 + lane;
            const bool ok = i < S;
            float z = 0, delta = 0, sigma = 0;
            float4 c = make_float4(0, 0, 0, 0);
            uint32_t row = 0;
            if (ok) {
                z = zr[i]; delta = (i + 1 < S) ? zr[i + 1] - z : sd;
                row = src_index ? src_index[(size_t)n * S + i] : n * S + i;
                sigma = sigmas[row]; c = c4[row];
            }
            z_[ch] = z; delta_[ch] = delta; sigma_[ch] = sigma; c_[ch] = c; row_[ch] = row;
            const float zn = rn_norm_depth(z, near, far);
            const float e = rn_edit(c.w, soft, thr);
#pragma unroll
            for (int v = 0; v < 3; v++) {
                if (!use[v]) continue;
                const float alpha = ok ? 1.0f - expf(-delta * (sigma * rn_variant_scale(v, e))) : 0.0f;
                const float tr = rn_excl_prod_scan(ok ? (1.0f - alpha + 1e-15f) : 1.0f, carry[v], lane);
                alpha_[v][ch] = alpha; tr_[v][ch] = tr;
                const float G = go[v][0] * c.x + go[v][1] * c.y + go[v][2] * c.z + go[v][3] * zn + go[v][4] + (detach_mask ? 0.0f : go[v][5] * c.w);
                if (ok) tot[v] += G * alpha * tr;
            }
        }
#pragma unroll
        for (int v = 0; v < 3; v++)
            if (use[v]) tot[v] = rn_wave_sum(tot[v]);
    }
    // pass 2: gradients
    float pref[3] = {0, 0, 0};
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
        const uint32_t i = ch * 64 + lane;
        const bool ok = i < S;
        const float z = z_[ch], delta = delta_[ch], sigma = sigma_[ch];
        const float4 c = c_[ch];
        const uint32_t row = row_[ch];
        const float zn = rn_norm_depth(z, near, far);
        const float e = rn_edit(c.w, soft, thr);
        float gs = 0.0f, ge = 0.0f;
        float4 gc = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 3; v++) {
            if (!use[v]) continue;
            const float m = rn_variant_scale(v, e);
            const float alpha = alpha_[v][ch];
            const float q = 1.0f - alpha + 1e-15f;
            const float tr = tr_[v][ch];
            const float w = alpha * tr;
            const float G = go[v][0] * c.x + go[v][1] * c.y + go[v][2] * c.z + go[v][3] * zn + go[v][4] + (detach_mask ? 0.0f : go[v][5] * c.w);
            const float incl = rn_incl_sum_scan(ok ? G * w : 0.0f, pref[v], lane);      // sum_{k<=i} G_k w_k
            const float suffix = tot[v] - incl;
            const float dalpha = G * tr - suffix / q;
            const float dsv = dalpha * delta * (1.0f - alpha);
            const bool detached = (v == 0) && detach_bg && !(c.w >= 0.5f);                 // renderer.py:409-418 (is_all call only)
            if (!detached) {
                gs += dsv * m;
                gc.x += go[v][0] * w; gc.y += go[v][1] * w; gc.z += go[v][2] * w;
            }
            gc.w += go[v][5] * w;                                                          // d render_mask / d conf
            if (v == 1) ge += dsv * sigma;
            if (v == 2) ge -= dsv * sigma;
        }
        if (soft) gc.w += ge * 100.0f * e * (1.0f - e);                                    // edit = sigmoid((conf - thr) * 100), renderer.py:387
        if (flush_thr > 0.0f) {
            // conservative by a factor of two against the half rounding threshold 2^-25 (last-bit differences between the forward's values
            // and the backward kernel's recomputation cannot revive a row); NaN / Inf gradients compare false and stay
            const float sc = fminf(fmaxf(sigma, e_lo), e_hi);                                // exp(clamp(., -15, 15)) of provider_utils.py:26-29
            const bool dead = fabsf(gs) * sc < flush_thr && fabsf(gc.x) * (c.x * (1.0f - c.x)) < flush_thr && fabsf(gc.y) * (c.y * (1.0f - c.y)) < flush_thr &&
                              fabsf(gc.z) * (c.z * (1.0f - c.z)) < flush_thr && fabsf(gc.w) * (c.w * (1.0f - c.w)) < flush_thr;
            if (dead) { gs = 0.0f; gc = make_float4(0, 0, 0, 0); }
            else if (ok && tile_live) {
                const uint32_t Pc = N * T_coarse, t_f = S - T_coarse;
                const uint32_t lt = row < Pc ? (row - n * T_coarse) >> 5 : (T_coarse >> 5) + ((row - Pc - n * t_f) >> 5);
                live_bits |= 1u << lt;
            }
        }
        if (ok) {
            g_sigma[row] = gs;
            reinterpret_cast<float4 *>(g_rgbc)[row] = gc;
        }
    }
    if (tile_live) {
        // the ray owns its tiles (T_coarse and S - T_coarse are multiples of 32: checked by the host): one byte each, written once — no atomics
        const uint32_t Pc = N * T_coarse, t_f = S - T_coarse, ntc = T_coarse >> 5, ntf = t_f >> 5;
        for (uint32_t k = 0; k < ntc + ntf; k++) {
            const bool any = __ballot((live_bits >> k) & 1u) != 0;
            if (lane == 0) tile_live[k < ntc ? (n * T_coarse >> 5) + k : ((Pc + n * t_f) >> 5) + (k - ntc)] = any ? 1 : 0;
        }
    }
}

// ------------------------------------------------------------------------------------------------ reconstruction loss
// train_step_pretrain's loss (utils_init_nerf.py:220-234) and its gradient in one launch instead of ~35 elementwise / reduce launches:
//   loss = w_rgb * mean((image - rgb_gt)^2) + w_conf * mean((render_mask - mask_gt)^2)      (image / render_mask of the `all` composite)
// writes d(loss)/d(out_ray) [3][N][6] (zero outside the `all` variant's image and render_mask slots) and per-workgroup partial sums;
// the workgroup that finishes last adds them in a fixed order into loss[0] (a second launch until round 6).
#define RL_BLOCKS 64
__device__ unsigned int g_recon_loss_ticket = 0;     // one k_recon_loss in flight per device (the last workgroup to finish adds the partials and clears it)
__global__ void __launch_bounds__(256) k_recon_loss(const float *__restrict__ out_ray, const float *__restrict__ rgb_gt, const float *__restrict__ mask_gt,
                                                    uint32_t N, float k_rgb, float k_m, float *__restrict__ partial, float *__restrict__ g_out,
                                                    const float *__restrict__ grad_scale, float *__restrict__ loss) {
    __shared__ float red[2][4];
    __shared__ unsigned int s_last;
    float s_rgb = 0.0f, s_m = 0.0f;
    const float gsc = grad_scale ? grad_scale[0] : 1.0f;          // the backward pass's seed (the loss scale), folded into the stored gradient
    for (uint32_t n = blockIdx.x * 256 + threadIdx.x; n < N; n += gridDim.x * 256) {
        const float *r = out_ray + (size_t)n * 6;
        float g[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float d = r[c] - rgb_gt[(size_t)n * 3 + c];
            s_rgb += d * d;
            g[c] = (2.0f * k_rgb * d) * gsc;
        }
        if (mask_gt) {
            const float d = r[5] - mask_gt[n];
            s_m += d * d;
            g[5] = (2.0f * k_m * d) * gsc;
        }
#pragma unroll
        for (int c = 0; c < 6; c++) {
            g_out[(size_t)n * 6 + c] = g[c];
            g_out[((size_t)N + n) * 6 + c] = 0.0f;
            g_out[((size_t)2 * N + n) * 6 + c] = 0.0f;
        }
    }
    s_rgb = rn_wave_sum(s_rgb);
    s_m = rn_wave_sum(s_m);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = s_rgb; red[1][wave] = s_m; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = k_rgb * (red[0][0] + red[0][1] + red[0][2] + red[0][3]) + k_m * (red[1][0] + red[1][1] + red[1][2] + red[1][3]);
        // round 6: the sum of the partials by whichever workgroup finishes last (one launch less), in the fixed lane order of the former second launch (a wave sum over the partials) — same bits.
        // Release the partial (agent scope: the other workgroups run on other XCDs), take a ticket; the last one acquires and adds.
        __threadfence();
        s_last = atomicAdd(&g_recon_loss_ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (s_last && threadIdx.x < 64) {
        __threadfence();
        const float v = rn_wave_sum(threadIdx.x < gridDim.x ? __builtin_nontemporal_load(partial + threadIdx.x) : 0.0f);
        if (threadIdx.x == 0) { loss[0] = v; g_recon_loss_ticket = 0; }
    }
}


extern "C" {

int cnerf_sample_coarse(const float *rays_o, const float *rays_d, const float *nears, const float *fars, const float *aabb, const float *noise,
                        uint32_t N, uint32_t T, float *z_vals, float *xyzs, void *stream) {
    if (T < 2 || T > RN_MAXS) return CNERF_EINVAL;
    if (N == 0) return CNERF_OK;
    if (!rays_o || !rays_d || !nears || !fars || !aabb || !z_vals || !xyzs) return CNERF_ENULL;
    hipLaunchKernelGGL(k_sample_coarse, dim3(cn_div_up(N * T, 256)), dim3(256), 0, CN_STREAM(stream), rays_o, rays_d, nears, fars, aabb, noise, N, T,
                       z_vals, xyzs, (float *)nullptr, 0.0f, (float *)nullptr, (float *)nullptr, 0.0f);
    return cn_launch_status();
}

int cnerf_sample_coarse_unit(const float *rays_o, const float *rays_d, const float *nears, const float *fars, const float *aabb, const float *noise,
                             uint32_t N, uint32_t T, float *z_vals, float *xyzs, float *unit, float bound, void *stream) {
    if (T < 2 || T > RN_MAXS || !(bound > 0.0f)) return CNERF_EINVAL;
    if (N == 0) return CNERF_OK;
    if (!rays_o || !rays_d || !nears || !fars || !aabb || !z_vals || !xyzs || !unit) return CNERF_ENULL;
    hipLaunchKernelGGL(k_sample_coarse, dim3(cn_div_up(N * T, 256)), dim3(256), 0, CN_STREAM(stream), rays_o, rays_d, nears, fars, aabb, noise, N, T,
                       z_vals, xyzs, unit, bound, (float *)nullptr, (float *)nullptr, 0.0f);
    return cn_launch_status();
}

int cnerf_sample_coarse_unit_aabb(const float *rays_o, const float *rays_d, const float *aabb, float min_near, const float *noise, uint32_t N, uint32_t T,
                                  float *nears, float *fars, float *z_vals, float *xyzs, float *unit, float bound, void *stream) {
    if (T < 2 || T > RN_MAXS || !(bound > 0.0f)) return CNERF_EINVAL;
    if (N == 0) return CNERF_OK;
    if (!rays_o || !rays_d || !nears || !fars || !aabb || !z_vals || !xyzs || !unit) return CNERF_ENULL;
    hipLaunchKernelGGL(k_sample_coarse, dim3(cn_div_up(N * T, 256)), dim3(256), 0, CN_STREAM(stream), rays_o, rays_d, (const float *)nullptr,
                       (const float *)nullptr, aabb, noise, N, T, z_vals, xyzs, unit, bound, nears, fars, min_near);
    return cn_launch_status();
}

int cnerf_sample_fine_merge_split(const float *rays_o, const float *rays_d, const float *nears, const float *fars, const float *aabb, const float *z_vals,
                                  const float *sigmas, const float *u, uint32_t N, uint32_t T, uint32_t t, float *z_all, float *xyz_all, float *xyz_fine,
                                  uint32_t *src_index, void *stream) {
    if (T < 3 || T > RN_MAXS || t < 2 || t > RN_MAXS) return CNERF_EINVAL;
    if ((uint64_t)N * (T + t) >= 0xFFFFFFFFull) return CNERF_EINVAL;
    if (N == 0) return CNERF_OK;
    if (!rays_o || !rays_d || !nears || !fars || !aabb || !z_vals || !sigmas || !z_all) return CNERF_ENULL;
    if (!xyz_all && !(xyz_fine && src_index)) return CNERF_ENULL;
    hipLaunchKernelGGL(k_sample_fine_merge, dim3(cn_div_up(N, RN_WAVES)), dim3(RN_THREADS), 0, CN_STREAM(stream), rays_o, rays_d, nears, fars, aabb,
                       z_vals, sigmas, u, N, T, t, z_all, xyz_all, xyz_fine, src_index, (float *)nullptr, 0.0f);
    return cn_launch_status();
}

int cnerf_sample_fine_merge_split_unit(const float *rays_o, const float *rays_d, const float *nears, const float *fars, const float *aabb,
                                       const float *z_vals, const float *sigmas, const float *u, uint32_t N, uint32_t T, uint32_t t, float *z_all,
                                       float *xyz_fine, uint32_t *src_index, float *unit_fine, float bound, void *stream) {
    if (T < 3 || T > RN_MAXS || t < 2 || t > RN_MAXS || !(bound > 0.0f)) return CNERF_EINVAL;
    if ((uint64_t)N * (T + t) >= 0xFFFFFFFFull) return CNERF_EINVAL;
    if (N == 0) return CNERF_OK;
    if (!rays_o || !rays_d || !nears || !fars || !aabb || !z_vals || !sigmas || !z_all || !xyz_fine || !src_index || !unit_fine) return CNERF_ENULL;
    hipLaunchKernelGGL(k_sample_fine_merge, dim3(cn_div_up(N, RN_WAVES)), dim3(RN_THREADS), 0, CN_STREAM(stream), rays_o, rays_d, nears, fars, aabb,
                       z_vals, sigmas, u, N, T, t, z_all, (float *)nullptr, xyz_fine, src_index, unit_fine, bound);
    return cn_launch_status();
}

int cnerf_sample_fine_merge(const float *rays_o, const float *rays_d, const float *nears, const float *fars, const float *aabb, const float *z_vals,
                            const float *sigmas, const float *u, uint32_t N, uint32_t T, uint32_t t, float *z_all, float *xyz_all, void *stream) {
    if (!xyz_all) return CNERF_ENULL;
    return cnerf_sample_fine_merge_split(rays_o, rays_d, nears, fars, aabb, z_vals, sigmas, u, N, T, t, z_all, xyz_all, nullptr, nullptr, stream);
}

int cnerf_composite_run_indexed(const float *sigmas, const float *rgbc, const float *z_vals, const float *nears, const float *fars, uint32_t N, uint32_t S,
                                uint32_t num_steps, int soft_mask, float conf_thr, const uint32_t *src_index, float *out_ray, float *out_weights,
                                float *sigma_sorted, float *rgbc_sorted, void *stream) {
    return cnerf_composite_run_indexed_variants(sigmas, rgbc, z_vals, nears, fars, N, S, num_steps, soft_mask, conf_thr, src_index, out_ray, out_weights,
                                                sigma_sorted, rgbc_sorted, 7u, stream);
}

int cnerf_composite_run_indexed_variants(const float *sigmas, const float *rgbc, const float *z_vals, const float *nears, const float *fars, uint32_t N,
                                         uint32_t S, uint32_t num_steps, int soft_mask, float conf_thr, const uint32_t *src_index, float *out_ray,
                                         float *out_weights, float *sigma_sorted, float *rgbc_sorted, uint32_t variant_mask, void *stream) {
    if (S == 0 || S > 64 * RN_MAXCH || num_steps == 0 || variant_mask == 0 || variant_mask > 7u) return CNERF_EINVAL;
    if (N == 0) return CNERF_OK;
    if (!sigmas || !rgbc || !z_vals || !nears || !fars || !out_ray) return CNERF_ENULL;
    if ((((uintptr_t)rgbc) | ((uintptr_t)rgbc_sorted)) & 15) return CNERF_EINVAL;
    hipLaunchKernelGGL(k_composite_run_fwd, dim3(cn_div_up(N, RN_WAVES)), dim3(RN_THREADS), 0, CN_STREAM(stream), sigmas, rgbc, z_vals, nears, fars, N, S,
                       num_steps, soft_mask, conf_thr, out_ray, out_weights, src_index, sigma_sorted, rgbc_sorted, variant_mask);
    return cn_launch_status();
}

int cnerf_composite_run(const float *sigmas, const float *rgbc, const float *z_vals, const float *nears, const float *fars, uint32_t N, uint32_t S,
                        uint32_t num_steps, int soft_mask, float conf_thr, float *out_ray, float *out_weights, void *stream) {
    return cnerf_composite_run_indexed(sigmas, rgbc, z_vals, nears, fars, N, S, num_steps, soft_mask, conf_thr, nullptr, out_ray, out_weights, nullptr,
                                       nullptr, stream);
}

int cnerf_composite_run_backward_indexed_flush(const float *grad_out_ray, const float *sigmas, const float *rgbc, const float *z_vals, const float *nears,
                                               const float *fars, uint32_t N, uint32_t S, uint32_t num_steps, int soft_mask, float conf_thr, int detach_bg,
                                               int detach_mask_from_field, const uint32_t *src_index, float *grad_sigmas, float *grad_rgbc, int flush_half_zero,
                                               uint8_t *tile_live, void *stream) {
    if (S == 0 || S > 64 * RN_MAXCH || num_steps == 0) return CNERF_EINVAL;
    if (N == 0) return CNERF_OK;
    if (!grad_out_ray || !sigmas || !rgbc || !z_vals || !nears || !fars || !grad_sigmas || !grad_rgbc) return CNERF_ENULL;
    if ((((uintptr_t)rgbc) | ((uintptr_t)grad_rgbc)) & 15) return CNERF_EINVAL;
    // tile flags need the split sample list ([coarse N * num_steps | fine N * (S - num_steps)], src_index) with whole 32-sample tiles per ray
    if (tile_live && (!flush_half_zero || !src_index || num_steps >= S || (num_steps & 31u) || ((S - num_steps) & 31u) || S > 256)) return CNERF_EINVAL;
    if ((uint64_t)N * S > 0xFFFFFFFFull) return CNERF_EINVAL;                        // (row indices are 32-bit, as src_index's)
#define RN_CBWD(NCH)                                                                                                                                  \
    hipLaunchKernelGGL(k_composite_run_bwd<NCH>, dim3(cn_div_up(N, RN_WAVES)), dim3(RN_THREADS), 0, CN_STREAM(stream), grad_out_ray, sigmas, rgbc, z_vals, \
                       nears, fars, N, S, num_steps, soft_mask, conf_thr, detach_bg, detach_mask_from_field, grad_sigmas, grad_rgbc, src_index,     \
                       flush_half_zero ? 1.4901161193847656e-08f /* 2^-26 */ : 0.0f, tile_live, num_steps)
    switch (cn_div_up(S, 64)) {
        case 1: RN_CBWD(1); break;
        case 2: RN_CBWD(2); break;
        case 3: RN_CBWD(3); break;
        default: RN_CBWD(4); break;
    }
#undef RN_CBWD
    return cn_launch_status();
}

int cnerf_composite_run_backward_indexed(const float *grad_out_ray, const float *sigmas, const float *rgbc, const float *z_vals, const float *nears,
                                         const float *fars, uint32_t N, uint32_t S, uint32_t num_steps, int soft_mask, float conf_thr, int detach_bg,
                                         int detach_mask_from_field, const uint32_t *src_index, float *grad_sigmas, float *grad_rgbc, void *stream) {
    return cnerf_composite_run_backward_indexed_flush(grad_out_ray, sigmas, rgbc, z_vals, nears, fars, N, S, num_steps, soft_mask, conf_thr, detach_bg,
                                                      detach_mask_from_field, src_index, grad_sigmas, grad_rgbc, 0, nullptr, stream);
}

int cnerf_composite_run_backward(const float *grad_out_ray, const float *sigmas, const float *rgbc, const float *z_vals, const float *nears,
                                 const float *fars, uint32_t N, uint32_t S, uint32_t num_steps, int soft_mask, float conf_thr, int detach_bg,
                                 int detach_mask_from_field, float *grad_sigmas, float *grad_rgbc, void *stream) {
    return cnerf_composite_run_backward_indexed(grad_out_ray, sigmas, rgbc, z_vals, nears, fars, N, S, num_steps, soft_mask, conf_thr, detach_bg,
                                                detach_mask_from_field, nullptr, grad_sigmas, grad_rgbc, stream);
}

int cnerf_recon_loss_scaled(const float *out_ray, const float *rgb_gt, const float *mask_gt, uint32_t N, float w_rgb, float w_conf,
                            const float *grad_scale, float *loss, float *grad_out_ray, void *stream) {
    if (N == 0) return CNERF_EINVAL;
    if (!out_ray || !rgb_gt || !loss || !grad_out_ray) return CNERF_ENULL;
    // the partial sums live in loss[1 .. RL_BLOCKS] (loss must hold 1 + 64 floats), added in a fixed order by the last workgroup to finish
    const uint32_t blocks = cn_div_up(N, 256) < RL_BLOCKS ? cn_div_up(N, 256) : RL_BLOCKS;
    hipLaunchKernelGGL(k_recon_loss, dim3(blocks), dim3(256), 0, CN_STREAM(stream), out_ray, rgb_gt, mask_gt, N, w_rgb / (3.0f * (float)N),
                       mask_gt ? w_conf / (float)N : 0.0f, loss + 1, grad_out_ray, grad_scale, loss);
    return cn_launch_status();
}

int cnerf_recon_loss(const float *out_ray, const float *rgb_gt, const float *mask_gt, uint32_t N, float w_rgb, float w_conf, float *loss,
                     float *grad_out_ray, void *stream) {
    return cnerf_recon_loss_scaled(out_ray, rgb_gt, mask_gt, N, w_rgb, w_conf, nullptr, loss, grad_out_ray, stream);
}

int cnerf_sample_pdf(const float *bins, const float *weights, const float *u, uint32_t B, uint32_t n_bins, uint32_t n_samples, float *samples, void *stream) {
    if (n_bins < 2 || n_bins > RN_PDF_MAXB) return CNERF_EINVAL;
    if (B == 0 || n_samples == 0) return CNERF_OK;
    if (!bins || !weights || !samples) return CNERF_ENULL;
    hipLaunchKernelGGL(k_sample_pdf, dim3(cn_div_up(B, RN_WAVES)), dim3(RN_THREADS), 0, CN_STREAM(stream), bins, weights, u, B, n_bins, n_samples, samples);
    return cn_launch_status();
}


}  // extern "C"
