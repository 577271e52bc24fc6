// Occupancy-grid refresh of the march path — `NeRFRenderer.update_extra_state` (nerf/renderer.py:1658-1715) as three gfx950 kernels around
// the field's density evaluation.  The reference runs a Python triple loop with boolean-mask indexing, a morton3D launch per chunk, a full-size
// temporary grid and two `.item()` host syncs; here the whole refresh is
//     k_occ_points        cell centres of one cascade + the jitter draw -> query positions (meshgrid order, renderer.py:1679-1695)
//     [density query]     the fused field (gather + MLP kernels), no change
//     k_occ_update        EMA-max into density_grid at the Morton index (:1697, :1707-1709) + block partial sums of the valid cells
//     k_occ_finalize_pack mean density (:1710), threshold min(mean, density_thresh) (:1712), packbits (:1713, raymarching.cu:267-289)
// with the mean and the threshold staying on the device: no host synchronisation.
#include "common.h"

__host__ __device__ __forceinline__ uint32_t oc_expand_bits(uint32_t v) {        // raymarching.cu:56-63
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__host__ __device__ __forceinline__ uint32_t oc_morton3D(uint32_t x, uint32_t y, uint32_t z) {
    return oc_expand_bits(x) | (oc_expand_bits(y) << 1) | (oc_expand_bits(z) << 2);
}

#define OC_BLOCK 256

// cell i of the H^3 grid in custom_meshgrid(X, Y, Z) / reshape(-1) order: x = i / H^2, y = (i / H) % H, z = i % H.
//   xyzs = 2 * coords / (H - 1) - 1                (:1690; torch divides by a host scalar as a multiplication by its float reciprocal)
//   cas_xyzs = xyzs * (bound - half) + (rand * 2 - 1) * half        (:1693-1695)
__global__ void __launch_bounds__(OC_BLOCK) k_occ_points(const float *__restrict__ rand, uint32_t H, float cas_bound, float half_grid, float *__restrict__ xyzs) {
    const uint32_t i = blockIdx.x * OC_BLOCK + threadIdx.x;
    const uint32_t n = H * H * H;
    if (i >= n) return;
    const uint32_t x = i / (H * H), y = (i / H) % H, z = i % H;
    const float inv = 1.0f / (float)(H - 1);
    const float span = cas_bound - half_grid;
    const uint32_t c[3] = {x, y, z};
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const float u = (2.0f * (float)c[d]) * inv - 1.0f;
        float p = u * span;
        p = p + (rand[(size_t)i * 3 + d] * 2.0f - 1.0f) * half_grid;
        xyzs[(size_t)i * 3 + d] = p;
    }
}

// density_grid[cas, morton(cell)] = max(old * decay, sigma) where old >= 0 (:1707-1709); partial[block] = (sum, count) of the valid cells' NEW values
__global__ void __launch_bounds__(OC_BLOCK) k_occ_update(const float *__restrict__ sigmas, uint32_t H, float decay, float *__restrict__ grid_cas,
                                                         double *__restrict__ partial) {
    const uint32_t i = blockIdx.x * OC_BLOCK + threadIdx.x;
    const uint32_t n = H * H * H;
    float v = 0.0f, cnt = 0.0f;
    if (i < n) {
        const uint32_t x = i / (H * H), y = (i / H) % H, z = i % H;
        const uint32_t idx = oc_morton3D(x, y, z);
        const float old = grid_cas[idx];
        if (old >= 0.0f) {
            v = fmaxf(old * decay, sigmas[i]);
            grid_cas[idx] = v;
            cnt = 1.0f;
        }
    }
    // fixed-order block reduction (wave sum on the DPP path, then the four waves through LDS): the refresh is reproducible run to run
    __shared__ double s_sum[OC_BLOCK / 64], s_cnt[OC_BLOCK / 64];
    const float ws = cn_wave_sum(v), wc = cn_wave_sum(cnt);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_sum[wave] = (double)ws; s_cnt[wave] = (double)wc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < OC_BLOCK / 64; w++) { a += s_sum[w]; b += s_cnt[w]; }
        partial[(size_t)blockIdx.x * 2] = a;
        partial[(size_t)blockIdx.x * 2 + 1] = b;
    }
}

// state[0] = mean of the valid cells (all cascades), state[1] = min(mean, density_thresh)
__global__ void __launch_bounds__(OC_BLOCK) k_occ_finalize(const double *__restrict__ partial, uint32_t n_partials, float density_thresh, float *__restrict__ state) {
    __shared__ double s_sum[OC_BLOCK], s_cnt[OC_BLOCK];
    double a = 0.0, b = 0.0;
    for (uint32_t k = threadIdx.x; k < n_partials; k += OC_BLOCK) { a += partial[(size_t)k * 2]; b += partial[(size_t)k * 2 + 1]; }
    s_sum[threadIdx.x] = a; s_cnt[threadIdx.x] = b;
    __syncthreads();
    for (uint32_t s = OC_BLOCK / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) { s_sum[threadIdx.x] += s_sum[threadIdx.x + s]; s_cnt[threadIdx.x] += s_cnt[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float mean = s_cnt[0] > 0.0 ? (float)(s_sum[0] / s_cnt[0]) : 0.0f / 0.0f;     // torch.mean of an empty selection is NaN
        state[0] = mean;
        state[1] = density_thresh < mean ? density_thresh : mean;                           // python's min(mean, thresh), NaN behaviour included
    }
}

// packbits (raymarching.cu:267-289) with the threshold read from device memory (state[1] of k_occ_finalize)
__global__ void __launch_bounds__(OC_BLOCK) k_packbits_dev(const float *__restrict__ grid, uint32_t N, const float *__restrict__ thresh_dev,
                                                           uint8_t *__restrict__ bitfield) {
    const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
    if (n >= N) return;
    const float thresh = *thresh_dev;
    const float4 a = reinterpret_cast<const float4 *>(grid)[(size_t)n * 2];
    const float4 b = reinterpret_cast<const float4 *>(grid)[(size_t)n * 2 + 1];
    uint32_t bits = 0;
    bits |= (a.x > thresh) ? 1u : 0u;
    bits |= (a.y > thresh) ? 2u : 0u;
    bits |= (a.z > thresh) ? 4u : 0u;
    bits |= (a.w > thresh) ? 8u : 0u;
    bits |= (b.x > thresh) ? 16u : 0u;
    bits |= (b.y > thresh) ? 32u : 0u;
    bits |= (b.z > thresh) ? 64u : 0u;
    bits |= (b.w > thresh) ? 128u : 0u;
    bitfield[n] = (uint8_t)bits;
}

extern "C" {

int cnerf_occupancy_points(const float *rand, uint32_t H, float cas_bound, float half_grid, float *xyzs, void *stream) {
    if (H < 2 || H > 1024) return CNERF_EINVAL;
    if (!rand || !xyzs) return CNERF_ENULL;
    const uint32_t n = H * H * H;
    hipLaunchKernelGGL(k_occ_points, dim3(cn_div_up(n, OC_BLOCK)), dim3(OC_BLOCK), 0, CN_STREAM(stream), rand, H, cas_bound, half_grid, xyzs);
    return cn_launch_status();
}

int cnerf_occupancy_update(const float *sigmas, uint32_t H, float decay, float *density_grid_cascade, double *partials, void *stream) {
    if (H < 2 || H > 1024) return CNERF_EINVAL;
    if (!sigmas || !density_grid_cascade || !partials) return CNERF_ENULL;
    if (((uintptr_t)partials) & 7) return CNERF_EINVAL;
    const uint32_t n = H * H * H;
    hipLaunchKernelGGL(k_occ_update, dim3(cn_div_up(n, OC_BLOCK)), dim3(OC_BLOCK), 0, CN_STREAM(stream), sigmas, H, decay, density_grid_cascade, partials);
    return cn_launch_status();
}

int cnerf_occupancy_finalize_pack(const double *partials, uint32_t n_partials, float density_thresh, const float *density_grid, uint32_t n_bytes,
                                  float *state, uint8_t *bitfield, void *stream) {
    if (!partials || !density_grid || !state || !bitfield) return CNERF_ENULL;
    if ((((uintptr_t)density_grid) & 15) || n_partials == 0) return CNERF_EINVAL;
    hipLaunchKernelGGL(k_occ_finalize, dim3(1), dim3(OC_BLOCK), 0, CN_STREAM(stream), partials, n_partials, density_thresh, state);
    if (n_bytes) hipLaunchKernelGGL(k_packbits_dev, dim3(cn_div_up(n_bytes, OC_BLOCK)), dim3(OC_BLOCK), 0, CN_STREAM(stream), density_grid, n_bytes, state + 1, bitfield);
    return cn_launch_status();
}

}  // extern "C"
