/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path.
 *
 * Plain-C, single-thread CPU restatement of the reference's ray-marching
 * CUDA extension (reference: raymarching/src/raymarching.cu).  Each function
 * cites the kernel it follows.  The reference cannot be compiled here (no
 * nvcc / no NVIDIA GPU), so this file IS the executable form of
 * those kernels for parity purposes.
 *
 * Arithmetic conventions (so that the HIP kernels can be bit-exact against
 * this file for all integer / index outputs):
 *   - compiled with -ffp-contract=off; every place where nvcc's default
 *     -fmad=true would contract a*b+c is written as an explicit fmaf().
 *   - the double promotions of the reference are kept: the literal `0.5`
 *     in the voxel-index expression (raymarching.cu:374-376) and in
 *     mip_from_dt (raymarching.cu:50) are doubles there.
 *   - `level * H3 + morton` is a float expression (H3 is float,
 *     raymarching.cu:339,378).
 *   - the atomics-ordered slot reservation (raymarching.cu:405-406) is
 *     nondeterministic in the reference; the canonical order used here is
 *     ray order (what a sequential execution of the kernel produces).
 *   - __expf (fast intrinsic, raymarching.cu:542) is restated as expf();
 *     the floating-point outputs are compared with a tolerance (1e-4 abs,
 *     north_star), not bit-exactly.
 */
#include <math.h>
#include <stdint.h>
#include <float.h>
#include <string.h>

#define SQRT3 1.7320508075688772f
#define RPI 0.3183098861837907f

static inline float signf_(float x) { return copysignf(1.0f, x); }           /* raymarching.cu:30-32 */
static inline float clampf_(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); } /* :34-36 */

/* raymarching.cu:42-47 */
static inline int mip_from_pos(float x, float y, float z, float max_cascade) {
    const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    int exponent;
    frexpf(mx, &exponent);
    return (int)fminf(max_cascade - 1, fmaxf(0, (float)exponent));
}

/* raymarching.cu:49-54 — `dt * H * 0.5` is evaluated in double, then narrowed to float */
static inline int mip_from_dt(float dt, float H, float max_cascade) {
    const float mx = (float)((double)(dt * H) * 0.5);
    int exponent;
    frexpf(mx, &exponent);
    return (int)fminf(max_cascade - 1, fmaxf(0, (float)exponent));
}

/* raymarching.cu:56-63 */
static inline uint32_t expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
/* raymarching.cu:65-71 */
static inline uint32_t morton3D_(uint32_t x, uint32_t y, uint32_t z) {
    return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}
/* raymarching.cu:73-81 */
static inline uint32_t morton3D_invert_(uint32_t x) {
    x = x & 0x49249249;
    x = (x | (x >> 2)) & 0xc30c30c3;
    x = (x | (x >> 4)) & 0x0f00f00f;
    x = (x | (x >> 8)) & 0xff0000ff;
    x = (x | (x >> 16)) & 0x0000ffff;
    return x;
}

/* kernel_near_far_from_aabb, raymarching.cu:91-145 */
void orc_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb,
                            uint32_t N, float min_near, float *nears, float *fars) {
    for (uint32_t n = 0; n < N; n++) {
        const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
        const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
        const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
        float near = (aabb[0] - ox) * rdx, far = (aabb[3] - ox) * rdx;
        if (near > far) { float c = near; near = far; far = c; }
        float near_y = (aabb[1] - oy) * rdy, far_y = (aabb[4] - oy) * rdy;
        if (near_y > far_y) { float c = near_y; near_y = far_y; far_y = c; }
        if (near > far_y || near_y > far) { nears[n] = fars[n] = FLT_MAX; continue; }
        if (near_y > near) near = near_y;
        if (far_y < far) far = far_y;
        float near_z = (aabb[2] - oz) * rdz, far_z = (aabb[5] - oz) * rdz;
        if (near_z > far_z) { float c = near_z; near_z = far_z; far_z = c; }
        if (near > far_z || near_z > far) { nears[n] = fars[n] = FLT_MAX; continue; }
        if (near_z > near) near = near_z;
        if (far_z < far) far = far_z;
        if (near < min_near) near = min_near;
        nears[n] = near;
        fars[n] = far;
    }
}

/* kernel_sph_from_ray, raymarching.cu:162-198 */
void orc_sph_from_ray(const float *rays_o, const float *rays_d, float radius, uint32_t N, float *coords) {
    for (uint32_t n = 0; n < N; n++) {
        const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
        const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
        const float A = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
        const float B = fmaf(oz, dz, fmaf(oy, dy, ox * dx));
        const float C = fmaf(-radius, radius, fmaf(oz, oz, fmaf(oy, oy, ox * ox)));
        const float t = (-B + sqrtf(fmaf(B, B, -(A * C)))) / A;
        const float x = fmaf(t, dx, ox), y = fmaf(t, dy, oy), z = fmaf(t, dz, oz);
        const float theta = atan2f(sqrtf(fmaf(z, z, x * x)), y);
        const float phi = atan2f(z, x);
        coords[n * 2] = fmaf(2 * theta, RPI, -1.0f);
        coords[n * 2 + 1] = phi * RPI;
    }
}

/* kernel_morton3D, raymarching.cu:214-226 */
void orc_morton3D(const int *coords, uint32_t N, int *indices) {
    for (uint32_t n = 0; n < N; n++)
        indices[n] = (int)morton3D_((uint32_t)coords[n * 3], (uint32_t)coords[n * 3 + 1], (uint32_t)coords[n * 3 + 2]);
}

/* kernel_morton3D_invert, raymarching.cu:237-254 */
void orc_morton3D_invert(const int *indices, uint32_t N, int *coords) {
    for (uint32_t n = 0; n < N; n++) {
        const int ind = indices[n];
        coords[n * 3] = (int)morton3D_invert_((uint32_t)(ind >> 0));
        coords[n * 3 + 1] = (int)morton3D_invert_((uint32_t)(ind >> 1));
        coords[n * 3 + 2] = (int)morton3D_invert_((uint32_t)(ind >> 2));
    }
}

/* kernel_packbits, raymarching.cu:267-289 */
void orc_packbits(const float *grid, uint32_t N, float density_thresh, uint8_t *bitfield) {
    for (uint32_t n = 0; n < N; n++) {
        uint8_t bits = 0;
        for (int i = 0; i < 8; i++) bits |= (grid[n * 8 + i] > density_thresh) ? ((uint8_t)1 << i) : 0;
        bitfield[n] = bits;
    }
}

/* One marching iteration shared by the three march loops
 * (raymarching.cu:359-400, 427-479, 934-988).  Returns occupancy and fills the
 * clamped sample position / step / voxel; on a miss advances *t to the next
 * voxel boundary exactly as the reference's do-while does. */
typedef struct { float x, y, z, dt; int occ; } march_step_t;

static inline march_step_t march_probe(float ox, float oy, float oz, float dx, float dy, float dz,
                                       float rdx, float rdy, float rdz, float *t,
                                       float bound, float dt_gamma, float dt_min, float dt_max,
                                       uint32_t C, uint32_t H, float rH, float H3, const uint8_t *grid) {
    march_step_t s;
    s.x = clampf_(fmaf(*t, dx, ox), -bound, bound);
    s.y = clampf_(fmaf(*t, dy, oy), -bound, bound);
    s.z = clampf_(fmaf(*t, dz, oz), -bound, bound);
    s.dt = clampf_(*t * dt_gamma, dt_min, dt_max);

    const int lp = mip_from_pos(s.x, s.y, s.z, (float)C);
    const int ld = mip_from_dt(s.dt, (float)H, (float)C);
    const int level = lp > ld ? lp : ld;

    const float mip_bound = fminf(scalbnf(1.0f, level), bound);
    const float mip_rbound = 1 / mip_bound;

    /* 0.5 * (x * mip_rbound + 1) * H : float fma, then double, then narrowed to float by clamp()'s
     * float parameter, then truncated to int (raymarching.cu:374-376) */
    const float Hm1 = (float)(H - 1);
    const int nx = (int)clampf_((float)(0.5 * (double)fmaf(s.x, mip_rbound, 1.0f) * (double)H), 0.0f, Hm1);
    const int ny = (int)clampf_((float)(0.5 * (double)fmaf(s.y, mip_rbound, 1.0f) * (double)H), 0.0f, Hm1);
    const int nz = (int)clampf_((float)(0.5 * (double)fmaf(s.z, mip_rbound, 1.0f) * (double)H), 0.0f, Hm1);

    const uint32_t index = (uint32_t)fmaf((float)level, H3, (float)morton3D_((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
    s.occ = (grid[index / 8] & (1 << (index % 8))) != 0;

    if (!s.occ) {
        const float tx = (fmaf(fmaf((nx + 0.5f + 0.5f * signf_(dx)) * rH, 2.0f, -1.0f), mip_bound, -s.x)) * rdx;
        const float ty = (fmaf(fmaf((ny + 0.5f + 0.5f * signf_(dy)) * rH, 2.0f, -1.0f), mip_bound, -s.y)) * rdy;
        const float tz = (fmaf(fmaf((nz + 0.5f + 0.5f * signf_(dz)) * rH, 2.0f, -1.0f), mip_bound, -s.z)) * rdz;
        const float tt = *t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
        do {
            *t += clampf_(*t * dt_gamma, dt_min, dt_max);
        } while (*t < tt);
    }
    return s;
}

/* kernel_march_rays_train, raymarching.cu:311-480.  counter[0] += samples, counter[1] += rays,
 * slots reserved in ray order (canonical form of the reference's atomics order). */
void orc_march_rays_train(const float *rays_o, const float *rays_d, const uint8_t *grid,
                          float bound, float dt_gamma, uint32_t max_steps,
                          uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                          const float *nears, const float *fars,
                          float *xyzs, float *dirs, float *deltas, int *rays, int *counter,
                          const float *noises) {
    const float rH = 1 / (float)H;
    const float H3 = (float)(H * H * H);
    const float dt_min = 2 * SQRT3 / max_steps;
    const float dt_max = 2 * SQRT3 * (1 << (C - 1)) / H;
    for (uint32_t n = 0; n < N; n++) {
        const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
        const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
        const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
        const float near = nears[n], far = fars[n], noise = noises[n];
        (void)near;
        float t0 = nears[n];
        t0 = fmaf(clampf_(t0 * dt_gamma, dt_min, dt_max), noise, t0);

        /* first pass */
        float t = t0;
        uint32_t num_steps = 0;
        while (t < far && num_steps < max_steps) {
            march_step_t s = march_probe(ox, oy, oz, dx, dy, dz, rdx, rdy, rdz, &t, bound, dt_gamma, dt_min, dt_max, C, H, rH, H3, grid);
            if (s.occ) { num_steps++; t += s.dt; }
        }

        const uint32_t point_index = (uint32_t)counter[0]; counter[0] += (int)num_steps;
        const uint32_t ray_index = (uint32_t)counter[1]; counter[1] += 1;
        rays[ray_index * 3] = (int)n;
        rays[ray_index * 3 + 1] = (int)point_index;
        rays[ray_index * 3 + 2] = (int)num_steps;
        if (num_steps == 0) continue;
        if (point_index + num_steps > M) continue;

        float *px = xyzs + (size_t)point_index * 3, *pd = dirs + (size_t)point_index * 3, *pl = deltas + (size_t)point_index * 2;
        t = t0;
        uint32_t step = 0;
        float last_t = t;
        while (t < far && step < num_steps) {
            march_step_t s = march_probe(ox, oy, oz, dx, dy, dz, rdx, rdy, rdz, &t, bound, dt_gamma, dt_min, dt_max, C, H, rH, H3, grid);
            if (s.occ) {
                px[0] = s.x; px[1] = s.y; px[2] = s.z;
                pd[0] = dx; pd[1] = dy; pd[2] = dz;
                t += s.dt;
                pl[0] = s.dt;
                pl[1] = t - last_t;
                last_t = t;
                px += 3; pd += 3; pl += 2;
                step++;
            }
        }
    }
}

/* kernel_composite_rays_train_forward, raymarching.cu:500-577 (the _sdf twin :579-657 is identical) */
void orc_composite_rays_train_forward(const float *sigmas, const float *rgbs, const float *deltas, const int *rays,
                                      uint32_t M, uint32_t N, float T_thresh,
                                      float *weights_sum, float *depth, float *image) {
    for (uint32_t n = 0; n < N; n++) {
        const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
        if (num_steps == 0 || offset + num_steps > M) {
            weights_sum[index] = 0; depth[index] = 0;
            image[index * 3] = image[index * 3 + 1] = image[index * 3 + 2] = 0;
            continue;
        }
        const float *ps = sigmas + offset, *pc = rgbs + (size_t)offset * 3, *pl = deltas + (size_t)offset * 2;
        uint32_t step = 0;
        float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, t = 0, d = 0;
        while (step < num_steps) {
            const float alpha = 1.0f - expf(-ps[0] * pl[0]);
            const float weight = alpha * T;
            r = fmaf(weight, pc[0], r); g = fmaf(weight, pc[1], g); b = fmaf(weight, pc[2], b);
            t += pl[1];
            d = fmaf(weight, t, d);
            ws += weight;
            T *= 1.0f - alpha;
            if (T < T_thresh) break;
            ps++; pc += 3; pl += 2; step++;
        }
        weights_sum[index] = ws; depth[index] = d;
        image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
    }
}

/* kernel_composite_rays_train_backward, raymarching.cu:691-772 (the _sdf twin :776-857 is identical).
 * grad_sigmas / grad_rgbs must be zero-initialised by the caller (raymarching.py:284-285). */
void orc_composite_rays_train_backward(const float *grad_weights_sum, const float *grad_image,
                                       const float *sigmas, const float *rgbs, const float *deltas, const int *rays,
                                       const float *weights_sum, const float *image,
                                       uint32_t M, uint32_t N, float T_thresh,
                                       float *grad_sigmas, float *grad_rgbs) {
    for (uint32_t n = 0; n < N; n++) {
        const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
        if (num_steps == 0 || offset + num_steps > M) continue;
        const float gws = grad_weights_sum[index];
        const float *gi = grad_image + (size_t)index * 3;
        const float r_final = image[index * 3], g_final = image[index * 3 + 1], b_final = image[index * 3 + 2];
        const float ws_final = weights_sum[index];
        const float *ps = sigmas + offset, *pc = rgbs + (size_t)offset * 3, *pl = deltas + (size_t)offset * 2;
        float *gs = grad_sigmas + offset, *gc = grad_rgbs + (size_t)offset * 3;
        uint32_t step = 0;
        float T = 1.0f, r = 0, g = 0, b = 0, ws = 0;
        while (step < num_steps) {
            const float alpha = 1.0f - expf(-ps[0] * pl[0]);
            const float weight = alpha * T;
            r = fmaf(weight, pc[0], r); g = fmaf(weight, pc[1], g); b = fmaf(weight, pc[2], b);
            ws += weight;
            T *= 1.0f - alpha;
            gc[0] = gi[0] * weight; gc[1] = gi[1] * weight; gc[2] = gi[2] * weight;
            gs[0] = pl[0] * (
                gi[0] * (fmaf(T, pc[0], -(r_final - r))) +
                gi[1] * (fmaf(T, pc[1], -(g_final - g))) +
                gi[2] * (fmaf(T, pc[2], -(b_final - b))) +
                gws * (1 - ws_final));
            if (T < T_thresh) break;
            ps++; pc += 3; pl += 2; gs++; gc += 3; step++;
        }
    }
}

/* kernel_march_rays, raymarching.cu:884-989 */
void orc_march_rays(uint32_t n_alive, uint32_t n_step, const int *rays_alive, const float *rays_t,
                    const float *rays_o, const float *rays_d, float bound, float dt_gamma, uint32_t max_steps,
                    uint32_t C, uint32_t H, const uint8_t *grid, const float *nears, const float *fars,
                    float *xyzs, float *dirs, float *deltas, const float *noises) {
    const float rH = 1 / (float)H;
    const float H3 = (float)(H * H * H);
    const float dt_min = 2 * SQRT3 / max_steps;
    const float dt_max = 2 * SQRT3 * (1 << (C - 1)) / H;
    for (uint32_t n = 0; n < n_alive; n++) {
        const int index = rays_alive[n];
        const float noise = noises[n];
        const float *o = rays_o + (size_t)index * 3, *dd = rays_d + (size_t)index * 3;
        float *px = xyzs + (size_t)n * n_step * 3, *pd = dirs + (size_t)n * n_step * 3, *pl = deltas + (size_t)n * n_step * 2;
        const float ox = o[0], oy = o[1], oz = o[2];
        const float dx = dd[0], dy = dd[1], dz = dd[2];
        const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
        float t = rays_t[index];
        const float far = fars[index];
        (void)nears;
        uint32_t step = 0;
        t = fmaf(clampf_(t * dt_gamma, dt_min, dt_max), noise, t);
        float last_t = t;
        while (t < far && step < n_step) {
            march_step_t s = march_probe(ox, oy, oz, dx, dy, dz, rdx, rdy, rdz, &t, bound, dt_gamma, dt_min, dt_max, C, H, rH, H3, grid);
            if (s.occ) {
                px[0] = s.x; px[1] = s.y; px[2] = s.z;
                pd[0] = dx; pd[1] = dy; pd[2] = dz;
                t += s.dt;
                pl[0] = s.dt;
                pl[1] = t - last_t;
                last_t = t;
                px += 3; pd += 3; pl += 2;
                step++;
            }
        }
    }
}

/* kernel_composite_rays, raymarching.cu:1002-1089 */
void orc_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int *rays_alive, float *rays_t,
                        const float *sigmas, const float *rgbs, const float *deltas,
                        float *weights_sum, float *depth, float *image) {
    for (uint32_t n = 0; n < n_alive; n++) {
        const int index = rays_alive[n];
        const float *ps = sigmas + (size_t)n * n_step, *pc = rgbs + (size_t)n * n_step * 3, *pl = deltas + (size_t)n * n_step * 2;
        float t = rays_t[index];
        float weight_sum = weights_sum[index], d = depth[index];
        float r = image[index * 3], g = image[index * 3 + 1], b = image[index * 3 + 2];
        uint32_t step = 0;
        while (step < n_step) {
            if (pl[0] == 0) break;
            const float alpha = 1.0f - expf(-ps[0] * pl[0]);
            const float T = 1 - weight_sum;
            const float weight = alpha * T;
            weight_sum += weight;
            t += pl[1];
            d = fmaf(weight, t, d);
            r = fmaf(weight, pc[0], r); g = fmaf(weight, pc[1], g); b = fmaf(weight, pc[2], b);
            if (T < T_thresh) break;
            ps++; pc += 3; pl += 2; step++;
        }
        if (step < n_step) rays_alive[n] = -1; else rays_t[index] = t;
        weights_sum[index] = weight_sum; depth[index] = d;
        image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
    }
}
