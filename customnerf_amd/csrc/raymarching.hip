// Ray-marching kernels for gfx950 (CDNA4, wave64).
//
// Replaces the reference's `_raymarching` extension (raymarching/src/raymarching.cu) behind the C-ABI of
// include/customnerf_hip.h.  Written from the kernels' documented behaviour, not translated: slot reservation
// is a deterministic ray-ordered scan (count -> scan -> write) instead of two global atomics per ray, the alive
// list is compacted on the device with wave ballots, and float arithmetic that decides integer outputs
// (num_steps, voxel indices) is spelled out operation by operation so it is bit-identical to
// oracle/raymarching_ref.c (compiled -ffp-contract=off on both sides).
#include "common.h"
#include <float.h>

#define RM_BLOCK 256

__device__ __forceinline__ float rm_signf(float x) { return copysignf(1.0f, x); }

__device__ __forceinline__ int rm_mip_from_pos(float x, float y, float z, float max_cascade) {
    const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    int exponent;
    frexpf(mx, &exponent);
    return (int)fminf(max_cascade - 1, fmaxf(0.0f, (float)exponent));
}

__device__ __forceinline__ int rm_mip_from_dt(float dt, float H, float max_cascade) {
    const float mx = (float)((double)(dt * H) * 0.5);
    int exponent;
    frexpf(mx, &exponent);
    return (int)fminf(max_cascade - 1, fmaxf(0.0f, (float)exponent));
}

__host__ __device__ __forceinline__ uint32_t rm_expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__host__ __device__ __forceinline__ uint32_t rm_morton3D(uint32_t x, uint32_t y, uint32_t z) {
    return rm_expand_bits(x) | (rm_expand_bits(y) << 1) | (rm_expand_bits(z) << 2);
}
__host__ __device__ __forceinline__ uint32_t rm_morton3D_invert(uint32_t x) {
    x = x & 0x49249249;
    x = (x | (x >> 2)) & 0xc30c30c3;
    x = (x | (x >> 4)) & 0x0f00f00f;
    x = (x | (x >> 8)) & 0xff0000ff;
    x = (x | (x >> 16)) & 0x0000ffff;
    return x;
}

// ------------------------------------------------------------------------------------------------ utils
__global__ void __launch_bounds__(RM_BLOCK) k_near_far_from_aabb(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                                 const float *__restrict__ aabb, uint32_t N, float min_near,
                                                                 float *__restrict__ nears, float *__restrict__ fars) {
    const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
    if (n >= N) return;
    float near, far;
    cn_near_far(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, aabb, min_near, near, far);
    nears[n] = near;
    fars[n] = far;
}

__global__ void __launch_bounds__(RM_BLOCK) k_sph_from_ray(const float *__restrict__ rays_o, const float *__restrict__ rays_d, float radius,
                                                           uint32_t N, float *__restrict__ coords) {
    const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
    if (n >= N) return;
    const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
    const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
    const float A = cn_fma(dz, dz, cn_fma(dy, dy, dx * dx));
    const float B = cn_fma(oz, dz, cn_fma(oy, dy, ox * dx));
    const float C = cn_fma(-radius, radius, cn_fma(oz, oz, cn_fma(oy, oy, ox * ox)));
    const float t = (-B + sqrtf(cn_fma(B, B, -(A * C)))) / A;
    const float x = cn_fma(t, dx, ox), y = cn_fma(t, dy, oy), z = cn_fma(t, dz, oz);
    const float theta = atan2f(sqrtf(cn_fma(z, z, x * x)), y);
    const float phi = atan2f(z, x);
    coords[n * 2] = cn_fma(2 * theta, 0.3183098861837907f, -1.0f);
    coords[n * 2 + 1] = phi * 0.3183098861837907f;
}

__global__ void __launch_bounds__(RM_BLOCK) k_morton3D(const int *__restrict__ coords, uint32_t N, int *__restrict__ indices) {
    const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
    if (n >= N) return;
    indices[n] = (int)rm_morton3D((uint32_t)coords[n * 3], (uint32_t)coords[n * 3 + 1], (uint32_t)coords[n * 3 + 2]);
}

__global__ void __launch_bounds__(RM_BLOCK) k_morton3D_invert(const int *__restrict__ indices, uint32_t N, int *__restrict__ coords) {
    const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
    if (n >= N) return;
    const int ind = indices[n];
    coords[n * 3] = (int)rm_morton3D_invert((uint32_t)(ind >> 0));
    coords[n * 3 + 1] = (int)rm_morton3D_invert((uint32_t)(ind >> 1));
    coords[n * 3 + 2] = (int)rm_morton3D_invert((uint32_t)(ind >> 2));
}

// one thread packs 8 consecutive cells = two float4 loads -> one byte
__global__ void __launch_bounds__(RM_BLOCK) k_packbits(const float *__restrict__ grid, uint32_t N, float thresh, uint8_t *__restrict__ bitfield) {
    const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
    if (n >= N) return;
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

// ------------------------------------------------------------------------------------------------ marching core
// DPP lane exchange: `old` is what a lane keeps when its source lane does not exist
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float rm_dpp(float old, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL, ROW_MASK, 0xF, false));
}
struct RayState {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz;
};
struct MarchConst {
    float bound, dt_gamma, dt_min, dt_max, rH, H3, Hf, Cf, Hm1;
    uint32_t H;
};
struct Probe {
    float x, y, z, dt;
    bool occ;
};

__device__ __forceinline__ MarchConst rm_consts(float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H) {
    MarchConst k;
    k.bound = bound;
    k.dt_gamma = dt_gamma;
    k.dt_min = 2 * 1.7320508075688772f / max_steps;
    k.dt_max = 2 * 1.7320508075688772f * (1 << (C - 1)) / H;
    k.rH = 1 / (float)H;
    k.H3 = (float)(H * H * H);
    k.Hf = (float)H;
    k.Cf = (float)C;
    k.Hm1 = (float)(H - 1);
    k.H = H;
    return k;
}

// One probe of the occupancy bitfield at parameter t.  On a miss `tt` is the parameter at which the ray leaves the empty voxel
// (the DDA-like skip target of raymarching.cu:452-461); t itself is not touched.
__device__ __forceinline__ Probe rm_probe_at(const RayState &r, const float t, const MarchConst &k, const uint8_t *__restrict__ grid, float &tt) {
    Probe s;
    s.x = cn_clamp(cn_fma(t, r.dx, r.ox), -k.bound, k.bound);
    s.y = cn_clamp(cn_fma(t, r.dy, r.oy), -k.bound, k.bound);
    s.z = cn_clamp(cn_fma(t, r.dz, r.oz), -k.bound, k.bound);
    s.dt = cn_clamp(t * k.dt_gamma, k.dt_min, k.dt_max);

    const int lp = rm_mip_from_pos(s.x, s.y, s.z, k.Cf);
    const int ld = rm_mip_from_dt(s.dt, k.Hf, k.Cf);
    const int level = lp > ld ? lp : ld;
    const float mip_bound = fminf(scalbnf(1.0f, level), k.bound);
    const float mip_rbound = 1 / mip_bound;

    const int nx = (int)cn_clamp((float)(0.5 * (double)cn_fma(s.x, mip_rbound, 1.0f) * (double)k.H), 0.0f, k.Hm1);
    const int ny = (int)cn_clamp((float)(0.5 * (double)cn_fma(s.y, mip_rbound, 1.0f) * (double)k.H), 0.0f, k.Hm1);
    const int nz = (int)cn_clamp((float)(0.5 * (double)cn_fma(s.z, mip_rbound, 1.0f) * (double)k.H), 0.0f, k.Hm1);

    const uint32_t index = (uint32_t)cn_fma((float)level, k.H3, (float)rm_morton3D((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
    s.occ = (grid[index >> 3] & (1u << (index & 7u))) != 0;

    const float tx = cn_fma(cn_fma((nx + 0.5f + 0.5f * rm_signf(r.dx)) * k.rH, 2.0f, -1.0f), mip_bound, -s.x) * r.rdx;
    const float ty = cn_fma(cn_fma((ny + 0.5f + 0.5f * rm_signf(r.dy)) * k.rH, 2.0f, -1.0f), mip_bound, -s.y) * r.rdy;
    const float tz = cn_fma(cn_fma((nz + 0.5f + 0.5f * rm_signf(r.dz)) * k.rH, 2.0f, -1.0f), mip_bound, -s.z) * r.rdz;
    tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    return s;
}

// One probe at t; on a miss t is advanced past the voxel.
__device__ __forceinline__ Probe rm_probe(const RayState &r, float &t, const MarchConst &k, const uint8_t *__restrict__ grid) {
    float tt;
    const Probe s = rm_probe_at(r, t, k, grid, tt);
    if (!s.occ) {
        do {
            t += cn_clamp(t * k.dt_gamma, k.dt_min, k.dt_max);
        } while (t < tt);
    }
    return s;
}

__device__ __forceinline__ RayState rm_load_ray(const float *__restrict__ rays_o, const float *__restrict__ rays_d, uint32_t n) {
    RayState r;
    r.ox = rays_o[n * 3]; r.oy = rays_o[n * 3 + 1]; r.oz = rays_o[n * 3 + 2];
    r.dx = rays_d[n * 3]; r.dy = rays_d[n * 3 + 1]; r.dz = rays_d[n * 3 + 2];
    r.rdx = 1 / r.dx; r.rdy = 1 / r.dy; r.rdz = 1 / r.dz;
    return r;
}

// pass 1: count occupied steps per ray; rays[n] = (n, <unset>, num_steps); optionally record (t, dt) of every occupied probe.
// One WAVE per ray (the reference's thread-per-ray loop, raymarching.cu:311-480, measured here at 0.59 ms per 128x128 view: 256 waves of
// divergent serial chains; a four-probe look-ahead inside that loop was bit-identical but slower, 0.79 ms).  Whatever the occupancy grid says, the serial march only ever visits points of the recurrence
// t_{k+1} = t_k + dt(t_k): one step after a hit, as many as it takes to reach the empty voxel's exit tt after a miss.  So: the 64 lanes
// hold 64 consecutive points of the recurrence (built by a lane-to-lane DPP chain: every value is the same sequence of float additions the
// serial loop performs), all 64 are probed at once, and the serial decisions are then replayed on the ballot masks — "probe the first
// point with t >= skip target; hit: take the whole run of hits behind it; miss: new skip target" — a few scalar iterations per chunk
// instead of one occupancy-grid round trip per probe on a single wave per CU.  Same accepted points and counts, bit for bit.
__global__ void __launch_bounds__(256) k_march_train_count_wave(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                                const uint8_t *__restrict__ grid, float bound, float dt_gamma,
                                                                uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                                                                const float *__restrict__ nears, const float *__restrict__ fars,
                                                                int *__restrict__ rays, const float *__restrict__ noises,
                                                                float2 *__restrict__ hits) {
    const uint32_t lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;                                                   // whole waves leave together
    const MarchConst k = rm_consts(bound, dt_gamma, max_steps, C, H);
    const RayState r = rm_load_ray(rays_o, rays_d, n);
    const float far = fars[n];
    float t_base = nears[n];
    t_base = cn_fma(cn_clamp(t_base * dt_gamma, k.dt_min, k.dt_max), noises[n], t_base);
    uint32_t num_steps = 0;
    float T_skip = -INFINITY;                                            // probe a point iff t >= T_skip
    float2 *__restrict__ hl = hits ? hits + (size_t)n * max_steps : nullptr;
    bool done = !(t_base < far);
    while (!done) {
        // the chunk's 64 points: lane l = t_base advanced l times
        float t = t_base;
        for (int i = 0; i < 63; i++) {
            const float prev = rm_dpp<0x138>(t, t);                     // lane l-1's value (lane 0 keeps its own)
            const float step = lane == 0 ? 0.0f : cn_clamp(prev * k.dt_gamma, k.dt_min, k.dt_max);
            t = prev + step;                                             // lanes <= i are final and recompute themselves
        }
        float tt;
        const Probe s = rm_probe_at(r, t, k, grid, tt);
        const uint64_t occm = __ballot(s.occ), validm = __ballot(t < far);
        uint64_t acc = 0;
        uint32_t cur = 0;
        while (cur < 64) {
            const uint64_t cand = __ballot(t >= T_skip) & (~0ull << cur);
            if (cand == 0) break;                                        // the rest of the chunk lies inside the skipped voxel
            const uint32_t l = (uint32_t)__builtin_ctzll(cand);
            if (!((validm >> l) & 1ull) || num_steps >= max_steps) { done = true; break; }      // the serial loop's condition fails at this probe
            if ((occm >> l) & 1ull) {
                // a hit is followed by the very next point: take the whole run of hits (while the loop condition holds)
                const uint64_t ok = (occm & validm) >> l;
                uint32_t len = ~ok == 0 ? 64u : (uint32_t)__builtin_ctzll(~ok);
                len = min(len, 64u - l);
                len = min(len, max_steps - num_steps);
                acc |= (len >= 64 ? ~0ull : ((1ull << len) - 1ull)) << l;
                num_steps += len;
                cur = l + len;
                T_skip = -INFINITY;
            } else {
                T_skip = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tt), (int)l));
                cur = l + 1;
            }
        }
        if (hl && acc) {
            const uint32_t before = num_steps - (uint32_t)__builtin_popcountll(acc);
            if ((acc >> lane) & 1ull) {
                const uint32_t rank = (uint32_t)__builtin_popcountll(acc & ((1ull << lane) - 1ull));
                hl[before + rank] = make_float2(t, s.dt);
            }
        }
        if (done) break;
        // next chunk starts one step behind lane 63
        const float t63 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t), 63));
        t_base = t63 + cn_clamp(t63 * k.dt_gamma, k.dt_min, k.dt_max);
        if (!(t_base < far)) done = true;
    }
    if (lane == 0) {
        rays[n * 3] = (int)n;
        rays[n * 3 + 2] = (int)num_steps;
    }
}

// pass 2: exclusive scan of num_steps in ray order (one workgroup; wave shuffles + one LDS hop).
// rays[n*3+1] = base + sum_{m<n} num_steps[m];  counter[0] += total, counter[1] += N.
// Fixed-budget callers (M < 2^32 - 1): when the call overflows its budget (base + total > M) the rays whose segments end beyond M are
// dropped (raymarching.cu:416).  With a scan that always starts at ray 0 those would always be the highest-numbered rays — the bottom
// rows of a whole-view batch — whereas the reference drops rays in atomicAdd arrival order, unrelated to image position.  On overflow
// the scan therefore starts at ray rot = floor(noises[0] * N) and wraps around: which rays are dropped moves with the per-call jitter
// draw (and stays a deterministic function of the inputs).  Calls that fit their budget keep the plain ray order.
#define SCAN_THREADS 1024
__global__ void __launch_bounds__(SCAN_THREADS) k_march_train_scan(int *__restrict__ rays, int *__restrict__ counter, uint32_t N, uint32_t M,
                                                                   const float *__restrict__ noises) {
    __shared__ uint32_t wave_tot[SCAN_THREADS / CN_WAVE];
    __shared__ uint32_t prefix_rot;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t chunk = (N + SCAN_THREADS - 1) / SCAN_THREADS;
    const uint32_t begin = min(tid * chunk, N), end = min(begin + chunk, N);
    uint32_t local = 0;
    for (uint32_t n = begin; n < end; n++) local += (uint32_t)rays[n * 3 + 2];
    const uint32_t incl = cn_wave_incl_scan(local);
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t wave_base = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < SCAN_THREADS / CN_WAVE; w++) {
        const uint32_t v = wave_tot[w];
        if (w < wave) wave_base += v;
        total += v;
    }
    const uint32_t base = (uint32_t)counter[0];
    const uint32_t first = wave_base + incl - local;                      // samples of the rays before this thread's chunk
    uint32_t rot = 0;
    if (M != 0xFFFFFFFFu && (uint64_t)base + total > M) rot = min((uint32_t)(noises[0] * (float)N), N - 1);   // workgroup-uniform
    if (rot) {
        if (rot >= begin && rot < end) {
            uint32_t run = first;
            for (uint32_t n = begin; n < rot; n++) run += (uint32_t)rays[n * 3 + 2];
            prefix_rot = run;
        }
        __syncthreads();
    }
    const uint32_t pr = rot ? prefix_rot : 0;
    uint32_t run = first;
    for (uint32_t n = begin; n < end; n++) {
        rays[n * 3 + 1] = (int)(base + (n >= rot ? run - pr : total - pr + run));
        run += (uint32_t)rays[n * 3 + 2];
    }
    __syncthreads();   // every thread has read counter[0] before it is updated
    if (tid == 0) {
        counter[0] = (int)(base + total);
        counter[1] += (int)N;
    }
}

// pass 3: re-march and write the samples of ray n at its reserved segment
__global__ void __launch_bounds__(RM_BLOCK) k_march_train_write(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                                const uint8_t *__restrict__ grid, float bound, float dt_gamma,
                                                                uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                                                                const float *__restrict__ nears, const float *__restrict__ fars,
                                                                float *__restrict__ xyzs, float *__restrict__ dirs, float *__restrict__ deltas,
                                                                const int *__restrict__ rays, const float *__restrict__ noises) {
    const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
    if (n >= N) return;
    const uint32_t point_index = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
    if (num_steps == 0 || point_index + num_steps > M) return;
    const MarchConst k = rm_consts(bound, dt_gamma, max_steps, C, H);
    const RayState r = rm_load_ray(rays_o, rays_d, n);
    const float far = fars[n];
    float t = nears[n];
    t = cn_fma(cn_clamp(t * dt_gamma, k.dt_min, k.dt_max), noises[n], t);
    float *px = xyzs + (size_t)point_index * 3, *pd = dirs + (size_t)point_index * 3, *pl = deltas + (size_t)point_index * 2;
    uint32_t step = 0;
    float last_t = t;
    while (t < far && step < num_steps) {
        const Probe s = rm_probe(r, t, k, grid);
        if (s.occ) {
            px[0] = s.x; px[1] = s.y; px[2] = s.z;
            pd[0] = r.dx; pd[1] = r.dy; pd[2] = r.dz;
            t += s.dt;
            pl[0] = s.dt;
            pl[1] = t - last_t;
            last_t = t;
            px += 3; pd += 3; pl += 2;
            step++;
        }
    }
}

// pass 3, from the hit list of pass 1 (no second march): one wave per ray, its samples on the lanes.  Same arithmetic as the serial
// writer: position = clamp(o + t d), deltas = (dt, (t + dt) - previous (t + dt)), the first sample measured from the jittered start.
__global__ void __launch_bounds__(256) k_march_train_write_hits(const float *__restrict__ rays_o, const float *__restrict__ rays_d, float bound,
                                                                float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                                                                const float *__restrict__ nears, const float *__restrict__ noises,
                                                                const float2 *__restrict__ hits, const int *__restrict__ rays,
                                                                float *__restrict__ xyzs, float *__restrict__ dirs, float *__restrict__ deltas) {
    const uint32_t lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const uint32_t point_index = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
    if (num_steps == 0 || point_index + num_steps > M) return;
    const MarchConst k = rm_consts(bound, dt_gamma, max_steps, C, H);
    const RayState r = rm_load_ray(rays_o, rays_d, n);
    float t0 = nears[n];
    t0 = cn_fma(cn_clamp(t0 * dt_gamma, k.dt_min, k.dt_max), noises[n], t0);       // the jittered start of the march kernels
    const float2 *__restrict__ hl = hits + (size_t)n * max_steps;
    for (uint32_t base = 0; base < num_steps; base += 64) {
        const uint32_t s = base + lane;
        if (s >= num_steps) continue;
        const float2 h = hl[s];
        float last_t = t0;
        if (s > 0) { const float2 hp = hl[s - 1]; last_t = hp.x + hp.y; }
        const size_t m = (size_t)point_index + s;
        xyzs[m * 3] = cn_clamp(cn_fma(h.x, r.dx, r.ox), -bound, bound);
        xyzs[m * 3 + 1] = cn_clamp(cn_fma(h.x, r.dy, r.oy), -bound, bound);
        xyzs[m * 3 + 2] = cn_clamp(cn_fma(h.x, r.dz, r.oz), -bound, bound);
        dirs[m * 3] = r.dx; dirs[m * 3 + 1] = r.dy; dirs[m * 3 + 2] = r.dz;
        deltas[m * 2] = h.y;
        deltas[m * 2 + 1] = (h.x + h.y) - last_t;
    }
}

// ------------------------------------------------------------------------------------------------ compositing (training)
// composite_rays_train forward / backward (raymarching.cu:500-577, 691-772) with one WAVE per ray: the ray's samples sit on the 64 lanes
// (chunks of 64), transmittance and the running colour / path-length sums are DPP wave scans.  The reference's thread-per-ray form keeps
// 16384 serial loops of ~125 dependent iterations on one wave per CU (measured here: 0.35 + 0.38 ms per 128x128 view); this form reads
// every sample once, coalesced.  Association of the sums / products differs from the serial loop (rounding level); a sample is kept iff
// it is the first or the transmittance before it is >= T_thresh — the serial loop's `if (T < T_thresh) break` after the update,
// restated per sample.
__device__ __forceinline__ float rm_incl_prod(float x) {
    x *= rm_dpp<0x111>(1.0f, x); x *= rm_dpp<0x112>(1.0f, x); x *= rm_dpp<0x114>(1.0f, x); x *= rm_dpp<0x118>(1.0f, x);
    x *= rm_dpp<0x142, 0xA>(1.0f, x); x *= rm_dpp<0x143, 0xC>(1.0f, x);
    return x;
}
__device__ __forceinline__ float rm_lane63(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63)); }

#define RMW_WAVES 4
template <int RS>
__global__ void __launch_bounds__(RMW_WAVES * 64) k_composite_train_fwd_wave(const float *__restrict__ sigmas, const float *__restrict__ rgbs,
                                                                              const float *__restrict__ deltas, const int *__restrict__ rays, uint32_t M,
                                                                              uint32_t N, float T_thresh, float *__restrict__ weights_sum,
                                                                              float *__restrict__ depth, float *__restrict__ image) {
    const uint32_t lane = threadIdx.x & 63, n = blockIdx.x * RMW_WAVES + (threadIdx.x >> 6);
    if (n >= N) return;                                                  // whole waves leave together
    const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
    float r = 0, g = 0, b = 0, ws = 0, d = 0;
    if (!(num_steps == 0 || offset + num_steps > M)) {
        const float *ps = sigmas + offset, *pc = rgbs + (size_t)offset * RS;
        const float2 *pl = reinterpret_cast<const float2 *>(deltas) + offset;
        float T_carry = 1.0f, t_carry = 0.0f;
        for (uint32_t base = 0; base < num_steps; base += 64) {
            const uint32_t i = base + lane;
            const bool ok = i < num_steps;
            float2 dl = make_float2(0.0f, 0.0f);
            float sigma = 0, c0 = 0, c1 = 0, c2 = 0;
            if (ok) {
                dl = pl[i]; sigma = ps[i];
                c0 = pc[(size_t)i * RS]; c1 = pc[(size_t)i * RS + 1]; c2 = pc[(size_t)i * RS + 2];
            }
            const float alpha = ok ? 1.0f - __expf(-sigma * dl.x) : 0.0f;
            const float incl = rm_incl_prod(1.0f - alpha);
            const float T_before = rm_dpp<0x138>(1.0f, incl) * T_carry;  // exclusive product, times the chunks before
            const float t = cn_wave_incl_scan(dl.y) + t_carry;          // path length up to and including this sample
            const bool keep = ok && (i == 0 || T_before >= T_thresh);
            const float w = keep ? alpha * T_before : 0.0f;
            r = cn_fma(w, c0, r); g = cn_fma(w, c1, g); b = cn_fma(w, c2, b);
            d = cn_fma(w, t, d);
            ws += w;
            T_carry *= rm_lane63(incl);
            t_carry = rm_lane63(t);
            if (T_carry < T_thresh) break;                               // wave-uniform: nothing after this chunk is kept
        }
        r = cn_wave_sum(r); g = cn_wave_sum(g); b = cn_wave_sum(b); d = cn_wave_sum(d); ws = cn_wave_sum(ws);
    }
    if (lane == 0) {
        weights_sum[index] = ws;
        depth[index] = d;
        image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
    }
}

template <int RS>
__global__ void __launch_bounds__(RMW_WAVES * 64) k_composite_train_bwd_wave(const float *__restrict__ grad_weights_sum, const float *__restrict__ grad_image,
                                                                              const float *__restrict__ sigmas, const float *__restrict__ rgbs,
                                                                              const float *__restrict__ deltas, const int *__restrict__ rays,
                                                                              const float *__restrict__ weights_sum, const float *__restrict__ image,
                                                                              uint32_t M, uint32_t N, float T_thresh, float *__restrict__ grad_sigmas,
                                                                              float *__restrict__ grad_rgbs) {
    const uint32_t lane = threadIdx.x & 63, n = blockIdx.x * RMW_WAVES + (threadIdx.x >> 6);
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
    if (num_steps == 0 || offset + num_steps > M) return;
    const float gws = grad_weights_sum[index];
    const float gi0 = grad_image[index * 3], gi1 = grad_image[index * 3 + 1], gi2 = grad_image[index * 3 + 2];
    const float r_final = image[index * 3], g_final = image[index * 3 + 1], b_final = image[index * 3 + 2];
    const float ws_final = weights_sum[index];
    const float *ps = sigmas + offset, *pc = rgbs + (size_t)offset * RS;
    const float2 *pl = reinterpret_cast<const float2 *>(deltas) + offset;
    float *gs = grad_sigmas + offset, *gc = grad_rgbs + (size_t)offset * RS;
    float T_carry = 1.0f, r_carry = 0.0f, g_carry = 0.0f, b_carry = 0.0f;
    for (uint32_t base = 0; base < num_steps; base += 64) {
        const uint32_t i = base + lane;
        const bool ok = i < num_steps;
        float dl = 0, sigma = 0, c0 = 0, c1 = 0, c2 = 0;
        if (ok) {
            dl = pl[i].x; sigma = ps[i];
            c0 = pc[(size_t)i * RS]; c1 = pc[(size_t)i * RS + 1]; c2 = pc[(size_t)i * RS + 2];
        }
        const float alpha = ok ? 1.0f - __expf(-sigma * dl) : 0.0f;
        const float incl = rm_incl_prod(1.0f - alpha);
        const float T_before = rm_dpp<0x138>(1.0f, incl) * T_carry;
        const float T_after = incl * T_carry;
        const bool keep = ok && (i == 0 || T_before >= T_thresh);
        const float w = keep ? alpha * T_before : 0.0f;
        const float r = cn_wave_incl_scan(w * c0) + r_carry, g = cn_wave_incl_scan(w * c1) + g_carry, b = cn_wave_incl_scan(w * c2) + b_carry;
        if (keep) {
            gc[(size_t)i * RS] = gi0 * w; gc[(size_t)i * RS + 1] = gi1 * w; gc[(size_t)i * RS + 2] = gi2 * w;
            gs[i] = dl * (gi0 * cn_fma(T_after, c0, -(r_final - r)) + gi1 * cn_fma(T_after, c1, -(g_final - g)) +
                          gi2 * cn_fma(T_after, c2, -(b_final - b)) + gws * (1 - ws_final));
        }
        T_carry *= rm_lane63(incl);
        r_carry = rm_lane63(r); g_carry = rm_lane63(g); b_carry = rm_lane63(b);
        if (T_carry < T_thresh) break;
    }
}

// ------------------------------------------------------------------------------------------------ inference
__global__ void __launch_bounds__(RM_BLOCK) k_march_rays(uint32_t n_alive, uint32_t n_step, const int *__restrict__ rays_alive,
                                                         const float *__restrict__ rays_t, const float *__restrict__ rays_o,
                                                         const float *__restrict__ rays_d, float bound, float dt_gamma, uint32_t max_steps,
                                                         uint32_t C, uint32_t H, const uint8_t *__restrict__ grid,
                                                         const float *__restrict__ fars, float *__restrict__ xyzs, float *__restrict__ dirs,
                                                         float *__restrict__ deltas, const float *__restrict__ noises) {
    const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
    if (n >= n_alive) return;
    const int index = rays_alive[n];
    const MarchConst k = rm_consts(bound, dt_gamma, max_steps, C, H);
    const RayState r = rm_load_ray(rays_o, rays_d, (uint32_t)index);
    float *px = xyzs + (size_t)n * n_step * 3, *pd = dirs + (size_t)n * n_step * 3, *pl = deltas + (size_t)n * n_step * 2;
    float t = rays_t[index];
    const float far = fars[index];
    t = cn_fma(cn_clamp(t * dt_gamma, k.dt_min, k.dt_max), noises[n], t);
    float last_t = t;
    uint32_t step = 0;
    while (t < far && step < n_step) {
        const Probe s = rm_probe(r, t, k, grid);
        if (s.occ) {
            px[0] = s.x; px[1] = s.y; px[2] = s.z;
            pd[0] = r.dx; pd[1] = r.dy; pd[2] = r.dz;
            t += s.dt;
            pl[0] = s.dt;
            pl[1] = t - last_t;
            last_t = t;
            px += 3; pd += 3; pl += 2;
            step++;
        }
    }
}

template <int RS>
__global__ void __launch_bounds__(RM_BLOCK) k_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int *__restrict__ rays_alive,
                                                             float *__restrict__ rays_t, const float *__restrict__ sigmas,
                                                             const float *__restrict__ rgbs, const float *__restrict__ deltas,
                                                             float *__restrict__ weights_sum, float *__restrict__ depth, float *__restrict__ image) {
    const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
    if (n >= n_alive) return;
    const int index = rays_alive[n];
    const float *ps = sigmas + (size_t)n * n_step, *pc = rgbs + (size_t)n * n_step * RS, *pl = deltas + (size_t)n * n_step * 2;
    float t = rays_t[index];
    float weight_sum = weights_sum[index], d = depth[index];
    float r = image[index * 3], g = image[index * 3 + 1], b = image[index * 3 + 2];
    uint32_t step = 0;
    while (step < n_step) {
        if (pl[0] == 0) break;
        const float alpha = 1.0f - __expf(-ps[0] * pl[0]);
        const float T = 1 - weight_sum;
        const float weight = alpha * T;
        weight_sum += weight;
        t += pl[1];
        d = cn_fma(weight, t, d);
        r = cn_fma(weight, pc[0], r); g = cn_fma(weight, pc[1], g); b = cn_fma(weight, pc[2], b);
        if (T < T_thresh) break;
        ps++; pc += RS; pl += 2; step++;
    }
    if (step < n_step) rays_alive[n] = -1; else rays_t[index] = t;
    weights_sum[index] = weight_sum;
    depth[index] = d;
    image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
}

// Order-preserving compaction of the alive list: ballot + popcount inside a wave, LDS for the wave bases,
// a running base across the 1024-ray tiles.  One workgroup: the list is at most H*W entries and shrinks fast.
__global__ void __launch_bounds__(SCAN_THREADS) k_compact_alive(const int *__restrict__ in, uint32_t n, int *__restrict__ out, int *__restrict__ count) {
    __shared__ uint32_t wave_cnt[SCAN_THREADS / CN_WAVE];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t base = 0;
    for (uint32_t start = 0; start < n; start += SCAN_THREADS) {
        const uint32_t i = start + tid;
        const int v = (i < n) ? in[i] : -1;
        const bool keep = v >= 0;
        const unsigned long long mask = __ballot(keep);
        const uint32_t before = __popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = (uint32_t)__popcll(mask);
        __syncthreads();
        uint32_t wbase = 0, tot = 0;
#pragma unroll
        for (uint32_t w = 0; w < SCAN_THREADS / CN_WAVE; w++) {
            const uint32_t c = wave_cnt[w];
            if (w < wave) wbase += c;
            tot += c;
        }
        if (keep) out[base + wbase + before] = v;
        base += tot;
        __syncthreads();
    }
    if (tid == 0) *count = (int)base;
}

// ------------------------------------------------------------------------------------------------ C-ABI
#define RM_GRID(n) dim3(cn_div_up((n), RM_BLOCK)), dim3(RM_BLOCK), 0, CN_STREAM(stream)

extern "C" {

int cnerf_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb, uint32_t N, float min_near, float *nears,
                             float *fars, void *stream) {
    if (N == 0) return CNERF_OK;
    if (!rays_o || !rays_d || !aabb || !nears || !fars) return CNERF_ENULL;
    hipLaunchKernelGGL(k_near_far_from_aabb, RM_GRID(N), rays_o, rays_d, aabb, N, min_near, nears, fars);
    return cn_launch_status();
}

int cnerf_sph_from_ray(const float *rays_o, const float *rays_d, float radius, uint32_t N, float *coords, void *stream) {
    if (N == 0) return CNERF_OK;
    if (!rays_o || !rays_d || !coords) return CNERF_ENULL;
    hipLaunchKernelGGL(k_sph_from_ray, RM_GRID(N), rays_o, rays_d, radius, N, coords);
    return cn_launch_status();
}

int cnerf_morton3D(const int32_t *coords, uint32_t N, int32_t *indices, void *stream) {
    if (N == 0) return CNERF_OK;
    if (!coords || !indices) return CNERF_ENULL;
    hipLaunchKernelGGL(k_morton3D, RM_GRID(N), coords, N, indices);
    return cn_launch_status();
}

int cnerf_morton3D_invert(const int32_t *indices, uint32_t N, int32_t *coords, void *stream) {
    if (N == 0) return CNERF_OK;
    if (!coords || !indices) return CNERF_ENULL;
    hipLaunchKernelGGL(k_morton3D_invert, RM_GRID(N), indices, N, coords);
    return cn_launch_status();
}

int cnerf_packbits(const float *grid, uint32_t N, float density_thresh, uint8_t *bitfield, void *stream) {
    if (N == 0) return CNERF_OK;
    if (!grid || !bitfield) return CNERF_ENULL;
    if (((uintptr_t)grid) & 15) return CNERF_EINVAL;
    hipLaunchKernelGGL(k_packbits, RM_GRID(N), grid, N, density_thresh, bitfield);
    return cn_launch_status();
}

// budget = the caller's sample budget M (fixed-budget march: enables the overflow rotation of k_march_train_scan) or 0xFFFFFFFF
static int rm_count_budget(const float *rays_o, const float *rays_d, const uint8_t *grid, float bound, float dt_gamma, uint32_t max_steps, uint32_t N,
                           uint32_t C, uint32_t H, const float *nears, const float *fars, int32_t *rays, int32_t *counter, const float *noises,
                           uint32_t budget, void *stream) {
    if (C == 0 || C > 8 || H == 0 || H > 1024 || max_steps == 0) return CNERF_EINVAL;
    if (N == 0) return CNERF_OK;
    if (!rays_o || !rays_d || !grid || !nears || !fars || !rays || !counter || !noises) return CNERF_ENULL;
    hipLaunchKernelGGL(k_march_train_count_wave, dim3(cn_div_up(N, 4)), dim3(256), 0, CN_STREAM(stream), rays_o, rays_d, grid, bound, dt_gamma, max_steps, N,
                       C, H, nears, fars, rays, noises, (float2 *)nullptr);
    hipLaunchKernelGGL(k_march_train_scan, dim3(1), dim3(SCAN_THREADS), 0, CN_STREAM(stream), rays, counter, N, budget, noises);
    return cn_launch_status();
}

int cnerf_march_rays_train_count(const float *rays_o, const float *rays_d, const uint8_t *grid, float bound, float dt_gamma,
                                 uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, const float *nears, const float *fars,
                                 int32_t *rays, int32_t *counter, const float *noises, void *stream) {
    return rm_count_budget(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, nears, fars, rays, counter, noises, 0xFFFFFFFFu, stream);
}

int cnerf_march_rays_train_count_hits(const float *rays_o, const float *rays_d, const uint8_t *grid, float bound, float dt_gamma,
                                      uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, const float *nears, const float *fars,
                                      int32_t *rays, int32_t *counter, const float *noises, float *hits, void *stream) {
    if (C == 0 || C > 8 || H == 0 || H > 1024 || max_steps == 0) return CNERF_EINVAL;
    if (N == 0) return CNERF_OK;
    if (!rays_o || !rays_d || !grid || !nears || !fars || !rays || !counter || !noises || !hits) return CNERF_ENULL;
    if (((uintptr_t)hits) & 7) return CNERF_EINVAL;
    hipLaunchKernelGGL(k_march_train_count_wave, dim3(cn_div_up(N, 4)), dim3(256), 0, CN_STREAM(stream), rays_o, rays_d, grid, bound, dt_gamma, max_steps, N,
                       C, H, nears, fars, rays, noises, reinterpret_cast<float2 *>(hits));
    hipLaunchKernelGGL(k_march_train_scan, dim3(1), dim3(SCAN_THREADS), 0, CN_STREAM(stream), rays, counter, N, 0xFFFFFFFFu, noises);
    return cn_launch_status();
}

int cnerf_march_rays_train_write_hits(const float *rays_o, const float *rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t N,
                                      uint32_t C, uint32_t H, uint32_t M, const float *nears, const float *noises, const float *hits,
                                      const int32_t *rays, float *xyzs, float *dirs, float *deltas, void *stream) {
    if (C == 0 || C > 8 || H == 0 || H > 1024 || max_steps == 0) return CNERF_EINVAL;
    if (N == 0 || M == 0) return CNERF_OK;
    if (!rays_o || !rays_d || !nears || !noises || !hits || !rays || !xyzs || !dirs || !deltas) return CNERF_ENULL;
    if (((uintptr_t)hits) & 7) return CNERF_EINVAL;
    hipLaunchKernelGGL(k_march_train_write_hits, dim3(cn_div_up(N, 4)), dim3(256), 0, CN_STREAM(stream), rays_o, rays_d, bound, dt_gamma, max_steps, N, C, H, M,
                       nears, noises, reinterpret_cast<const float2 *>(hits), rays, xyzs, dirs, deltas);
    return cn_launch_status();
}

int cnerf_march_rays_train_write(const float *rays_o, const float *rays_d, const uint8_t *grid, float bound, float dt_gamma,
                                 uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float *nears,
                                 const float *fars, float *xyzs, float *dirs, float *deltas, const int32_t *rays, const float *noises,
                                 void *stream) {
    if (C == 0 || C > 8 || H == 0 || H > 1024 || max_steps == 0) return CNERF_EINVAL;
    if (N == 0 || M == 0) return CNERF_OK;
    if (!rays_o || !rays_d || !grid || !nears || !fars || !rays || !noises) return CNERF_ENULL;
    if (!xyzs || !dirs || !deltas) return CNERF_ENULL;
    hipLaunchKernelGGL(k_march_train_write, RM_GRID(N), rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs,
                       deltas, rays, noises);
    return cn_launch_status();
}

int cnerf_march_rays_train(const float *rays_o, const float *rays_d, const uint8_t *grid, float bound, float dt_gamma, uint32_t max_steps,
                           uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float *nears, const float *fars, float *xyzs, float *dirs,
                           float *deltas, int32_t *rays, int32_t *counter, const float *noises, void *stream) {
    int rc = rm_count_budget(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, nears, fars, rays, counter, noises, M, stream);
    if (rc) return rc;
    return cnerf_march_rays_train_write(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas, rays,
                                        noises, stream);
}

int cnerf_composite_rays_train_forward(const float *sigmas, const float *rgbs, const float *deltas, const int32_t *rays, uint32_t M, uint32_t N,
                                       float T_thresh, float *weights_sum, float *depth, float *image, uint32_t rgb_stride, void *stream) {
    if (rgb_stride != 3 && rgb_stride != 4) return CNERF_EINVAL;
    if (N == 0) return CNERF_OK;
    if (!rays || !weights_sum || !depth || !image) return CNERF_ENULL;
    if (M > 0 && (!sigmas || !rgbs || !deltas)) return CNERF_ENULL;
    const dim3 wgrid(cn_div_up(N, RMW_WAVES)), wblock(RMW_WAVES * 64);
    if (rgb_stride == 3) hipLaunchKernelGGL(k_composite_train_fwd_wave<3>, wgrid, wblock, 0, CN_STREAM(stream), sigmas, rgbs, deltas, rays, M, N, T_thresh, weights_sum, depth, image);
    else hipLaunchKernelGGL(k_composite_train_fwd_wave<4>, wgrid, wblock, 0, CN_STREAM(stream), sigmas, rgbs, deltas, rays, M, N, T_thresh, weights_sum, depth, image);
    return cn_launch_status();
}

int cnerf_composite_rays_train_backward(const float *grad_weights_sum, const float *grad_image, const float *sigmas, const float *rgbs,
                                        const float *deltas, const int32_t *rays, const float *weights_sum, const float *image, uint32_t M,
                                        uint32_t N, float T_thresh, float *grad_sigmas, float *grad_rgbs, uint32_t rgb_stride, void *stream) {
    if (rgb_stride != 3 && rgb_stride != 4) return CNERF_EINVAL;
    if (N == 0 || M == 0) return CNERF_OK;
    if (!rays || !weights_sum || !image || !grad_weights_sum || !grad_image) return CNERF_ENULL;
    if (!sigmas || !rgbs || !deltas || !grad_sigmas || !grad_rgbs) return CNERF_ENULL;
    const dim3 wgrid(cn_div_up(N, RMW_WAVES)), wblock(RMW_WAVES * 64);
    if (rgb_stride == 3)
        hipLaunchKernelGGL(k_composite_train_bwd_wave<3>, wgrid, wblock, 0, CN_STREAM(stream), grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum,
                           image, M, N, T_thresh, grad_sigmas, grad_rgbs);
    else
        hipLaunchKernelGGL(k_composite_train_bwd_wave<4>, wgrid, wblock, 0, CN_STREAM(stream), grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum,
                           image, M, N, T_thresh, grad_sigmas, grad_rgbs);
    return cn_launch_status();
}

int cnerf_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t, const float *rays_o, const float *rays_d,
                     float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t *grid, const float *nears,
                     const float *fars, float *xyzs, float *dirs, float *deltas, const float *noises, void *stream) {
    (void)nears;
    if (C == 0 || C > 8 || H == 0 || H > 1024 || max_steps == 0 || n_step == 0) return CNERF_EINVAL;
    if (n_alive == 0) return CNERF_OK;
    if (!rays_alive || !rays_t || !rays_o || !rays_d || !grid || !fars || !xyzs || !dirs || !deltas || !noises) return CNERF_ENULL;
    hipLaunchKernelGGL(k_march_rays, RM_GRID(n_alive), n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid,
                       fars, xyzs, dirs, deltas, noises);
    return cn_launch_status();
}

int cnerf_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *rays_alive, float *rays_t, const float *sigmas,
                         const float *rgbs, const float *deltas, float *weights_sum, float *depth, float *image, uint32_t rgb_stride,
                         void *stream) {
    if (rgb_stride != 3 && rgb_stride != 4) return CNERF_EINVAL;
    if (n_alive == 0) return CNERF_OK;
    if (!rays_alive || !rays_t || !sigmas || !rgbs || !deltas || !weights_sum || !depth || !image) return CNERF_ENULL;
    if (rgb_stride == 3)
        hipLaunchKernelGGL(k_composite_rays<3>, RM_GRID(n_alive), n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image);
    else
        hipLaunchKernelGGL(k_composite_rays<4>, RM_GRID(n_alive), n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image);
    return cn_launch_status();
}

int cnerf_compact_rays_alive(const int32_t *rays_alive_in, uint32_t n, int32_t *rays_alive_out, int32_t *count, void *stream) {
    if (!count) return CNERF_ENULL;
    if (n > 0 && (!rays_alive_in || !rays_alive_out)) return CNERF_ENULL;
    hipLaunchKernelGGL(k_compact_alive, dim3(1), dim3(SCAN_THREADS), 0, CN_STREAM(stream), rays_alive_in, n, rays_alive_out, count);
    return cn_launch_status();
}

}  // extern "C"
