// Gather laboratory (round 3): the production forward gather against candidate layouts / schedules on ray-structured samples of the
// benchmark scene, with a bit-exactness check of every variant against the production kernel.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math scratch/gather_lab.hip -o scratch/gather_lab
//   scratch/gather_lab            (prints one line per variant)
#include "../customnerf_amd/csrc/gridencoder.hip"
#include <vector>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <random>
#include <functional>

// the binned backward lives in another translation unit of the library: not needed here
bool bn_eligible(uint32_t, uint32_t, uint32_t, uint32_t, const GridLevels &) { return false; }
uint64_t bn_workspace_bytes(uint32_t, uint32_t, const GridLevels &, int) { return 0; }
int bn_backward(const void *, const float *, const GridLevels &, float *, uint32_t, uint32_t, uint32_t, int, uint32_t, int, void *, hipStream_t, bool) { return -1; }
int bn_prepare(const float *, const GridLevels &, uint32_t, uint32_t, uint32_t, int, uint32_t, int, void *, hipStream_t) { return -1; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// ------------------------------------------------------------------------------------------------ candidate kernel
// Work classes:
//   LDS class   workgroups 0 .. n_lds-1: each owns a contiguous range of points and evaluates the levels whose tables it stages in LDS
//               (phase 1: levels [lds_a0, lds_a1), phase 2: levels [lds_b0, lds_b1) — empty ranges allowed); no texture path at all;
//   TA class    the remaining workgroups: (level, point block) items of the other levels, XCD-swizzled cost-balanced slices as in production;
//               dense levels optionally read a cell-major expansion (8 corners = 32 contiguous bytes per cell).
struct LabCfg {
    uint32_t n_lds;               // LDS-class workgroups (0 = none)
    uint32_t lds_a0, lds_a1, lds_b0, lds_b1;
    uint32_t pts_per_lds;         // points per LDS workgroup (multiple of 256)
    uint32_t cell_mask;           // bit l: level l reads the cell-major expansion
    uint32_t cell_off[GE_MAX_LEVELS];   // first cell record (32 B units) of level l in the expansion buffer
    uint32_t ta_levels;           // levels in the TA class work list (lv.order[0 .. ta_levels-1])
    uint32_t nb;                  // point blocks
};

__device__ __forceinline__ uint32_t lab_eval_dense_lds(const uint32_t *__restrict__ tab, const float (&in)[3], float scale, uint32_t resolution) {
    float fr[3], om[3];
    uint32_t pg[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const float pos = cn_fma(in[d], scale, 0.5f);
        pg[d] = (uint32_t)floorf(pos);
        fr[d] = pos - (float)pg[d];
        om[d] = 1 - fr[d];
    }
    const float a00 = om[0] * om[1], a10 = fr[0] * om[1], a01 = om[0] * fr[1], a11 = fr[0] * fr[1];
    const float w[8] = {a00 * om[2], a10 * om[2], a01 * om[2], a11 * om[2], a00 * fr[2], a10 * fr[2], a01 * fr[2], a11 * fr[2]};
    const uint32_t s1 = resolution + 1, s2 = s1 * s1;
    const uint32_t i00 = pg[0] + pg[1] * s1 + pg[2] * s2;
    uint32_t c[8];
    c[0] = tab[i00]; c[1] = tab[i00 + 1];
    c[2] = tab[i00 + s1]; c[3] = tab[i00 + s1 + 1];
    c[4] = tab[i00 + s2]; c[5] = tab[i00 + s2 + 1];
    c[6] = tab[i00 + s1 + s2]; c[7] = tab[i00 + s1 + s2 + 1];
    const float negzero = -0.0f;
    cn_gf_h2 acc = {(_Float16)0, (_Float16)0};
#pragma unroll
    for (int k = 0; k < 8; k++) gf_accum(acc, w[k], c[k], negzero);
    return __builtin_bit_cast(uint32_t, acc);
}

__global__ void __launch_bounds__(GE_BLOCK) k_lab(const float *__restrict__ inputs, const __half *__restrict__ grid, const uint4 *__restrict__ cells,
                                                  const GridLevels lv, const LabCfg cfg, __half *__restrict__ outputs, uint32_t B, uint32_t gridtype,
                                                  uint32_t ostride) {
    extern __shared__ uint32_t lab_tab[];
    if (blockIdx.x < cfg.n_lds) {
        // ---------------- LDS class
        const uint32_t p0 = blockIdx.x * cfg.pts_per_lds;
        const uint32_t p1 = min(B, p0 + cfg.pts_per_lds);
        for (int phase = 0; phase < 2; phase++) {
            const uint32_t l0 = phase ? cfg.lds_b0 : cfg.lds_a0, l1 = phase ? cfg.lds_b1 : cfg.lds_a1;
            if (l0 >= l1) continue;
            if (phase) __syncthreads();
            const uint32_t e0 = lv.offset[l0], e1 = lv.offset[l1 - 1] + lv.size[l1 - 1];      // contiguous entries of the staged levels
            const uint4 *src = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint32_t *>(grid) + e0);   // level offsets are multiples of 8 entries
            const uint32_t n4 = (e1 - e0) / 4;
            for (uint32_t i = threadIdx.x; i < n4; i += GE_BLOCK) reinterpret_cast<uint4 *>(lab_tab)[i] = src[i];
            __syncthreads();
            for (uint32_t b = p0 + threadIdx.x; b < p1; b += GE_BLOCK) {
                float in[3];
                ge_load_coords<3>(inputs, b, in);
                const bool oob = in[0] < 0 || in[0] > 1 || in[1] < 0 || in[1] > 1 || in[2] < 0 || in[2] > 1;
                for (uint32_t l = l0; l < l1; l++) {
                    uint32_t *out = reinterpret_cast<uint32_t *>(outputs) + ((size_t)l * ostride + b);
                    *out = oob ? 0u : lab_eval_dense_lds(lab_tab + (lv.offset[l] - e0), in, lv.scale[l], lv.resolution[l]);
                }
            }
        }
        return;
    }
    // ---------------- TA class: blockIdx.x - n_lds indexes the swizzled list
    uint32_t level, pb;
    {
        const uint32_t bid = blockIdx.x - cfg.n_lds;
        const uint32_t xcd = bid % CN_NXCD, k = bid / CN_NXCD;
        const uint32_t w = lv.xcd_first[xcd] + k;
        if (w >= lv.xcd_first[xcd + 1]) return;
        level = lv.order[w / cfg.nb];
        pb = w % cfg.nb;
    }
    const uint32_t b = pb * GE_BLOCK + threadIdx.x;
    if (b >= B) return;
    uint32_t *out = reinterpret_cast<uint32_t *>(outputs) + ((size_t)level * ostride + b);
    float in[3];
    ge_load_coords<3>(inputs, b, in);
    if (in[0] < 0 || in[0] > 1 || in[1] < 0 || in[1] > 1 || in[2] < 0 || in[2] > 1) { *out = 0u; return; }
    const uint32_t size = lv.size[level], resolution = lv.resolution[level];
    const float scale = lv.scale[level];
    const unsigned char *__restrict__ table = reinterpret_cast<const unsigned char *>(grid) + (size_t)lv.offset[level] * 4;
    float fr[3], om[3];
    uint32_t pg[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const float pos = cn_fma(in[d], scale, 0.5f);
        pg[d] = (uint32_t)floorf(pos);
        fr[d] = pos - (float)pg[d];
        om[d] = 1 - fr[d];
    }
    const float a00 = om[0] * om[1], a10 = fr[0] * om[1], a01 = om[0] * fr[1], a11 = fr[0] * fr[1];
    const float w[8] = {a00 * om[2], a10 * om[2], a01 * om[2], a11 * om[2], a00 * fr[2], a10 * fr[2], a01 * fr[2], a11 * fr[2]};
    uint32_t c[8];
    const int mode = ge_level_mode<3>(gridtype, false, size, resolution);
    if (mode == GE_MODE_DENSE) {
        if ((cfg.cell_mask >> level) & 1u) {
            const uint32_t cell = pg[0] + pg[1] * resolution + pg[2] * resolution * resolution;
            const uint4 *rec = cells + ((size_t)cfg.cell_off[level] + cell) * 2;
            const uint4 r0 = rec[0], r1 = rec[1];
            c[0] = r0.x; c[1] = r0.y; c[2] = r0.z; c[3] = r0.w; c[4] = r1.x; c[5] = r1.y; c[6] = r1.z; c[7] = r1.w;
        } else {
            const uint32_t s1 = resolution + 1, s2 = s1 * s1;
            const uint32_t i00 = (pg[0] + pg[1] * s1 + pg[2] * s2) * 4u;
            const gf_u2 r0 = *reinterpret_cast<const gf_u2 *>(table + i00);
            const gf_u2 r1 = *reinterpret_cast<const gf_u2 *>(table + (i00 + s1 * 4u));
            const gf_u2 r2 = *reinterpret_cast<const gf_u2 *>(table + (i00 + s2 * 4u));
            const gf_u2 r3 = *reinterpret_cast<const gf_u2 *>(table + (i00 + (s1 + s2) * 4u));
            c[0] = r0.x; c[1] = r0.y; c[2] = r1.x; c[3] = r1.y; c[4] = r2.x; c[5] = r2.y; c[6] = r3.x; c[7] = r3.y;
        }
    } else {
        const uint32_t mask = size - 1;
        const uint32_t hy0 = pg[1] * 2654435761u, hy1 = hy0 + 2654435761u, hz0 = pg[2] * 805459861u, hz1 = hz0 + 805459861u;
        const uint32_t x0 = pg[0], xm = x0 ^ (x0 + 1);
        const bool in_quad = (xm & mask) < 4u;
        const bool odd = (x0 & 1u) != 0;
        const uint32_t hyz[4] = {hy0 ^ hz0, hy1 ^ hz0, hy0 ^ hz1, hy1 ^ hz1};
        uint32_t i0[4];
        gf_u4 v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            i0[q] = (x0 ^ hyz[q]) & mask;
            v[q] = *reinterpret_cast<const gf_u4 *>(table + ((i0[q] & ~3u) * 4u));
        }
        uint32_t far[4] = {0u, 0u, 0u, 0u};
        if (!in_quad) {
#pragma unroll
            for (int q = 0; q < 4; q++) far[q] = gf_ld1(table, ((i0[q] ^ xm) & mask) * 4u);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const bool b0 = (i0[q] & 1u) != 0, b1 = (i0[q] & 2u) != 0;
            const uint32_t plo = b1 ? v[q].z : v[q].x, phi = b1 ? v[q].w : v[q].y;
            c[2 * q] = b0 ? phi : plo;
            const uint32_t same = b0 ? plo : phi;
            const uint32_t olo = b1 ? v[q].x : v[q].z, ohi = b1 ? v[q].y : v[q].w;
            const uint32_t cross = b0 ? olo : ohi;
            const uint32_t near = odd ? cross : same;
            c[2 * q + 1] = in_quad ? near : far[q];
        }
    }
    const float negzero = -0.0f;
    cn_gf_h2 acc = {(_Float16)0, (_Float16)0};
#pragma unroll
    for (int k = 0; k < 8; k++) gf_accum(acc, w[k], c[k], negzero);
    *out = __builtin_bit_cast(uint32_t, acc);
}

// ------------------------------------------------------------------------------------------------ hashed-level limiter probes
// FLAGS: 1 = no fallback loads (results wrong for x = 3 mod 4), 2 = only two of the four windows loaded (wrong), 4 = non-temporal table loads,
//        8 = L1-bypassing (agent-scope relaxed atomic = sc1) table loads, 16 = non-temporal coordinate load / output store
template <int FLAGS, int WPS, int RPW = 0>
__global__ void __launch_bounds__(GE_BLOCK, WPS) k_lab_hash(const float *__restrict__ inputs, const __half *__restrict__ grid, const GridLevels lv,
                                                           __half *__restrict__ outputs, uint32_t B, uint32_t nb, uint32_t ta_levels, uint32_t ostride,
                                                           uint32_t T = 64) {
    uint32_t level, pb;
    {
        const uint32_t xcd = blockIdx.x % CN_NXCD, k = blockIdx.x / CN_NXCD;
        const uint32_t w = lv.xcd_first[xcd] + k;
        if (w >= lv.xcd_first[xcd + 1]) return;
        level = lv.order[w / nb];
        pb = w % nb;
    }
    uint32_t b = pb * GE_BLOCK + threadIdx.x;
    if constexpr (RPW != 0) {
        // tile remap inside the [ray][sample] layout: a workgroup covers RPW rays x (256 / RPW) consecutive samples
        constexpr uint32_t SPW = GE_BLOCK / (RPW ? RPW : 1);
        const uint32_t tiles_per_raygroup = T / SPW;                      // sample tiles along a ray
        const uint32_t rg = pb / tiles_per_raygroup, st = pb % tiles_per_raygroup;
        b = (rg * RPW + threadIdx.x / SPW) * T + st * SPW + threadIdx.x % SPW;
    }
    if (b >= B) return;
    uint32_t *out = reinterpret_cast<uint32_t *>(outputs) + ((size_t)level * ostride + b);
    float in[3];
    if (FLAGS & 16) { in[0] = __builtin_nontemporal_load(inputs + (size_t)b * 3); in[1] = __builtin_nontemporal_load(inputs + (size_t)b * 3 + 1); in[2] = __builtin_nontemporal_load(inputs + (size_t)b * 3 + 2); }
    else ge_load_coords<3>(inputs, b, in);
    if (in[0] < 0 || in[0] > 1 || in[1] < 0 || in[1] > 1 || in[2] < 0 || in[2] > 1) { *out = 0u; return; }
    const uint32_t size = lv.size[level];
    const float scale = lv.scale[level];
    const unsigned char *__restrict__ table = reinterpret_cast<const unsigned char *>(grid) + (size_t)lv.offset[level] * 4;
    float fr[3], om[3];
    uint32_t pg[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const float pos = cn_fma(in[d], scale, 0.5f);
        pg[d] = (uint32_t)floorf(pos);
        fr[d] = pos - (float)pg[d];
        om[d] = 1 - fr[d];
    }
    const float a00 = om[0] * om[1], a10 = fr[0] * om[1], a01 = om[0] * fr[1], a11 = fr[0] * fr[1];
    const float w[8] = {a00 * om[2], a10 * om[2], a01 * om[2], a11 * om[2], a00 * fr[2], a10 * fr[2], a01 * fr[2], a11 * fr[2]};
    uint32_t c[8];
    const uint32_t mask = size - 1;
    const uint32_t hy0 = pg[1] * 2654435761u, hy1 = hy0 + 2654435761u, hz0 = pg[2] * 805459861u, hz1 = hz0 + 805459861u;
    const uint32_t x0 = pg[0], xm = x0 ^ (x0 + 1);
    const bool in_quad = (xm & mask) < 4u;
    const bool odd = (x0 & 1u) != 0;
    const uint32_t hyz[4] = {hy0 ^ hz0, hy1 ^ hz0, hy0 ^ hz1, hy1 ^ hz1};
    uint32_t i0[4];
    gf_u4 v[4];
    if constexpr ((FLAGS & 32) != 0) {
        // unaligned 16-byte windows: [lo, lo + 3] holds both x corners whenever their entries are at most 3 apart (83 % of the lanes
        // instead of the 75 % an aligned quad covers); the rest keeps the aligned quad + the lane-masked partner load
        uint32_t i1[4], base[4];
        bool cov[4];
        bool any_far = false;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            i0[q] = (x0 ^ hyz[q]) & mask;
            i1[q] = (i0[q] ^ xm) & mask;
            const uint32_t lo = min(i0[q], i1[q]), hi = max(i0[q], i1[q]);
            cov[q] = hi - lo <= 3u;
            base[q] = cov[q] ? min(lo, size - 4u) : (i0[q] & ~3u);
            any_far = any_far || !cov[q];
        }
        struct __attribute__((packed, aligned(4))) u4a { uint32_t x, y, z, w; };
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const u4a t = *reinterpret_cast<const u4a *>(table + base[q] * 4u);
            v[q].x = t.x; v[q].y = t.y; v[q].z = t.z; v[q].w = t.w;
        }
        uint32_t far[4] = {0u, 0u, 0u, 0u};
        if (any_far) {
#pragma unroll
            for (int q = 0; q < 4; q++) if (!cov[q]) far[q] = gf_ld1(table, i1[q] * 4u);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t k0 = i0[q] - base[q], k1 = i1[q] - base[q];
            const uint32_t a01 = (k0 & 1u) ? v[q].y : v[q].x, a23 = (k0 & 1u) ? v[q].w : v[q].z;
            c[2 * q] = (k0 & 2u) ? a23 : a01;
            const uint32_t b01 = (k1 & 1u) ? v[q].y : v[q].x, b23 = (k1 & 1u) ? v[q].w : v[q].z;
            const uint32_t e1 = (k1 & 2u) ? b23 : b01;
            c[2 * q + 1] = cov[q] ? e1 : far[q];
        }
        const float negzero = -0.0f;
        cn_gf_h2 acc = {(_Float16)0, (_Float16)0};
#pragma unroll
        for (int k = 0; k < 8; k++) gf_accum(acc, w[k], c[k], negzero);
        *out = __builtin_bit_cast(uint32_t, acc);
        return;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        i0[q] = (x0 ^ hyz[q]) & mask;
        const gf_u4 *p = reinterpret_cast<const gf_u4 *>(table + ((i0[q] & ~3u) * 4u));
        if ((FLAGS & 2) && q >= 2) { v[q] = v[q - 2]; continue; }
        if (FLAGS & 4) {
            const uint32_t *pw = reinterpret_cast<const uint32_t *>(p);
            typedef uint32_t u4v __attribute__((ext_vector_type(4)));
            const u4v t = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(pw));
            v[q].x = t.x; v[q].y = t.y; v[q].z = t.z; v[q].w = t.w;
        } else if (FLAGS & 8) {
            const unsigned long long *pq = reinterpret_cast<const unsigned long long *>(p);
            const unsigned long long lo = __hip_atomic_load(pq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), hi = __hip_atomic_load(pq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v[q].x = (uint32_t)lo; v[q].y = (uint32_t)(lo >> 32); v[q].z = (uint32_t)hi; v[q].w = (uint32_t)(hi >> 32);
        } else v[q] = *p;
    }
    uint32_t far[4] = {0u, 0u, 0u, 0u};
    if (!(FLAGS & 1) && !in_quad) {
#pragma unroll
        for (int q = 0; q < 4; q++) far[q] = gf_ld1(table, ((i0[q] ^ xm) & mask) * 4u);
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const bool b0 = (i0[q] & 1u) != 0, b1 = (i0[q] & 2u) != 0;
        const uint32_t plo = b1 ? v[q].z : v[q].x, phi = b1 ? v[q].w : v[q].y;
        c[2 * q] = b0 ? phi : plo;
        const uint32_t same = b0 ? plo : phi;
        const uint32_t olo = b1 ? v[q].x : v[q].z, ohi = b1 ? v[q].y : v[q].w;
        const uint32_t cross = b0 ? olo : ohi;
        const uint32_t near = odd ? cross : same;
        c[2 * q + 1] = in_quad ? near : far[q];
    }
    const float negzero = -0.0f;
    cn_gf_h2 acc = {(_Float16)0, (_Float16)0};
#pragma unroll
    for (int k = 0; k < 8; k++) gf_accum(acc, w[k], c[k], negzero);
    if (FLAGS & 16) __builtin_nontemporal_store(__builtin_bit_cast(uint32_t, acc), out);
    else *out = __builtin_bit_cast(uint32_t, acc);
}

// cell-major expansion of the dense levels in `mask`: record (cell) = the 8 corner entries in corner order x + 2 y + 4 z
__global__ void __launch_bounds__(256) k_build_cells(const uint32_t *__restrict__ grid, const GridLevels lv, uint32_t level, uint32_t cell_off, uint4 *__restrict__ cells) {
    const uint32_t res = lv.resolution[level], n = res * res * res;
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    const uint32_t x = c % res, y = (c / res) % res, z = c / (res * res);
    const uint32_t s1 = res + 1, s2 = s1 * s1;
    const uint32_t *t = grid + lv.offset[level];
    const uint32_t i = x + y * s1 + z * s2;
    uint4 r0 = {t[i], t[i + 1], t[i + s1], t[i + s1 + 1]};
    uint4 r1 = {t[i + s2], t[i + s2 + 1], t[i + s1 + s2], t[i + s1 + s2 + 1]};
    cells[((size_t)cell_off + c) * 2] = r0;
    cells[((size_t)cell_off + c) * 2 + 1] = r1;
}

// ------------------------------------------------------------------------------------------------ host
struct Scene { std::vector<float> unit; uint32_t B; };
static Scene make_samples(uint32_t HW, uint32_t S, bool fine_like, uint32_t patch = 1) {
    // camera on the radius-3.5 circle at elevation 20 degrees looking at the origin, fovy 50, aabb [-2,2]^3; S jittered samples per ray
    Scene sc; sc.B = HW * HW * S; sc.unit.resize((size_t)sc.B * 3);
    std::mt19937 rng(fine_like ? 7 : 3);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    const float el = 20.0f * 3.14159265f / 180.0f, th = 0.7f;
    const float eye[3] = {3.5f * cosf(el) * cosf(th), 3.5f * sinf(el), 3.5f * cosf(el) * sinf(th)};
    float fwd[3] = {-eye[0], -eye[1], -eye[2]};
    float n = sqrtf(fwd[0] * fwd[0] + fwd[1] * fwd[1] + fwd[2] * fwd[2]); for (auto &v : fwd) v /= n;
    float right[3] = {fwd[1] * 0 - fwd[2] * 1, fwd[2] * 0 - fwd[0] * 0, fwd[0] * 1 - fwd[1] * 0};
    n = sqrtf(right[0] * right[0] + right[1] * right[1] + right[2] * right[2]); for (auto &v : right) v /= n;
    const float up[3] = {right[1] * fwd[2] - right[2] * fwd[1], right[2] * fwd[0] - right[0] * fwd[2], right[0] * fwd[1] - right[1] * fwd[0]};
    const float f = 0.5f * HW / tanf(0.5f * 50.0f * 3.14159265f / 180.0f);
    size_t k = 0;
    for (uint32_t ri = 0; ri < HW * HW; ri++) {
        // ray order: row-major (patch 1) or patch-major (patch x patch pixel tiles, row-major inside a tile)
        uint32_t px, py;
        if (patch <= 1) { py = ri / HW; px = ri % HW; }
        else {
            const uint32_t per = patch * patch, tile = ri / per, in = ri % per, tiles_x = HW / patch;
            py = (tile / tiles_x) * patch + in / patch; px = (tile % tiles_x) * patch + in % patch;
        }
        const float cx = (px + 0.5f - HW / 2.0f) / f, cy = -(py + 0.5f - HW / 2.0f) / f;
        float d[3];
        for (int c = 0; c < 3; c++) d[c] = right[c] * cx + up[c] * cy + fwd[c];
        n = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]); for (auto &v : d) v /= n;
        float tn = 0.01f, tf = 1e9f;
        for (int c = 0; c < 3; c++) {
            const float a = (-2.f - eye[c]) / d[c], b2 = (2.f - eye[c]) / d[c];
            tn = fmaxf(tn, fminf(a, b2)); tf = fminf(tf, fmaxf(a, b2));
        }
        if (tf < tn) tf = tn;
        // fine-like: importance samples cluster around the middle of the segment (a surface), in draw order
        for (uint32_t s = 0; s < S; s++) {
            float t;
            if (!fine_like) t = tn + (tf - tn) * ((s + U(rng)) / S);
            else t = tn + (tf - tn) * (0.45f + 0.1f * (U(rng) + U(rng) - 1.0f));
            for (int c = 0; c < 3; c++) {
                float p = fminf(fmaxf(eye[c] + d[c] * t, -2.f), 2.f);
                sc.unit[k++] = (p + 2.f) * 0.25f;
            }
        }
    }
    return sc;
}

static float time_it(const std::function<void()> &fn, int iters = 20) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) fn();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; i++) fn();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.0f / iters;
}

int main(int argc, char **argv) {
    const uint32_t L = 16, H = 16, T19 = 1u << 19;
    const float S = log2f(expf(logf(2048.0f / 16.0f) / 15.0f));
    // offsets as GridEncoder.__init__ (grid.py:124-135)
    std::vector<int32_t> offs(L + 1); uint32_t off = 0;
    const double pls = exp2(log2(2048.0 / 16.0) / 15.0);
    for (uint32_t l = 0; l < L; l++) {
        const uint32_t res = (uint32_t)ceil(16.0 * pow(pls, (double)l));
        uint64_t p = (uint64_t)(res + 1) * (res + 1) * (res + 1);
        if (p > T19) p = T19;
        p = (p + 7) / 8 * 8;
        offs[l] = off; off += (uint32_t)p;
    }
    offs[L] = off;
    printf("entries %u\n", off);
    std::vector<uint32_t> tab(off);
    std::mt19937 rng(1);
    for (auto &v : tab) { const __half a = __float2half(((rng() & 0xFFFF) / 65536.0f - 0.5f)), b = __float2half(((rng() & 0xFFFF) / 65536.0f - 0.5f)); v = (uint32_t)__half_as_ushort(a) | ((uint32_t)__half_as_ushort(b) << 16); }
    uint32_t *d_tab; CK(hipMalloc(&d_tab, (size_t)off * 4)); CK(hipMemcpy(d_tab, tab.data(), (size_t)off * 4, hipMemcpyHostToDevice));

    GridLevels lv;
    if (ge_levels(offs.data(), L, L, S, H, lv)) { printf("ge_levels failed\n"); return 1; }
    for (uint32_t l = 0; l < L; l++) printf("level %2u res %4u size %7u %s\n", l, lv.resolution[l], lv.size[l], (uint64_t)(lv.resolution[l] + 1) * (lv.resolution[l] + 1) * (lv.resolution[l] + 1) <= lv.size[l] ? "dense" : "hashed");

    // cell-major expansion of all dense levels
    LabCfg base{}; uint32_t coff = 0;
    for (uint32_t l = 0; l < 5; l++) { base.cell_off[l] = coff; coff += lv.resolution[l] * lv.resolution[l] * lv.resolution[l]; }
    uint4 *d_cells; CK(hipMalloc(&d_cells, (size_t)coff * 32));
    float t_cells = time_it([&] { for (uint32_t l = 2; l < 5; l++) { const uint32_t n = lv.resolution[l] * lv.resolution[l] * lv.resolution[l];
        hipLaunchKernelGGL(k_build_cells, dim3((n + 255) / 256), dim3(256), 0, 0, d_tab, lv, l, base.cell_off[l], d_cells); } });
    for (uint32_t l = 0; l < 2; l++) { const uint32_t n = lv.resolution[l] * lv.resolution[l] * lv.resolution[l];
        hipLaunchKernelGGL(k_build_cells, dim3((n + 255) / 256), dim3(256), 0, 0, d_tab, lv, l, base.cell_off[l], d_cells); }
    printf("cell expansion of levels 2-4: %.1f us (%u cells total over levels 0-4)\n", t_cells, coff);

    for (int fine = 0; fine < 2; fine++) {
        Scene sc = make_samples(128, 64, fine != 0);
        const uint32_t B = sc.B;
        float *d_in; CK(hipMalloc(&d_in, (size_t)B * 12)); CK(hipMemcpy(d_in, sc.unit.data(), (size_t)B * 12, hipMemcpyHostToDevice));
        __half *d_ref, *d_out; CK(hipMalloc(&d_ref, (size_t)L * B * 4)); CK(hipMalloc(&d_out, (size_t)L * B * 4));
        printf("---- %s samples, B = %u\n", fine ? "importance-like (draw order)" : "stratified coarse", B);
        auto prod = [&](uint32_t nl) { cnerf_grid_encode_forward_strided(d_in, d_tab, offs.data(), d_ref, B, 3, 2, L, nl, S, H, nullptr, 0, 0, 0, CNERF_F16, B, nullptr); };
        const float t_all = time_it([&] { prod(16); });
        printf("production k_grid_fwd_fast, 16 levels: %7.1f us  (%.2f of 8 TB/s)\n", t_all, (double)B * 588 / (t_all * 1e-6) / 8e12);
        for (uint32_t nl : {5u, 3u, 2u}) printf("production, levels 0..%u only:           %7.1f us\n", nl - 1, time_it([&] { prod(nl); }));
        prod(16); CK(hipDeviceSynchronize());
        std::vector<uint32_t> ref((size_t)L * B), got((size_t)L * B);
        CK(hipMemcpy(ref.data(), d_ref, (size_t)L * B * 4, hipMemcpyDeviceToHost));

        const uint32_t nb = (B + GE_BLOCK - 1) / GE_BLOCK;
        auto run_variant = [&](const char *name, uint32_t n_lds, uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1, uint32_t cell_mask, uint32_t level_mask, double dense_w, int order_mode) {
            LabCfg cfg = base;
            cfg.n_lds = n_lds; cfg.lds_a0 = a0; cfg.lds_a1 = a1; cfg.lds_b0 = b0; cfg.lds_b1 = b1; cfg.cell_mask = cell_mask; cfg.nb = nb;
            cfg.pts_per_lds = n_lds ? ((B + n_lds - 1) / n_lds + 255) / 256 * 256 : 0;
            GridLevels l2 = lv;
            // TA work list: levels in level_mask that no LDS phase covers; order_mode 0 = production interleave (coarse/fine alternate), 1 = hashed first then dense,
            // 2 = each XCD slice starts with hashed levels and ends with dense ones (round-robin deal)
            std::vector<uint8_t> ta;
            for (uint32_t l = 0; l < L; l++) {
                const bool in_lds = n_lds && ((l >= a0 && l < a1) || (l >= b0 && l < b1));
                if (!in_lds && ((level_mask >> l) & 1u)) ta.push_back((uint8_t)l);
            }
            std::vector<uint8_t> ord;
            if (order_mode == 0) { size_t lo = 0, hi = ta.size(); for (size_t i = 0; i < ta.size(); i++) ord.push_back((i & 1) ? ta[--hi] : ta[lo++]); }
            else if (order_mode == 1) { for (size_t i = ta.size(); i-- > 0;) ord.push_back(ta[i]); }
            else ord = ta;
            cfg.ta_levels = (uint32_t)ord.size();
            for (size_t i = 0; i < ord.size(); i++) l2.order[i] = ord[i];
            uint32_t longest = cfg.ta_levels ? ge_balance(l2, cfg.ta_levels, nb, 3, 0, false, dense_w) : 0;
            uint32_t lds_bytes = 0;
            if (n_lds) {
                uint32_t ea = a1 > a0 ? (lv.offset[a1 - 1] + lv.size[a1 - 1] - lv.offset[a0]) * 4 : 0, eb = b1 > b0 ? (lv.offset[b1 - 1] + lv.size[b1 - 1] - lv.offset[b0]) * 4 : 0;
                lds_bytes = ea > eb ? ea : eb;
            }
            CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_lab), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            const dim3 grid(n_lds + CN_NXCD * longest);
            CK(hipMemset(d_out, 0xFF, (size_t)L * B * 4));
            auto go = [&] { hipLaunchKernelGGL(k_lab, grid, dim3(GE_BLOCK), lds_bytes, 0, d_in, (const __half *)d_tab, d_cells, l2, cfg, d_out, B, 0u, B); };
            go(); CK(hipDeviceSynchronize()); CK(hipGetLastError());
            CK(hipMemcpy(got.data(), d_out, (size_t)L * B * 4, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (uint32_t l = 0; l < L; l++) if ((level_mask >> l) & 1u) for (uint32_t b = 0; b < B; b++) bad += got[(size_t)l * B + b] != ref[(size_t)l * B + b];
            const float t = time_it(go);
            printf("%-58s %7.1f us  lds %6u B  grid %6u  mismatches %zu\n", name, t, lds_bytes, grid.x, bad);
        };
        const uint32_t ALL = 0xFFFF, HASHED = 0xFFE0, DENSE = 0x001F;
        run_variant("lab: production structure, all levels", 0, 0, 0, 0, 0, 0, ALL, 0.7, 0);
        run_variant("lab: hashed levels only (5-15)", 0, 0, 0, 0, 0, 0, HASHED, 0.7, 0);
        run_variant("lab: dense levels only (0-4), vertex layout", 0, 0, 0, 0, 0, 0, DENSE, 0.7, 0);
        run_variant("lab: dense levels only (0-4), cell-major", 0, 0, 0, 0, 0, DENSE, DENSE, 0.7, 0);
        run_variant("lab: levels 3-4 only, vertex layout", 0, 0, 0, 0, 0, 0, 0x18, 0.7, 0);
        run_variant("lab: levels 3-4 only, cell-major", 0, 0, 0, 0, 0, 0x18, 0x18, 0.7, 0);
        run_variant("lab: levels 0-1 only, LDS (256 WGs)", 256, 0, 2, 0, 0, 0, 0x3, 0.7, 0);
        run_variant("lab: levels 0-2 only, LDS two phases (256 WGs)", 256, 0, 2, 2, 3, 0, 0x7, 0.7, 0);
        run_variant("lab: levels 0-2 only, LDS two phases (512 WGs)", 512, 0, 2, 2, 3, 0, 0x7, 0.7, 0);
        run_variant("lab: all; cell-major 2-4", 0, 0, 0, 0, 0, 0x1C, ALL, 0.7, 0);
        run_variant("lab: all; cell-major 0-4", 0, 0, 0, 0, 0, 0x1F, ALL, 0.7, 0);
        run_variant("lab: all; cell-major 0-4, dense weight 0.4", 0, 0, 0, 0, 0, 0x1F, ALL, 0.4, 0);
        run_variant("lab: all; LDS 0-1 (256 WGs), vertex 2-4", 256, 0, 2, 0, 0, 0, ALL, 0.7, 0);
        run_variant("lab: all; LDS 0-1 (256 WGs), cell-major 2-4", 256, 0, 2, 0, 0, 0x1C, ALL, 0.7, 0);
        run_variant("lab: all; LDS 0-1 (256), cell 2-4, dense w 0.4", 256, 0, 2, 0, 0, 0x1C, ALL, 0.4, 0);
        run_variant("lab: all; LDS 0-1 + 2 (256 WGs), cell-major 3-4", 256, 0, 2, 2, 3, 0x18, ALL, 0.7, 0);
        run_variant("lab: all; LDS 0-1 + 2 (256), cell 3-4, dense w 0.4", 256, 0, 2, 2, 3, 0x18, ALL, 0.4, 0);
        run_variant("lab: all; LDS 0-1 + 2 (512 WGs), cell-major 3-4", 512, 0, 2, 2, 3, 0x18, ALL, 0.7, 0);
        run_variant("lab: all; LDS 0-1 + 2 (256), cell 3-4, hashed first", 256, 0, 2, 2, 3, 0x18, ALL, 0.7, 1);
        run_variant("lab: all; LDS 0-1 + 2 (256), cell 3-4, plain order", 256, 0, 2, 2, 3, 0x18, ALL, 0.7, 2);
        {
            auto probe = [&](const char *name, auto kern, uint32_t level_mask) {
                GridLevels l2 = lv;
                std::vector<uint8_t> ta;
                for (uint32_t l = 0; l < L; l++) if ((level_mask >> l) & 1u) ta.push_back((uint8_t)l);
                for (size_t i = 0; i < ta.size(); i++) l2.order[i] = ta[i];
                const uint32_t nt = (uint32_t)ta.size();
                const uint32_t longest = ge_balance(l2, nt, nb, 3, 0, false, 0.7);
                auto go = [&] { hipLaunchKernelGGL(kern, dim3(CN_NXCD * longest), dim3(GE_BLOCK), 0, 0, d_in, (const __half *)d_tab, l2, d_out, B, nb, nt, B, 64u); };
                go(); CK(hipDeviceSynchronize()); CK(hipGetLastError());
                CK(hipMemcpy(got.data(), d_out, (size_t)L * B * 4, hipMemcpyDeviceToHost));
                size_t bad = 0;
                for (uint32_t l = 0; l < L; l++) if ((level_mask >> l) & 1u) for (uint32_t b = 0; b < B; b++) bad += got[(size_t)l * B + b] != ref[(size_t)l * B + b];
                printf("%-58s %7.1f us  (%5.1f per level)  mismatches %zu\n", name, time_it(go), time_it(go) / nt, bad);
            };
            probe("probe: hashed 5-15, as production", k_lab_hash<0, 1>, HASHED);
            probe("probe: hashed 5-15, no fallback loads", k_lab_hash<1, 1>, HASHED);
            probe("probe: hashed 5-15, unaligned windows (83 % cover)", k_lab_hash<32, 1>, HASHED);
            probe("probe: levels 10-15, as production", k_lab_hash<0, 1>, 0xFC00);
            probe("probe: levels 10-15, unaligned windows", k_lab_hash<32, 1>, 0xFC00);
            probe("probe: hashed 5-15, two of four windows", k_lab_hash<3, 1>, HASHED);
            probe("probe: hashed 5-15, non-temporal table loads", k_lab_hash<4, 1>, HASHED);
            probe("probe: hashed 5-15, L1-bypassing (sc1) table loads", k_lab_hash<8, 1>, HASHED);
            probe("probe: hashed 5-15, non-temporal coords / outputs", k_lab_hash<16, 1>, HASHED);
            probe("probe: hashed 5-15, nt table + nt coords / outputs", k_lab_hash<20, 1>, HASHED);
            probe("probe: hashed 5-15, 4 waves per SIMD", k_lab_hash<0, 4>, HASHED);
            for (uint32_t l = 5; l < 16; l++) { char nm[64]; snprintf(nm, sizeof nm, "probe: level %u alone", l); probe(nm, k_lab_hash<0, 1>, 1u << l); }
            if (!fine) {
                for (uint32_t patch : {1u, 4u, 8u}) {
                    Scene s2 = make_samples(128, 64, false, patch);
                    CK(hipMemcpy(d_in, s2.unit.data(), (size_t)B * 12, hipMemcpyHostToDevice));
                    prod(16); CK(hipDeviceSynchronize());
                    CK(hipMemcpy(ref.data(), d_ref, (size_t)L * B * 4, hipMemcpyDeviceToHost));
                    char nm[96];
                    snprintf(nm, sizeof nm, "tile: ray order patch %u, WG = 4 rays x 64 samples (linear)", patch); probe(nm, k_lab_hash<0, 1, 0>, HASHED);
                    snprintf(nm, sizeof nm, "tile: ray order patch %u, WG = 16 rays x 16 samples", patch); probe(nm, k_lab_hash<0, 1, 16>, HASHED);
                    snprintf(nm, sizeof nm, "tile: ray order patch %u, WG = 64 rays x 4 samples", patch); probe(nm, k_lab_hash<0, 1, 64>, HASHED);
                    snprintf(nm, sizeof nm, "tile: ray order patch %u, WG = 32 rays x 8 samples", patch); probe(nm, k_lab_hash<0, 1, 32>, HASHED);
                    snprintf(nm, sizeof nm, "tile: patch %u, 64 x 4, levels 5-9 only", patch); probe(nm, k_lab_hash<0, 1, 64>, 0x03E0);
                    snprintf(nm, sizeof nm, "tile: patch %u, 64 x 4, levels 10-15 only", patch); probe(nm, k_lab_hash<0, 1, 64>, 0xFC00);
                    snprintf(nm, sizeof nm, "tile: patch %u, linear, levels 10-15 only", patch); probe(nm, k_lab_hash<0, 1, 0>, 0xFC00);
                }
                CK(hipMemcpy(d_in, sc.unit.data(), (size_t)B * 12, hipMemcpyHostToDevice));
                prod(16); CK(hipDeviceSynchronize());
                CK(hipMemcpy(ref.data(), d_ref, (size_t)L * B * 4, hipMemcpyDeviceToHost));
            }
            probe("probe: levels 5-7", k_lab_hash<0, 1>, 0x00E0);
            probe("probe: levels 8-15", k_lab_hash<0, 1>, 0xFF00);
        }
        CK(hipFree(d_in)); CK(hipFree(d_ref)); CK(hipFree(d_out));
    }
    return 0;
}
