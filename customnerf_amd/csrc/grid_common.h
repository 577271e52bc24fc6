// Shared definitions of the grid-encoder kernels (gridencoder.hip, gridencoder_binned.hip).
#pragma once
#include "common.h"
#include <math.h>
#include <type_traits>

typedef _Float16 cn_gf_h2 __attribute__((ext_vector_type(2)));

#define GE_MAX_LEVELS 32
#define GE_BLOCK 256

struct GridLevels {
    uint32_t offset[GE_MAX_LEVELS];       // first entry of the level's table
    uint32_t size[GE_MAX_LEVELS];         // entries in the level's table (hashmap_size)
    uint32_t resolution[GE_MAX_LEVELS];
    float scale[GE_MAX_LEVELS];
    uint8_t order[GE_MAX_LEVELS];         // work-list position -> level (coarse/fine interleave)
    uint32_t xcd_first[CN_NXCD + 1];      // swizzle 2: first work item of each XCD's slice of the level-major list (cost-balanced)
};

// one sample's D coordinates in one load (global_load_dwordx2 / x3 / x4: 4-byte alignment is enough) instead of D strided dword loads
template <int D>
struct GeCoords { float v[D]; };
template <int D>
__device__ __forceinline__ void ge_load_coords(const float *__restrict__ inputs, size_t b, float (&in)[D]) {
    const GeCoords<D> c = reinterpret_cast<const GeCoords<D> *>(inputs)[b];
#pragma unroll
    for (int d = 0; d < D; d++) in[d] = c.v[d];
}

template <typename T, int C>
struct alignas(sizeof(T) * C) FeatVec {
    T v[C];
};

__device__ __forceinline__ float ge_to_float(float x) { return x; }
__device__ __forceinline__ float ge_to_float(__half x) { return __half2float(x); }
template <typename T> __device__ __forceinline__ T ge_from_float(float x);
template <> __device__ __forceinline__ float ge_from_float<float>(float x) { return x; }
template <> __device__ __forceinline__ __half ge_from_float<__half>(float x) { return __float2half_rn(x); }

// acc += w * g with the accumulator type of the reference (`scalar_t results[C]`): float -> one fma;
// half -> product rounded to half, sum rounded to half.
__device__ __forceinline__ void ge_accum(float &acc, float w, float g) { acc = cn_fma(w, g, acc); }
// The empty asm keeps the fp32 product a separately rounded value: without it the compiler folds
// cvt_f16(w * g) into one v_fma_mixlo_f16 (a single rounding), which differs from the reference's
// multiply-then-convert in rare double-rounding cases (1 fp16 ulp).
__device__ __forceinline__ float ge_opaque(float x) {
    asm("" : "+v"(x));
    return x;
}
// The half + half sum is the native v_add_f16: float32 holds the sum of two binary16 values exactly enough (24 >= 2 * 11 + 2 bits)
// that "add in float32, round to half" and the correctly rounded half add are the same function.
__device__ __forceinline__ void ge_accum(__half &acc, float w, __half g) {
    const float prod = ge_opaque(w * __half2float(g));
    acc = __hadd(acc, __float2half_rn(prod));
}
// two channels at once (v_pk_add_f16)
__device__ __forceinline__ void ge_accum2(__half2 &acc, float w, __half2 g) {
    const float2 gf = __half22float2(g);
    const float p0 = ge_opaque(w * gf.x), p1 = ge_opaque(w * gf.y);
    acc = __hadd2(acc, __floats2half2_rn(p0, p1));
}

__device__ __forceinline__ float ge_smoothstep(float v) { return v * v * (3.0f - 2.0f * v); }
__device__ __forceinline__ float ge_smoothstep_derivative(float v) { return 6 * v * (1.0f - v); }

template <int D>
__device__ __forceinline__ uint32_t ge_fast_hash(const uint32_t (&p)[D]) {
    constexpr uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < D; ++i) r ^= p[i] * primes[i];
    return r;
}

// entry index (not yet multiplied by C) of a grid vertex
template <int D>
__device__ __forceinline__ uint32_t ge_index(uint32_t gridtype, bool align_corners, uint32_t hashmap_size, uint32_t resolution,
                                             const uint32_t (&p)[D]) {
    uint32_t stride = 1, index = 0;
    const uint32_t step = align_corners ? resolution : (resolution + 1);
#pragma unroll
    for (int d = 0; d < D; d++) {
        if (stride <= hashmap_size) {
            index += p[d] * stride;
            stride *= step;
        }
    }
    if (gridtype == 0 && stride > hashmap_size) index = ge_fast_hash<D>(p);
    return index % hashmap_size;
}

// Level-uniform shortcuts of ge_index (same results, bit for bit).  The generic form costs a 32-bit remainder per corner (~30 VALU
// instructions, eight times per sample and level); which branch it takes depends on the level only:
//   GE_MODE_DENSE   every dimension enters the stride product and the product fits the table: index = sum p[d] * step^d < size, no wrap
//                   (align_corners: the boundary corner wraps once, see ge_index_m)
//   GE_MODE_HASH2   hashed level whose table size is a power of two: index = hash & (size - 1)
//   GE_MODE_GENERIC anything else (tiled levels that wrap, odd table sizes)
#define GE_MODE_GENERIC 0
#define GE_MODE_DENSE 1
#define GE_MODE_HASH2 2
#define GE_MODE_TILED2 3        // tiled level that wraps, power-of-two table: the strided sum of ge_index, masked instead of divided
template <int D>
__device__ __forceinline__ int ge_level_mode(uint32_t gridtype, bool align_corners, uint32_t hashmap_size, uint32_t resolution) {
    uint32_t stride = 1;
    bool all = true;
    const uint32_t step = align_corners ? resolution : (resolution + 1);
#pragma unroll
    for (int d = 0; d < D; d++) {
        if (stride <= hashmap_size) stride *= step;
        else all = false;
    }
    if (all && stride <= hashmap_size) return GE_MODE_DENSE;
    if (gridtype == 0 && stride > hashmap_size && (hashmap_size & (hashmap_size - 1)) == 0) return GE_MODE_HASH2;
    if (gridtype == 1 && (hashmap_size & (hashmap_size - 1)) == 0) return GE_MODE_TILED2;
    return GE_MODE_GENERIC;
}

template <int D, int MODE>
__device__ __forceinline__ uint32_t ge_index_m(uint32_t gridtype, bool align_corners, uint32_t hashmap_size, uint32_t resolution,
                                               const uint32_t (&p)[D]) {
    if constexpr (MODE == GE_MODE_HASH2) {
        return ge_fast_hash<D>(p) & (hashmap_size - 1);
    } else if constexpr (MODE == GE_MODE_DENSE) {
        const uint32_t step = align_corners ? resolution : (resolution + 1);
        uint32_t stride = step, index = p[0];
#pragma unroll
        for (int d = 1; d < D; d++) {
            index += p[d] * stride;
            stride *= step;
        }
        // align_corners: a point on the upper boundary has corner coordinate `resolution`, one past the last grid line, so the strided sum
        // can pass the level (by less than its size: step (step^D - 1) / (step - 1) < 2 step^D); the reference wraps it (`% hashmap_size`,
        // gridencoder.cu:66-84) — the corner's weight is zero there, but the entry must stay inside the level
        if (align_corners && index >= hashmap_size) index -= hashmap_size;
        return index;
    } else if constexpr (MODE == GE_MODE_TILED2) {
        uint32_t stride = 1, index = 0;
        const uint32_t step = align_corners ? resolution : (resolution + 1);
#pragma unroll
        for (int d = 0; d < D; d++) {
            if (stride <= hashmap_size) {                   // dimensions beyond the table's span are dropped, as in ge_index
                index += p[d] * stride;
                stride *= step;
            }
        }
        return index & (hashmap_size - 1);
    } else {
        return ge_index<D>(gridtype, align_corners, hashmap_size, resolution, p);
    }
}

// run f(std::integral_constant<int, MODE>) for the level's mode (the branch is wave-uniform: the level is a function of blockIdx)
template <typename F>
__device__ __forceinline__ void ge_dispatch_mode(int mode, F &&f) {
    if (mode == GE_MODE_HASH2) f(std::integral_constant<int, GE_MODE_HASH2>{});
    else if (mode == GE_MODE_DENSE) f(std::integral_constant<int, GE_MODE_DENSE>{});
    else if (mode == GE_MODE_TILED2) f(std::integral_constant<int, GE_MODE_TILED2>{});
    else f(std::integral_constant<int, GE_MODE_GENERIC>{});
}

// does ge_index take the hash branch on this level (gridtype hash only)?
template <int D>
__device__ __forceinline__ bool ge_is_hashed(bool align_corners, uint32_t hashmap_size, uint32_t resolution) {
    uint32_t stride = 1;
    const uint32_t step = align_corners ? resolution : (resolution + 1);
#pragma unroll
    for (int d = 0; d < D; d++)
        if (stride <= hashmap_size) stride *= step;
    return stride > hashmap_size;
}

// blockIdx -> (level, point block).  Swizzled: XCD x (= blockIdx % 8 as dispatched) walks a contiguous slice of the
// level-major work list, so at any moment it gathers from one or two tables that fit its own L2.
__device__ __forceinline__ bool ge_work_item(uint32_t nb, uint32_t n_levels, int swizzle, const GridLevels &lv, uint32_t &level, uint32_t &pb) {
    const uint32_t total = nb * n_levels;
    uint32_t w = blockIdx.x;
    if (swizzle == 2) {
        // cost-balanced slices: dense levels are cheap (their corners hit L1), hashed levels cost ~4 L2 requests per sample, and 16 levels
        // do not divide into 8 equal-cost pairs — the XCDs holding two hashed levels finished last with the others idle.  The host cuts
        // the list where the prefix COST crosses k/8 (ge_balance); slices are shorter where the items are dearer, surplus blocks exit.
        const uint32_t xcd = blockIdx.x % CN_NXCD, k = blockIdx.x / CN_NXCD;
        w = lv.xcd_first[xcd] + k;
        if (w >= lv.xcd_first[xcd + 1]) return false;
    } else if (swizzle) {
        // bijective chunking for any total: XCD x gets q+1 items if x < r else q (q = total/8, r = total%8)
        const uint32_t q = total / CN_NXCD, r = total % CN_NXCD;
        const uint32_t xcd = blockIdx.x % CN_NXCD, k = blockIdx.x / CN_NXCD;
        w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    if (w >= total) return false;
    level = lv.order[w / nb];
    pb = w % nb;
    return true;
}


// ------------------------------------------------------------------------------------------------ host side
static inline int ge_levels(const int32_t *offsets_host, uint32_t L, uint32_t n_levels, float S, uint32_t H, GridLevels &lv) {
    if (!offsets_host) return CNERF_ENULL;
    if (L == 0 || L > GE_MAX_LEVELS || n_levels > L) return CNERF_EINVAL;
    for (uint32_t l = 0; l < L; l++) {
        if (offsets_host[l + 1] <= offsets_host[l]) return CNERF_EINVAL;
        lv.offset[l] = (uint32_t)offsets_host[l];
        lv.size[l] = (uint32_t)(offsets_host[l + 1] - offsets_host[l]);
        const float scale = exp2f(l * S) * H - 1.0f;          // gridencoder.cu:138-139 (host libm, same as the oracle)
        lv.scale[l] = scale;
        lv.resolution[l] = (uint32_t)ceilf(scale) + 1;
    }
    // coarse/fine interleave: 0, n-1, 1, n-2, ... so each XCD's slice holds one cheap and one expensive level
    uint32_t lo = 0, hi = n_levels;
    for (uint32_t i = 0; i < n_levels; i++) lv.order[i] = (uint8_t)((i & 1) ? --hi : lo++);
    for (uint32_t x = 0; x <= CN_NXCD; x++) lv.xcd_first[x] = 0;
    return CNERF_OK;
}

// cost-balanced XCD slices of the level-major work list (nb blocks per level, list order lv.order): weight 1 for levels whose table is
// larger than what a wave's neighbouring samples keep re-hitting in L1 (hashed, or dense beyond `dense_entries`), `dense_w` otherwise.
// Returns the longest slice (blocks per XCD) — the launch needs 8 x that many workgroups.
static inline uint32_t ge_balance(GridLevels &lv, uint32_t n_levels, uint32_t nb, uint32_t D, uint32_t gridtype, bool align_corners, double dense_w) {
    double w[GE_MAX_LEVELS], total = 0;
    for (uint32_t i = 0; i < n_levels; i++) {
        const uint32_t l = lv.order[i];
        const uint64_t step = align_corners ? lv.resolution[l] : lv.resolution[l] + 1;
        uint64_t cells = 1;
        for (uint32_t d = 0; d < D; d++) cells *= step;
        const bool dense = cells <= lv.size[l];
        w[i] = dense ? dense_w : 1.0;
        (void)gridtype;
        total += w[i] * nb;
    }
    uint32_t longest = 0;
    lv.xcd_first[0] = 0;
    uint32_t pos = 0;                 // work item index
    double acc = 0;                   // cost before `pos`
    uint32_t i = 0, in_level = 0;     // current list position and blocks of it already consumed
    for (uint32_t x = 1; x <= CN_NXCD; x++) {
        const double target = total * x / CN_NXCD;
        while (i < n_levels) {
            const double left = (nb - in_level) * w[i];
            if (acc + left <= target + 1e-9) { acc += left; pos += nb - in_level; in_level = 0; i++; continue; }
            const uint32_t take = (uint32_t)((target - acc) / w[i]);
            acc += take * w[i]; pos += take; in_level += take;
            break;
        }
        if (x == CN_NXCD) pos = nb * n_levels;
        lv.xcd_first[x] = pos;
        if (pos - lv.xcd_first[x - 1] > longest) longest = pos - lv.xcd_first[x - 1];
    }
    return longest;
}

