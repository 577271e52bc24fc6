// Shared helpers for the gfx950 kernels.  Everything here is written for CDNA4 only (wave64, 256 CUs / 8 XCDs).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "../../include/customnerf_hip.h"

#define CN_WAVE 64
#define CN_NXCD 8

static inline int cn_launch_status() {
    hipError_t e = hipGetLastError();
    return (int)e;
}

static inline uint32_t cn_div_up(uint32_t a, uint32_t b) { return (a + b - 1) / b; }
static inline uint64_t cn_div_up64(uint64_t a, uint64_t b) { return (a + b - 1) / b; }

#define CN_STREAM(s) ((hipStream_t)(s))

// All float arithmetic in these kernels is compiled with -ffp-contract=off; fused multiply-adds are spelled out
// with cn_fma so that index-producing arithmetic is bit-identical to the oracle (oracle/raymarching_ref.c).
__device__ __forceinline__ float cn_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ float cn_clamp(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }

// wave-level helpers (wave64)
__device__ __forceinline__ uint32_t cn_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

template <typename T>
__device__ __forceinline__ T cn_wave_incl_scan(T v) {   // inclusive prefix sum across the 64 lanes
    const uint32_t lane = cn_lane();
#pragma unroll
    for (int off = 1; off < CN_WAVE; off <<= 1) {
        T n = __shfl_up(v, off, CN_WAVE);
        if ((int)lane >= off) v += n;
    }
    return v;
}

template <typename T>
__device__ __forceinline__ T cn_wave_sum(T v) {
#pragma unroll
    for (int off = CN_WAVE / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, CN_WAVE);
    return v;
}
