// GroupNorm statistics as 64-bit fixed point (CNERF_SD_GN_FRAC_BITS fractional bits): shared by sd_ops.hip (statistics / apply kernels)
// and sd_gemm.hip (statistics fused into the producing GEMM's epilogue).
#pragma once
#include "../../include/customnerf_sd.h"

// a float32 partial sum -> the fixed-point grid (round to nearest; saturating, NaN -> 0: the element that caused it still propagates
// through the element-wise apply pass)
__device__ __forceinline__ long long gn_fix(float v) {
    const double s = (double)v * (double)(1ll << CNERF_SD_GN_FRAC_BITS);
    const double c = fmin(fmax(s, -9.0e18), 9.0e18);
    return (c == c) ? __double2ll_rn(c) : 0ll;
}
__device__ __forceinline__ float gn_unfix(long long v) { return (float)((double)v * (1.0 / (double)(1ll << CNERF_SD_GN_FRAC_BITS))); }
// LDS / global accumulation: integer atomics are exact and commute
__device__ __forceinline__ void gn_add(long long *dst, float partial) { atomicAdd(reinterpret_cast<unsigned long long *>(dst), (unsigned long long)gn_fix(partial)); }
__device__ __forceinline__ void gn_add_fixed(long long *dst, long long v) {
    if (v != 0) atomicAdd(reinterpret_cast<unsigned long long *>(dst), (unsigned long long)v);
}
