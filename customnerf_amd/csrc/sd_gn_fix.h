// GroupNorm statistics as 64-bit fixed point (CNERF_SD_GN_FRAC_BITS fractional bits): shared by sd_ops.hip (statistics / apply kernels)
// and sd_gemm.hip (statistics fused into the producing GEMM's epilogue).
#pragma once
#include "../../include/customnerf_sd.h"

// Range and failure semantics.  Quantum 2^-20 (absolute), legitimate totals |sum| < 2^40 (2^60 in fixed point).  A sum of integers cannot
// carry an Inf / NaN, and a 64-bit sum wraps silently — so overflow is made STICKY instead of silent: a partial that is not finite, or whose magnitude
// reaches 2^30 (2^50 fixed: one thread's partial — no finite fp16 tensor of a healthy network gets there; the largest launch geometry adds < 2^13
// partials per group, i.e. < 2^63 before the check below sees it), replaces the sum by the poison pattern 0100...0 (atomic exchange: later adds, each
// < 2^50, cannot move bits 63..60 back to all-equal), and gn_unfix turns every value whose top four bits are not all equal — poisoned, or a total
// beyond +-2^60 — into NaN: the normalised tensor is NaN, the loss is NaN, GradScaler skips the step, exactly what float sums used to surface.
// Backward statistics are sums of dy-sized values: partials below 2^-21 (4.8e-7) flush to zero, so they are NOT scale-invariant — the trainers run
// them under the dynamic loss scale (65536 at start, gradients >= 1e-3 here); a static loss scale below ~64 on this path loses small-gradient groups.
#define GN_POISON 0x4000000000000000ll
__device__ __forceinline__ bool gn_is_poison(long long v) { const long long t = v >> 60; return t != 0 && t != -1; }
// a float32 partial sum -> the fixed-point grid (round to nearest), or GN_POISON
__device__ __forceinline__ long long gn_fix(float v) {
    const double s = (double)v * (double)(1ll << CNERF_SD_GN_FRAC_BITS);
    return (fabs(s) < 1125899906842624.0 /* 2^50 */) ? __double2ll_rn(s) : GN_POISON;          // (false for NaN)
}
__device__ __forceinline__ float gn_unfix(long long v) {
    return gn_is_poison(v) ? __builtin_nanf("") : (float)((double)v * (1.0 / (double)(1ll << CNERF_SD_GN_FRAC_BITS)));
}
// LDS / global accumulation: integer atomics are exact and commute
__device__ __forceinline__ void gn_add_fixed(long long *dst, long long v) {
    if (gn_is_poison(v)) atomicExch(reinterpret_cast<unsigned long long *>(dst), (unsigned long long)GN_POISON);
    else if (v != 0) atomicAdd(reinterpret_cast<unsigned long long *>(dst), (unsigned long long)v);
}
__device__ __forceinline__ void gn_add(long long *dst, float partial) { gn_add_fixed(dst, gn_fix(partial)); }
