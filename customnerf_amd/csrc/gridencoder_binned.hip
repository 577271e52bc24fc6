// Grid-encoder backward WITHOUT global float atomics (gfx950).
//
// Why: MI355X executes every global float atomic memory-side, at a flat ~21 G atomics/s whatever the scope,
// locality or table size (scratch/atomic_bench.hip, measured).  The straight scatter of the reference
// (gridencoder.cu:324-337: one atomicAdd per corner per channel) needs 16 levels x 8 corners x 2 channels = 256
// atomics per sample — 537 M for a 128x128x128-sample step, i.e. >25 ms at the hardware's ceiling.
//
// Instead the scatter is a two-sweep radix partition by destination, so that all additions happen in LDS:
//   sweep 1  k_bin_hist   per (level, 1024-point block): count the corner updates per 8192-entry table chunk
//            k_bin_scan   exclusive scans -> every (chunk, block) pair owns a contiguous record range
//            k_bin_emit   re-derive the corners and write one 8/12-byte record {entry-in-chunk, w*g} per corner
//   sweep 2  k_bin_accum  one workgroup per chunk (segment): zero a 64 KiB LDS image of the chunk, add its records
//                         with LDS atomics, then add the image into the float32 gradient table with coalesced plain
//                         read-modify-writes (one owner per chunk => no global atomics); oversized bins (the small
//                         dense levels) are split and only those flush with atomics.
//                         LDS float atomics (ds_add_f32) run ~8x slower than LDS integer atomics on gfx950
//                         (scratch/lds_atomic_bench.hip: 101 vs 866 G records/s), so fp16 records are summed as 64-bit
//                         fixed point with 24 fractional bits: every binary16 value is an exact multiple of 2^-24, so
//                         the chunk sums are EXACT and independent of record order (deterministic), rounded once to
//                         float32 at the flush.  float32 records keep ds_add_f32 (the parity path).
// Traffic: 8 corners x 16 levels x 8 B = 1 KiB/sample written + read once — about the algorithmic RMW bytes of the
// scatter itself (2048 B/sample fp32), all of it coalesced or L2-merged.
// D = 3, C = 2 only (the configuration CustomNeRF uses); other shapes take the atomic kernel in gridencoder.hip.
//
// Forms in this file (round 6): the FIRST form described above (k_bin_*: histogram + scans + one record per corner) serves float32 records and the
// fp16 shapes the third form does not take (more than 512 bins per level, hashed levels of odd size or fewer than 32 entries); the THIRD form (k_bin3_*: no histogram,
// 8-byte pair records, block-local counting, fixed-capacity bin regions with spill) serves everything CustomNeRF runs — the benchmark table and the
// reference field's own 2^21-entry table.  The second form (histogram-driven pair records, rounds 2-4) is gone; its record format lives on.
#include "grid_common.h"
#include <vector>
#include <algorithm>
#include <cstdio>

#define BN_CHUNK_LOG2 12
#define BN_CHUNK (1u << BN_CHUNK_LOG2)            // table entries per bin  (x 2 ch x 8 B fixed point = 64 KiB LDS)
#define BN_THREADS 256
#define BN_PPT 4                                   // points per thread in the hist / emit sweeps
#define BN_PTS (BN_THREADS * BN_PPT)               // points per block
#define BN_MAX_CHUNKS 512                          // 2^21-entry level / 4096 (the reference's bear table); the staged emit serves <= B2S_MAX_CHUNKS
#define BN_SEG (1u << 18)                          // records per accumulate workgroup (split unit for oversized bins)

struct BinPlan {
    uint32_t bin_first[GE_MAX_LEVELS + 1];         // first bin of each level (prefix over chunks per level)
    uint32_t nb;                                   // point blocks per level
    uint32_t total_bins;
};

// workspace layout (all uint32 unless stated), offsets in bytes are computed by bn_layout()
struct BinWs {
    uint32_t *hist;        // [total_bins][nb]   counts, then (after scan) exclusive offsets inside the bin
    uint32_t *bin_base;    // [total_bins + 1]   first record of each bin
    uint32_t *seg_first;   // [total_bins + 1]   first accumulate-workgroup of each bin
    void *records;
};

template <typename T> struct BinRec;
template <> struct alignas(8) BinRec<__half> { uint32_t idx; __half2 v; };
template <> struct BinRec<float> { uint32_t idx; float v0, v1; };

__device__ __forceinline__ void bn_corners(const float (&in)[3], const GridLevels &lv, uint32_t level, uint32_t gridtype, bool align_corners,
                                           uint32_t interp, uint32_t (&index)[8], float (&wgt)[8]) {
    const uint32_t hashmap_size = lv.size[level];
    const float scale = lv.scale[level];
    const uint32_t resolution = lv.resolution[level];
    float pos[3];
    uint32_t pos_grid[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        pos[d] = cn_fma(in[d], scale, align_corners ? 0.0f : 0.5f);
        pos_grid[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)pos_grid[d];
        if (interp == 1) pos[d] = ge_smoothstep(pos[d]);
    }
#pragma unroll
    for (int idx = 0; idx < 8; idx++) {
        float w = 1;
        uint32_t pgl[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            if ((idx & (1 << d)) == 0) { w *= 1 - pos[d]; pgl[d] = pos_grid[d]; }
            else { w *= pos[d]; pgl[d] = pos_grid[d] + 1; }
        }
        wgt[idx] = w;
        index[idx] = ge_index<3>(gridtype, align_corners, hashmap_size, resolution, pgl);
    }
}

__device__ __forceinline__ bool bn_load_point(const float *__restrict__ inputs, uint32_t b, uint32_t B, float (&in)[3]) {
    if (b >= B) return false;
    bool ok = true;
    ge_load_coords<3>(inputs, b, in);
#pragma unroll
    for (int d = 0; d < 3; d++) ok = ok && !(in[d] < 0 || in[d] > 1);
    return ok;
}

// ---- sweep 1a: histogram of corner updates per chunk, one block = BN_PTS points of one level
__global__ void __launch_bounds__(BN_THREADS) k_bin_hist(const float *__restrict__ inputs, const GridLevels lv, const BinPlan plan,
                                                         uint32_t *__restrict__ hist, uint32_t B, uint32_t gridtype, int align_corners,
                                                         uint32_t interp) {
    __shared__ uint32_t cnt[BN_MAX_CHUNKS];
    const uint32_t level = blockIdx.x / plan.nb, pb = blockIdx.x % plan.nb;
    const uint32_t nch = plan.bin_first[level + 1] - plan.bin_first[level];
    for (uint32_t c = threadIdx.x; c < BN_MAX_CHUNKS; c += BN_THREADS) cnt[c] = 0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < BN_PPT; i++) {
        const uint32_t b = pb * BN_PTS + i * BN_THREADS + threadIdx.x;
        float in[3];
        if (!bn_load_point(inputs, b, B, in)) continue;
        uint32_t index[8];
        float wgt[8];
        bn_corners(in, lv, level, gridtype, align_corners, interp, index, wgt);
        // corners 2q / 2q+1 differ in x only: their entries almost always share a chunk (dense: neighbours; hashed: the x term of
        // the hash is the identity, x -> x+1 flips low bits) -> one counter update per pair
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t c0 = index[2 * q] >> BN_CHUNK_LOG2, c1 = index[2 * q + 1] >> BN_CHUNK_LOG2;
            atomicAdd(&cnt[c0], c0 == c1 ? 2u : 1u);
            if (c0 != c1) atomicAdd(&cnt[c1], 1u);
        }
    }
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < nch; c += BN_THREADS) hist[(size_t)(plan.bin_first[level] + c) * plan.nb + pb] = cnt[c];
}

// ---- sweep 1b: per bin, exclusive scan over the point blocks (in place) and the bin total.  256-thread workgroups: this runs beside the
// forward pass on a second stream, where a 1024-thread workgroup waits a long time for sixteen free wave slots on one CU.
#define BN_SCAN_THREADS 256
__global__ void __launch_bounds__(BN_SCAN_THREADS) k_bin_scan_blocks(uint32_t *__restrict__ hist, uint32_t *__restrict__ bin_total, uint32_t nb) {
    constexpr uint32_t NW = BN_SCAN_THREADS / 64;
    __shared__ uint32_t wave_tot[NW];
    __shared__ uint32_t carry;
    uint32_t *h = hist + (size_t)blockIdx.x * nb;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint32_t start = 0; start < nb; start += BN_SCAN_THREADS * 4) {
        // four consecutive counts per thread
        const uint32_t i = start + tid * 4;
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = i + k < nb ? h[i + k] : 0;
        const uint32_t mine = v[0] + v[1] + v[2] + v[3];
        const uint32_t incl = cn_wave_incl_scan(mine);
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        uint32_t wbase = 0, tot = 0;
#pragma unroll
        for (uint32_t w = 0; w < NW; w++) {
            const uint32_t t = wave_tot[w];
            if (w < wave) wbase += t;
            tot += t;
        }
        const uint32_t base = carry;
        uint32_t run = base + wbase + incl - mine;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (i + k < nb) h[i + k] = run;
            run += v[k];
        }
        __syncthreads();
        if (tid == 0) carry = base + tot;
        __syncthreads();
    }
    if (tid == 0) bin_total[blockIdx.x] = carry;
}

// ---- sweep 1c: scan over bins: record base per bin and accumulate-workgroup base per bin (one workgroup)
// bin_total and bin_base may alias (in-place): every thread reads its element before anyone overwrites it
__global__ void __launch_bounds__(1024) k_bin_scan_bins(const uint32_t *bin_total, uint32_t *bin_base,
                                                        uint32_t *__restrict__ seg_first, uint32_t total_bins, uint32_t seg_records,
                                                        uint32_t *__restrict__ seg_bin, uint32_t *__restrict__ split_list = nullptr, int min_one = 0,
                                                        const float *__restrict__ adam_state = nullptr, float adam_lr = 0.0f, float adam_beta1 = 0.0f,
                                                        float adam_beta2 = 0.0f, float adam_extra_inv = 0.0f, float *__restrict__ adam_const = nullptr) {
    __shared__ uint32_t wt_r[16], wt_s[16];
    __shared__ uint32_t carry_r, carry_s, n_split;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (adam_const && tid == 1023) {
        // the optimiser step carried by the accumulate (B3Adam): k_adam_scaled's prologue, evaluated ONCE per backward pass here (two double-precision
        // powers) — the scaler's found_inf is final: field backward and emit, its only producers on this path, precede this launch
        const double step = (double)adam_state[3] + 1.0;
        const double bc1 = 1.0 - pow((double)adam_beta1, step), bc2 = 1.0 - pow((double)adam_beta2, step);
        adam_const[0] = adam_extra_inv / adam_state[0];
        adam_const[1] = (float)((double)adam_lr / bc1);
        adam_const[2] = (float)(1.0 / sqrt(bc2));
        adam_const[3] = adam_state[2] != 0.0f ? 1.0f : 0.0f;
    }
    if (tid == 0) { carry_r = 0; carry_s = 0; n_split = 0; }
    __syncthreads();
    for (uint32_t start = 0; start < total_bins; start += 1024) {
        const uint32_t i = start + tid;
        const uint32_t r = i < total_bins ? bin_total[i] : 0;
        // (segments rounded to nearest with the last one taking the remainder — no near-empty second segment for a bin a few records over the
        // segment size — were measured: nothing at random initialisation, accumulate 181 -> 200 us on a fitted field, whose crowded bins then
        // run 1.5 segments in one workgroup: profiles/r06_reduce_split_ab.txt)
        uint32_t s = (r + seg_records - 1) / seg_records;
        if (min_one && i < total_bins && s == 0) s = 1;              // fused optimiser step: a bin without records still has to visit its entries
        const uint32_t ir = cn_wave_incl_scan(r), is = cn_wave_incl_scan(s);
        if (lane == 63) { wt_r[wave] = ir; wt_s[wave] = is; }
        __syncthreads();
        uint32_t br = 0, bs = 0, tr = 0, ts = 0;
#pragma unroll
        for (uint32_t w = 0; w < 16; w++) {
            if (w < wave) { br += wt_r[w]; bs += wt_s[w]; }
            tr += wt_r[w]; ts += wt_s[w];
        }
        const uint32_t cr = carry_r, cs = carry_s;
        if (i < total_bins) {
            bin_base[i] = cr + br + ir - r;
            seg_first[i] = cs + bs + is - s;
            if (seg_bin)                                            // segment -> bin map: the accumulate workgroups look their bin up in one load
                for (uint32_t k = 0; k < s; k++) seg_bin[cs + bs + is - s + k] = i;
            if (split_list && s > 1) split_list[1 + atomicAdd(&n_split, 1u)] = i;      // the bins k_bin3_reduce_split has to visit (any order)
        }
        __syncthreads();
        if (tid == 0) { carry_r = cr + tr; carry_s = cs + ts; }
        __syncthreads();
    }
    if (tid == 0) {
        bin_base[total_bins] = carry_r; seg_first[total_bins] = carry_s;
        if (split_list) split_list[0] = n_split;
    }
}

// ---- sweep 1d: write the records
template <typename T>
__global__ void __launch_bounds__(BN_THREADS) k_bin_emit(const T *__restrict__ grad, const float *__restrict__ inputs, const GridLevels lv,
                                                         const BinPlan plan, const uint32_t *__restrict__ hist,
                                                         const uint32_t *__restrict__ bin_base, BinRec<T> *__restrict__ records, uint32_t B,
                                                         uint32_t gridtype, int align_corners, uint32_t interp, float *__restrict__ grad_grid,
                                                         float *__restrict__ found_inf) {
    __shared__ uint32_t cursor[BN_MAX_CHUNKS];
    const uint32_t level = blockIdx.x / plan.nb, pb = blockIdx.x % plan.nb;
    const uint32_t nch = plan.bin_first[level + 1] - plan.bin_first[level];
    for (uint32_t c = threadIdx.x; c < nch; c += BN_THREADS) {
        const uint32_t bin = plan.bin_first[level] + c;
        cursor[c] = bin_base[bin] + hist[(size_t)bin * plan.nb + pb];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < BN_PPT; i++) {
        const uint32_t b = pb * BN_PTS + i * BN_THREADS + threadIdx.x;
        float in[3];
        if (!bn_load_point(inputs, b, B, in)) continue;
        uint32_t index[8];
        float wgt[8];
        bn_corners(in, lv, level, gridtype, align_corners, interp, index, wgt);
        using Vec = FeatVec<T, 2>;
        const Vec g = reinterpret_cast<const Vec *>(grad)[(size_t)level * B + b];
        const float g0 = ge_to_float(g.v[0]), g1 = ge_to_float(g.v[1]);
        if constexpr (sizeof(T) == 2) {                      // fixed-point sums: see b2_poison below
            if (!(fabsf(g0) <= 65504.0f) || !(fabsf(g1) <= 65504.0f)) {
                grad_grid[((size_t)lv.offset[level] + index[0]) * 2] = __builtin_nanf("");
                if (found_inf) *found_inf = 1.0f;
            }
        } else {
            if ((!(fabsf(g0) <= 3.4e38f) || !(fabsf(g1) <= 3.4e38f)) && found_inf) *found_inf = 1.0f;      // (float32 records carry the value itself)
        }
        // one cursor update and one double-width store per x-pair of corners (see k_bin_hist); record order inside a bin is
        // irrelevant: fp16 sums are exact fixed point, fp32 sums are order-dependent at rounding level only
        struct alignas(8) RecPair { BinRec<T> a, b; };
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t i0 = index[2 * q], i1 = index[2 * q + 1];
            const uint32_t c0 = i0 >> BN_CHUNK_LOG2, c1 = i1 >> BN_CHUNK_LOG2;
            RecPair rp;
            rp.a.idx = i0 & (BN_CHUNK - 1);
            rp.b.idx = i1 & (BN_CHUNK - 1);
            if constexpr (sizeof(T) == 2) {
                rp.a.v = __floats2half2_rn(wgt[2 * q] * g0, wgt[2 * q] * g1);     // as gridencoder.cu:328
                rp.b.v = __floats2half2_rn(wgt[2 * q + 1] * g0, wgt[2 * q + 1] * g1);
            } else {
                rp.a.v0 = wgt[2 * q] * g0; rp.a.v1 = wgt[2 * q] * g1;
                rp.b.v0 = wgt[2 * q + 1] * g0; rp.b.v1 = wgt[2 * q + 1] * g1;
            }
            const uint32_t pos0 = atomicAdd(&cursor[c0], c0 == c1 ? 2u : 1u);
            if (c0 == c1) {
                *reinterpret_cast<RecPair *>(records + pos0) = rp;
            } else {
                const uint32_t pos1 = atomicAdd(&cursor[c1], 1u);
                records[pos0] = rp.a;
                records[pos1] = rp.b;
            }
        }
    }
}

// ---- sweep 2: accumulate one bin segment in LDS and add it to the gradient table
#define BN_FIX_BITS 24
template <typename T> struct BinAcc;
template <> struct BinAcc<__half> { using type = long long; };          // exact fixed point
template <> struct BinAcc<float> { using type = float; };

template <typename T>
__device__ __forceinline__ void bn_add_record(typename BinAcc<T>::type *acc, const BinRec<T> &r) {
    if constexpr (sizeof(T) == 2) {
        const float2 f = __half22float2(r.v);
        // |f| <= 65504 and f is a multiple of 2^-24 (or 0): f * 2^24 is an exactly representable integer < 2^41
        const long long a = (long long)(f.x * 16777216.0f), b = (long long)(f.y * 16777216.0f);
        atomicAdd(reinterpret_cast<unsigned long long *>(&acc[r.idx * 2]), (unsigned long long)a);
        atomicAdd(reinterpret_cast<unsigned long long *>(&acc[r.idx * 2 + 1]), (unsigned long long)b);
    } else {
        unsafeAtomicAdd(&acc[r.idx * 2], r.v0);
        unsafeAtomicAdd(&acc[r.idx * 2 + 1], r.v1);
    }
}

template <typename T>
__device__ __forceinline__ float bn_acc_to_float(typename BinAcc<T>::type a) {
    if constexpr (sizeof(T) == 2) return (float)((double)a * (1.0 / 16777216.0));
    else return a;
}

template <typename T>
__global__ void __launch_bounds__(1024) k_bin_accum(const BinRec<T> *__restrict__ records, const uint32_t *__restrict__ bin_base,
                                                    const uint32_t *__restrict__ seg_first, const GridLevels lv, const BinPlan plan,
                                                    float *__restrict__ grad_grid) {
    using A = typename BinAcc<T>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char bn_lds[];   // [BN_CHUNK][2] accumulators, then one uint32 (one LDS object)
    A *acc = reinterpret_cast<A *>(bn_lds);
    uint32_t &s_bin = *reinterpret_cast<uint32_t *>(bn_lds + sizeof(A) * BN_CHUNK * 2);
    const uint32_t nseg_total = seg_first[plan.total_bins];
    if (blockIdx.x >= nseg_total) return;
    if (threadIdx.x == 0) {                                            // which bin does this workgroup serve: upper_bound(seg_first, blockIdx) - 1
        uint32_t lo = 0, hi = plan.total_bins;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (seg_first[mid + 1] <= blockIdx.x) lo = mid + 1; else hi = mid;
        }
        s_bin = lo;
    }
    for (uint32_t i = threadIdx.x; i < sizeof(A) * BN_CHUNK * 2 / 16; i += 1024) reinterpret_cast<uint4 *>(bn_lds)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const uint32_t bin = s_bin;
    const uint32_t seg = blockIdx.x - seg_first[bin], nseg = seg_first[bin + 1] - seg_first[bin];
    const uint32_t r0 = bin_base[bin], r1 = bin_base[bin + 1];
    const uint32_t begin = r0 + seg * BN_SEG, end = min(begin + BN_SEG, r1);
    constexpr int UNR = 4;
    uint32_t i = begin + threadIdx.x;
    for (; i + (UNR - 1) * 1024 < end; i += UNR * 1024) {
        BinRec<T> r[UNR];
#pragma unroll
        for (int u = 0; u < UNR; u++) r[u] = records[i + u * 1024];
#pragma unroll
        for (int u = 0; u < UNR; u++) bn_add_record<T>(acc, r[u]);
    }
    for (; i < end; i += 1024) bn_add_record<T>(acc, records[i]);
    __syncthreads();
    // locate the chunk in the table
    uint32_t level = 0;
    while (plan.bin_first[level + 1] <= bin) level++;
    const uint32_t chunk = bin - plan.bin_first[level];
    const uint32_t e0 = chunk << BN_CHUNK_LOG2;
    const uint32_t n_entries = min(BN_CHUNK, lv.size[level] - e0);
    float *__restrict__ dst = grad_grid + ((size_t)lv.offset[level] + e0) * 2;
    if (nseg == 1) {
        // sole owner of this chunk in this launch: coalesced plain read-modify-write (table sizes are multiples of 8 entries)
        for (uint32_t j = threadIdx.x; j < n_entries * 2 / 4; j += 1024) {
            float4 g = reinterpret_cast<float4 *>(dst)[j];
            g.x += bn_acc_to_float<T>(acc[j * 4]); g.y += bn_acc_to_float<T>(acc[j * 4 + 1]);
            g.z += bn_acc_to_float<T>(acc[j * 4 + 2]); g.w += bn_acc_to_float<T>(acc[j * 4 + 3]);
            reinterpret_cast<float4 *>(dst)[j] = g;
        }
    } else {
        for (uint32_t j = threadIdx.x; j < n_entries * 2; j += 1024) {
            const float a = bn_acc_to_float<T>(acc[j]);
            if (a != 0.0f) unsafeAtomicAdd(&dst[j], a);
        }
    }
}


// ================================================================================================ fp16 pair records (rounds 2-4: the "second form"; its
// histogram-driven emit / accumulate kernels were removed in round 6 — what follows is the record format and the helpers the third form keeps)
// What the first form above pays for (rocprofv3, 2.1 M samples x 16 levels): 2.1 GB of records written by 16-byte stores that land in
// up to 128 bins per wave instruction, then read back — emit 1.0 ms + accumulate 0.55 ms, one after the other.  Measured on the side
// (scratch/mall_bench.hip, scratch/lds_atomic_bench.hip): (i) the LDS integer atomics are NOT the accumulate's limit (>= 1.6 T
// ds_add_u64/s against the 0.92 T/s it runs at) — its record stream is; (ii) scattered partial-line stores into a slab that is re-used
// pass after pass (and so stays in the 256 MiB Infinity Cache) cost 2.7x less than into a fresh 2 GB range.  Hence:
//   * ONE 8-byte record per x-PAIR of corners: {entry-in-chunk:12 | t:4 | fx:16, half2(w_yz * g)}.  The two entries of a pair differ by
//     i0 ^ i1 = 2^(t+1) - 1 (dense: i1 = i0 + 1; hashed: the x term of the hash is x itself, so i1 = i0 ^ x ^ (x+1) — either way a run of
//     low ones), and their weights are (1 - fx) * w_yz and fx * w_yz: the accumulate workgroup rebuilds both corner updates.  Pairs that
//     straddle a chunk border (1 in 4096 on dense levels) become two single records (t = 15).  Half the bytes of the first form.
// Measured and dropped: processing the levels in groups through two alternating slabs with the emit of group g+1 overlapping the accumulate
// of group g on a second stream (3.49 ms/step at 2 levels per group, 3.20 at 4, against 3.00 for the single pass: the passes underfill
// the chip and the cross-stream hand-offs cost more than the on-die re-read saves).
// Sums stay 64-bit fixed point with 24 fractional bits (order-independent, so bit-deterministic wherever one workgroup owns a chunk);
// the per-corner products are rounded to that grid instead of to binary16 (the reference rounds w*g to half, gridencoder.cu:328: the
// difference is below one half ulp of each product).
#define B2_SEG_MIN ((1u << 16) + (1u << 13))       // records per accumulate workgroup, see b2_seg()
#define B2_SINGLE 15u

struct Bin2Plan {
    uint32_t bin_first[GE_MAX_LEVELS + 1];         // first bin of each SLOT; slot i serves level lv.order[i] (coarse / fine interleaved)
    uint32_t nb;                                   // point blocks per level
    uint32_t total_bins;
};


// the four (y, z) corner pairs of a sample on one level: entries of the x and x+1 corner, the weight of the pair, the x fraction
__device__ __forceinline__ void b2_pairs(const float (&in)[3], const GridLevels &lv, uint32_t level, uint32_t gridtype, bool align_corners,
                                         uint32_t interp, uint32_t (&i0)[4], uint32_t (&i1)[4], float (&wyz)[4], float &fx) {
    const uint32_t hashmap_size = lv.size[level];
    const float scale = lv.scale[level];
    const uint32_t resolution = lv.resolution[level];
    float pos[3];
    uint32_t pos_grid[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        pos[d] = cn_fma(in[d], scale, align_corners ? 0.0f : 0.5f);
        pos_grid[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)pos_grid[d];
        if (interp == 1) pos[d] = ge_smoothstep(pos[d]);
    }
    fx = pos[0];
    const float oy = 1 - pos[1], oz = 1 - pos[2];
    wyz[0] = oy * oz; wyz[1] = pos[1] * oz; wyz[2] = oy * pos[2]; wyz[3] = pos[1] * pos[2];
    // Level-uniform index arithmetic, shared between the corners (ge_index_m called once per corner recomputes the y / z hash products
    // — sixteen 32-bit multiplies, each a quarter-rate instruction — and the strided sums eight times):
    const int mode = ge_level_mode<3>(gridtype, align_corners, hashmap_size, resolution);
    if (mode == GE_MODE_HASH2) {
        const uint32_t mask = hashmap_size - 1;
        const uint32_t hy0 = pos_grid[1] * 2654435761u, hy1 = hy0 + 2654435761u, hz0 = pos_grid[2] * 805459861u, hz1 = hz0 + 805459861u;
        const uint32_t hyz[4] = {hy0 ^ hz0, hy1 ^ hz0, hy0 ^ hz1, hy1 ^ hz1};
        const uint32_t x0 = pos_grid[0], x1 = x0 + 1;
#pragma unroll
        for (int q = 0; q < 4; q++) { i0[q] = (x0 ^ hyz[q]) & mask; i1[q] = (x1 ^ hyz[q]) & mask; }
    } else if (mode == GE_MODE_DENSE) {
        const uint32_t s1 = align_corners ? resolution : resolution + 1, s2 = s1 * s1;
        const uint32_t lin = pos_grid[0] + pos_grid[1] * s1 + pos_grid[2] * s2;
        i0[0] = lin; i0[1] = lin + s1; i0[2] = lin + s2; i0[3] = lin + s1 + s2;
#pragma unroll
        for (int q = 0; q < 4; q++) i1[q] = i0[q] + 1;
        if (align_corners) {                                    // boundary corners wrap into the level (ge_index_m<GE_MODE_DENSE>)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (i0[q] >= hashmap_size) i0[q] -= hashmap_size;
                if (i1[q] >= hashmap_size) i1[q] -= hashmap_size;
            }
        }
    } else {
        ge_dispatch_mode(mode, [&](auto mode_c) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t pgl[3] = {pos_grid[0], pos_grid[1] + (q & 1), pos_grid[2] + (q >> 1)};
            i0[q] = ge_index_m<3, decltype(mode_c)::value>(gridtype, align_corners, hashmap_size, resolution, pgl);
            pgl[0] += 1;
            i1[q] = ge_index_m<3, decltype(mode_c)::value>(gridtype, align_corners, hashmap_size, resolution, pgl);
        }
        });
    }
}

// can the pair travel as one record?  (same chunk, and the two entries differ by a run of low ones)
__device__ __forceinline__ bool b2_paired(uint32_t i0, uint32_t i1) {
    const uint32_t m = i0 ^ i1;
    return m != 0 && (m >> BN_CHUNK_LOG2) == 0 && (m & (m + 1)) == 0;
}

// LDS counter ticket, wave-aggregated when every active lane wants the same counter — the rule on the dense levels, where a wave's samples
// (neighbours on one ray) fall into one chunk and 64 same-address atomics would serialise; hashed levels take the per-lane atomic.
__device__ __forceinline__ uint32_t b2_ticket(uint32_t *counters, uint32_t c) {
    const uint64_t act = __ballot(1);
    const uint32_t c_lead = __builtin_amdgcn_readfirstlane(c);
    if (__ballot(c == c_lead) == act) {
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0u));
        uint32_t base = 0;
        if (rank == 0) base = atomicAdd(&counters[c_lead], (uint32_t)__popcll(act));
        return __builtin_amdgcn_readfirstlane(base) + rank;                         // the first active lane is the rank-0 lane
    }
    return atomicAdd(&counters[c], 1u);
}

// The fixed-point sums cannot carry an infinity or a NaN (llrint of one is an arbitrary finite pattern), but the loss scaler finds overflow by
// looking for exactly those in the gradients (the reference's half2 atomics propagate them: gridencoder.cu:324-337).  A non-finite incoming
// gradient therefore poisons one of its destination entries directly; the accumulate's read-modify-write keeps it non-finite.
__device__ __forceinline__ void b2_poison(float g0, float g1, float *__restrict__ grad_grid, const GridLevels &lv, uint32_t level, uint32_t entry,
                                          float *__restrict__ found_inf) {
    if (!(fabsf(g0) <= 65504.0f) || !(fabsf(g1) <= 65504.0f)) {
        grad_grid[((size_t)lv.offset[level] + entry) * 2] = __builtin_nanf("");
        if (found_inf) *found_inf = 1.0f;                                          // cnerf_scaler_watch (benign race: every writer stores the same value)
    }
}

// (The second form's emit kernels — direct and LDS-staged, both driven by a histogram pre-pass — were removed in round 6: the histogram-free
// third form below serves every table they served, up to 512 bins per level.)
#define B2S_MAX_CHUNKS 128                        // bins per level of a NARROW level of the third form (one-byte bin ids in its staging area)
// llrint(a) as a two's-complement 64-bit pattern, a = (value * 2^24) already in float32, |a| < 2^41.  The library's float -> int64
// conversion is a dozen vector instructions (no such conversion in hardware) and there are four of them per record: they were most of this
// kernel's instruction stream, and the kernel is issue-bound.  In double precision the classic "add 1.5 * 2^52" trick does it in one add:
// the sum's mantissa holds the integer, rounded to nearest-even by the add itself; the low word of the constant's bit pattern is zero, so
// only the high word needs the constant subtracted.  The 2^24 scale rides on the weight (a power of two commutes with the float32
// rounding of the product), so the values are those of __float2ll_rn((w * f) * 16777216.0f), bit for bit.
__device__ __forceinline__ unsigned long long b2_fix(float a) {
    const double y = (double)a + 6755399441055744.0;                                // 1.5 * 2^52
    return __builtin_bit_cast(unsigned long long, y) - 0x4338000000000000ull;
}
// the LDS image keeps the two channels in separate halves (acc[e], acc[BN_CHUNK + e]): random 8-byte atomics at a 16-byte stride reach
// only half of the bank pairs (scratch/lds_atomic_peak.hip: 2.6 T ds_add_u64/s interleaved, 3.8 T/s split)
__device__ __forceinline__ void b2_add(long long *acc, uint32_t e, float a, float b) {
    // |value| <= 65504: value * 2^24 < 2^41, rounded to the fixed-point grid
    atomicAdd(reinterpret_cast<unsigned long long *>(&acc[e]), b2_fix(a));
    atomicAdd(reinterpret_cast<unsigned long long *>(&acc[BN_CHUNK + e]), b2_fix(b));
}

__device__ __forceinline__ void b2_add_record(long long *acc, const uint2 r) {
    union { uint32_t u; __half2 h; } v;
    v.u = r.y;
    const float2 f = __half22float2(v.h);
    const uint32_t e = r.x & (BN_CHUNK - 1), t = (r.x >> 12) & 15u;
    if (t == B2_SINGLE) {
        b2_add(acc, e, f.x * 16777216.0f, f.y * 16777216.0f);
    } else {
        const float w1 = (float)(r.x >> 16) * (1.0f / 65536.0f), w0 = 1.0f - w1;
        const float w1s = w1 * 16777216.0f, w0s = w0 * 16777216.0f;
        b2_add(acc, e, w0s * f.x, w0s * f.y);
        b2_add(acc, e ^ ((2u << t) - 1u), w1s * f.x, w1s * f.y);
    }
}

#ifndef B2_COMBINE
#define B2_COMBINE 1
#endif
#ifndef B3_WALK
#define B3_WALK 8                                  // consecutive records of a run per thread in the run walk (combined in registers when they share entries);
                                                  // measured: 4 -> 372 / 195 us (random init / fitted field), 8 -> 368 / 180, 16 -> 461 / 223 (over the 64-VGPR budget of two
                                                  // resident workgroups; capped to it, 16 on the two coarsest tables only: 375 / 179 — no better than 8 everywhere)
#endif
struct B3Pending { uint32_t key; unsigned long long a0, b0, a1, b1; };   // key = (local entry | pair shift << 12): both entries of the record; sums per entry and channel
__device__ __forceinline__ void b3_flush_pending(long long *acc, const B3Pending &c) {
    if (c.key == 0xFFFFFFFFu) return;
    const uint32_t e = c.key & (BN_CHUNK - 1), t = (c.key >> 12) & 15u;
    atomicAdd(reinterpret_cast<unsigned long long *>(&acc[e]), c.a0);
    atomicAdd(reinterpret_cast<unsigned long long *>(&acc[BN_CHUNK + e]), c.b0);
    if (t != B2_SINGLE) {
        const uint32_t e1 = e ^ ((2u << t) - 1u);
        atomicAdd(reinterpret_cast<unsigned long long *>(&acc[e1]), c.a1);
        atomicAdd(reinterpret_cast<unsigned long long *>(&acc[BN_CHUNK + e1]), c.b1);
    }
}
__device__ __forceinline__ void b3_push_record(long long *acc, B3Pending &c, const uint2 r) {
    union { uint32_t u; __half2 h; } v;
    v.u = r.y;
    const float2 f = __half22float2(v.h);
    const uint32_t key = r.x & 0xFFFFu, t = (r.x >> 12) & 15u;
    unsigned long long a0, b0, a1 = 0, b1 = 0;
    if (t == B2_SINGLE) {
        a0 = b2_fix(f.x * 16777216.0f); b0 = b2_fix(f.y * 16777216.0f);
    } else {                                                                        // (the arithmetic of b2_add_record, value for value)
        const float w1 = (float)(r.x >> 16) * (1.0f / 65536.0f), w0 = 1.0f - w1;
        const float w1s = w1 * 16777216.0f, w0s = w0 * 16777216.0f;
        a0 = b2_fix(w0s * f.x); b0 = b2_fix(w0s * f.y);
        a1 = b2_fix(w1s * f.x); b1 = b2_fix(w1s * f.y);
    }
    if (key == c.key) {
        c.a0 += a0; c.b0 += b0; c.a1 += a1; c.b1 += b1;
    } else {
        b3_flush_pending(acc, c);
        c.key = key; c.a0 = a0; c.b0 = b0; c.a1 = a1; c.b1 = b1;
    }
}

// ================================================================================================ fp16 records, third form (round 5)
// The second form needs the EXACT record count of every (point block, bin) pair before the first record is written: a histogram pass over
// all 16 levels (the corner arithmetic of the emit, a second time) plus two scans, issued on a side stream beside the forward.  Measured with
// the plan frozen (scratch/stale_plan.py, profiles/r05_stale_plan_control.json): that "hidden" pre-pass costs the step 113 us — its 1.1 ms of
// side-stream kernel time takes CUs from the gathers and the field backward.  It also fixes the record count before the gradients exist, so
// samples whose gradient is exactly zero still travel as records.  Since the sums became exact fixed point the ORDER of a bin's records is
// free, so nothing has to be counted ahead of time — the emit counts for itself:
//   k_bin3_emit   a block (2048 points of one level) derives its corner pairs ONCE, takes an LDS ticket per record against per-bin counters
//                 that start at zero, scans the 128 counters (the block's own histogram) and sorts the records by bin into the LDS staging area.
//                 Samples whose gradient is exactly zero on this level emit nothing (they add exactly zero to every fixed-point sum).
//     hashed levels (interleaved bins, loads near-uniform): every bin owns a slab region of fixed capacity (1.5 x its expected load); the block
//                 reserves its run in each bin with ONE returning atomic per (block, bin) on the bin's cursor — issued right after the scan,
//                 consumed at copy-out — and copies its runs there: bin-major, densely packed, exactly the layout the accumulate kernel streams.
//                 What does not fit (importance samples crowding a few hundred entries of a coarse hashed level) SPILLS: it stays in the block's
//                 private region and the run table says so.
//     dense levels (a few crowded, uneven bins): the staging area leaves as ONE contiguous run into the block's private region, and the run
//                 table gets {count, offset} per (bin, block).
//   k_bin3_totals per bin: dense (and overflowed hashed) — exclusive prefix of the run counts over the blocks; the bin's record total
//   k_bin_scan_bins   segments per bin (as before)
//   k_bin3_accum      a segment of a hashed bin streams its range of the bin's region; a segment of a dense bin (and the first segment of an
//                     overflowed hashed bin, for the spill) walks the runs of the point blocks through a tile list in LDS
//   k_bin3_reduce_split
// Same 8-byte pair records, same exact 64-bit fixed-point sums, same split-bin reduction: the gradient is bit-identical to the second form's.
#ifndef B3_PTS
#define B3_PTS 2048
#endif
#ifndef B3_THREADS
#define B3_THREADS 1024                          // workgroup of k_bin3_emit (B3_PTS / B3_THREADS samples per thread)
#endif
// staging capacity in records: a hashed level emits 4.03 records per sample on average (one x-pair in 2^B3_K = 128 leaves as two singles: 8256 per
// block; 8448 +- 31 with the 32-entry granules of round 5), a dense one 4; 8832 leaves room for both (what still does not fit spills), and two
// blocks (2 x 79.1 KiB) share a CU
#define B3_CAP (B3_PTS * 4 + 640)
#define B3_WIDE_CHUNKS BN_MAX_CHUNKS               // bins per level of a wide level (T = 2^21: 512 chunks of 4096 entries)
#ifndef B3_WALK_BLOCKS
#define B3_WALK_BLOCKS 256                       // point blocks whose runs the run walk flattens at a time (a multiple of 64; round 6: 64 -> 256, see b3_walk_runs)
#endif
#define B3_MAXT (B3_WALK_BLOCKS + 8)             // LDS word pairs of the run walk: B3_WALK_BLOCKS + 1 prefix words + B3_WALK_BLOCKS run positions
#define B3_REGION (B3_PTS * 8)                    // records a block may emit on one level (8 single records per sample): its region on a dense level

struct Bin3Plan {
    Bin2Plan p;
    uint32_t capb;                                 // record capacity of a hashed bin's region
    uint32_t dense_slot[GE_MAX_LEVELS];            // slot -> index among the dense slots (block-major regions), or 0xFFFFFFFF for a hashed level
    uint8_t hbits[GE_MAX_LEVELS];                  // hashed slot: log2(bins of the level)
};

// Hashed levels: INTERLEAVED bins.  The hash is x ^ h(y, z): with contiguous 4096-entry chunks the bin of a record is a function of the (y, z)
// row alone (x < 4096 never reaches the chunk bits), so on the coarser hashed levels — a few thousand rows — the bin loads follow the scene (bins
// at twice the mean load on the benchmark's level 5).  A bin here owns the entries whose index bits [B3_K, B3_K + log2 bins) equal the bin id:
// x spreads every row over the bins, the loads are uniform wherever the samples are spread, and a bin is still 4096 entries = one 64 KiB LDS
// image, written back as granules of 2^B3_K entries (x 2 channels x float32).  An x-pair stays one record while x ^ (x + 1) < 2^B3_K; the others
// travel as two single records (the four pairs of a sample share x, so the unpaired ones come in fours).
// Granule size (round 6 sweep, profiles/r06_granule_sweep.txt; emit / accumulate us, benchmark table | bear table | fitted field):
//   2^5 (round 5)  461 / 396 | 530 / 602 | 264 / 206        2^6  457 / 390 | 516 / 580 | 256 / 205
//   2^7 (shipped)  449 / 378 | 498 / 567 | 248 / 204        2^8  448 / 384 | 493 / 559 | 246 / 200
// — fewer unpaired records (1 pair in 128 instead of 1 in 32) and 1 KiB instead of 256-byte granules under the flush (which carries the table's
// optimiser step since round 6); at 2^8 the x coordinate of a level coarser than 256 lines no longer reaches the bin bits at all (the imbalance the
// interleave exists to remove), for no further gain.
#ifndef B3_K
#define B3_K 7u
#endif
__device__ __forceinline__ uint32_t b3_bin_of(uint32_t e, uint32_t hbits) { return (e >> B3_K) & ((1u << hbits) - 1u); }
__device__ __forceinline__ uint32_t b3_local_of(uint32_t e, uint32_t hbits) { return ((e >> (B3_K + hbits)) << B3_K) | (e & ((1u << B3_K) - 1u)); }
__device__ __forceinline__ uint32_t b3_entry_of(uint32_t local, uint32_t bin, uint32_t hbits) {
    return ((local >> B3_K) << (B3_K + hbits)) | (bin << B3_K) | (local & ((1u << B3_K) - 1u));
}
__device__ __forceinline__ bool b3_paired_h(uint32_t i0, uint32_t i1) {
    const uint32_t m = i0 ^ i1;
    return m != 0 && m < (1u << B3_K) && (m & (m + 1)) == 0;
}

__global__ void __launch_bounds__(B3_THREADS, B3_THREADS >= 1024 ? 8 : 4) k_bin3_emit(const __half *__restrict__ grad, const float *__restrict__ inputs, const GridLevels lv,
                                                          const Bin3Plan plan, uint32_t *__restrict__ runs, uint32_t *__restrict__ cursor,
                                                          uint2 *__restrict__ hslab, uint2 *__restrict__ dslab, uint32_t B,
                                                          uint32_t gridtype, int align_corners, uint32_t interp, float *__restrict__ grad_grid, uint32_t n_slots,
                                                          uint32_t abl_arg, float *__restrict__ found_inf) {
#ifdef CNERF_TUNING
    const uint32_t abl = abl_arg;                  // timing aid (results wrong): 1 no reservations, 2 no copy-out, 4 no phase 2, 8 no run-table stores, 16 no tickets, 32 no corner arithmetic, 64 no loads
#else
    constexpr uint32_t abl = 0;
    (void)abl_arg;
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char b3_lds[];        // one LDS object: staged records, bin ids, counters, run starts, run destinations
    uint2 *s_rec = reinterpret_cast<uint2 *>(b3_lds);
    uint8_t *s_bin = b3_lds + (size_t)B3_CAP * 8;
    const uint32_t nb = plan.p.nb;
    // level fastest: the workgroups resident at one time cover all levels of a few point blocks — their reservations spread over every bin
    // cursor of the table instead of hammering the 128 of one level (same-address atomics serialise at the memory side), and the blocks that
    // share a point block's coordinates run together
    const uint32_t slot = blockIdx.x % n_slots, pb = blockIdx.x / n_slots;
    const uint32_t level = lv.order[slot];
    const uint32_t bin0 = plan.p.bin_first[slot], nch = plan.p.bin_first[slot + 1] - bin0;
    const bool dense_lvl = plan.dense_slot[slot] != 0xFFFFFFFFu;
    const uint32_t hb = plan.hbits[slot];
    // WIDE levels (round 6: more than 128 bins — T = 2^20 / 2^21 tables, the reference field's own): B3_WIDE_CHUNKS-entry counter tables in place of
    // the one-byte bin ids of the staged records (the copy-out then goes bin by bin instead of slot by slot), so that the workgroup's LDS
    // stays below half a CU's.  Block-uniform.
    const bool wide = nch > B2S_MAX_CHUNKS;
    const uint32_t NBN = wide ? B3_WIDE_CHUNKS : B2S_MAX_CHUNKS;
    uint32_t *cnt = reinterpret_cast<uint32_t *>(b3_lds + (size_t)B3_CAP * (wide ? 8 : 9));
    uint32_t *start = cnt + NBN;
    uint32_t *gdst = start + NBN;                                                   // hashed: first record of the block's run inside the bin's region
    uint32_t *s_total = gdst + NBN;
    auto bin_of = [&](uint32_t e) { return dense_lvl ? e >> BN_CHUNK_LOG2 : b3_bin_of(e, hb); };
    auto local_of = [&](uint32_t e) { return dense_lvl ? e & (BN_CHUNK - 1) : b3_local_of(e, hb); };
    auto paired = [&](uint32_t a, uint32_t b) { return dense_lvl ? b2_paired(a, b) : b3_paired_h(a, b); };
    if (threadIdx.x < NBN) cnt[threadIdx.x] = 0;
    __syncthreads();
    constexpr int PPT = B3_PTS / B3_THREADS;
    bool ok[PPT];
    uint32_t i0[PPT][4], i1[PPT][4], tk[PPT][4];
    float wyz[PPT][4], fx[PPT], g0[PPT], g1[PPT];
    // ---- phase 1: corner pairs, gradients, one LDS ticket per record (ticket = rank of the record inside the block's run for its bin)
    // every load of the thread's samples first, unconditionally (clamped row): the coordinates and the gradient of a sample are independent
    // loads — `if (in range) load gradient` made them two serial memory round trips per sample (A.1 item 7)
    float in_[PPT][3];
    FeatVec<__half, 2> gh[PPT];
#pragma unroll
    for (int i = 0; i < PPT; i++) {
        const uint32_t bc = min(pb * B3_PTS + i * B3_THREADS + threadIdx.x, B - 1);
        ge_load_coords<3>(inputs, bc, in_[i]);
        gh[i] = reinterpret_cast<const FeatVec<__half, 2> *>(grad)[(size_t)level * B + bc];
    }
#pragma unroll
    for (int i = 0; i < PPT; i++) {
        const uint32_t b = pb * B3_PTS + i * B3_THREADS + threadIdx.x;
        const float (&in)[3] = in_[i];
        ok[i] = b < B && !(in[0] < 0 || in[0] > 1 || in[1] < 0 || in[1] > 1 || in[2] < 0 || in[2] > 1);
        g0[i] = __half2float(gh[i].v[0]); g1[i] = __half2float(gh[i].v[1]);
        ok[i] = ok[i] && (g0[i] != 0.0f || g1[i] != 0.0f);                       // a zero gradient adds zero to every sum: no records,
        fx[i] = 0.0f;                                                            // and no corner arithmetic either (a NaN is not zero)
        if (ok[i]) {
            if (abl & 32) {
#pragma unroll
                for (int q = 0; q < 4; q++) { i0[i][q] = (b * 4 + q) * 2654435761u % lv.size[level]; i1[i][q] = i0[i][q] ^ 1u; wyz[i][q] = 0.25f; }
                fx[i] = in[0] - floorf(in[0]);
            } else {
                b2_pairs(in, lv, level, gridtype, align_corners, interp, i0[i], i1[i], wyz[i], fx[i]);
            }
            b2_poison(g0[i], g1[i], grad_grid, lv, level, i0[i][0], found_inf);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (abl & 16) { tk[i][q] = 0; continue; }
                const uint32_t c0 = bin_of(i0[i][q]);
                if (paired(i0[i][q], i1[i][q])) {
                    tk[i][q] = dense_lvl ? b2_ticket(cnt, c0) : atomicAdd(&cnt[c0], 1u);
                } else {
                    const uint32_t t0 = atomicAdd(&cnt[c0], 1u);
                    const uint32_t t1 = atomicAdd(&cnt[bin_of(i1[i][q])], 1u);
                    tk[i][q] = t0 | (t1 << 16);                                  // (a block emits at most 16384 records per level)
                }
            }
        }
    }
    __syncthreads();
    // ---- the block's histogram is complete: run starts (two bins per lane of the first wave); dense: the run table; hashed: the reservations
    uint32_t res_a = 0, res_b = 0;
    if (wide) {
        // 512 counters: the first wave takes eight consecutive bins per lane (run starts in bin order, as below); the reservations are then
        // issued by 512 threads, one bin each (one register per thread: they come back during phase 2)
        if (threadIdx.x < 64) {
            const uint32_t lane = threadIdx.x;
            constexpr uint32_t KPL = B3_WIDE_CHUNKS / 64;
            uint32_t c8[KPL], loc = 0;
#pragma unroll
            for (uint32_t k = 0; k < KPL; k++) { c8[k] = cnt[lane * KPL + k]; loc += c8[k]; }
            const uint32_t incl = cn_wave_incl_scan(loc);
            uint32_t run = incl - loc;
#pragma unroll
            for (uint32_t k = 0; k < KPL; k++) {
                const uint32_t c = lane * KPL + k;
                start[c] = run;
                if (dense_lvl && c < nch && !(abl & 8)) runs[(size_t)(bin0 + c) * nb + pb] = c8[k] | (run << 16);
                run += c8[k];
            }
            if (lane == 63) *s_total = incl;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // The reservations of TWO neighbouring bins in one 64-bit returning atomic (the cursors are consecutive 32-bit words; a cursor stays
        // below 2^31 plus its flag bit, so the low word never carries): 512 bins x 1024 point blocks x 11 levels are 5.8 M reservations per step,
        // 0.27 ms of the memory-side atomic unit (A.1 item 1) that phase 2 cannot hide — pairs halve them.  Pairs are aligned on the GLOBAL bin
        // index; a pair that straddles the level's first / last bin adds zero to its neighbour.
        if (!dense_lvl && !(abl & 1) && threadIdx.x <= NBN / 2) {
            const uint32_t b_lo = ((bin0 >> 1) + threadIdx.x) * 2u;                 // global bin of the pair's low word
            uint32_t sa[2] = {0u, 0u};
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint32_t c = b_lo + h - bin0;                                 // (wraps for a bin in front of the level: fails c < nch)
                if (c < nch) {
                    const uint32_t n = cnt[c], st0 = start[c];
                    sa[h] = st0 >= B3_CAP ? 0u : min(n, (uint32_t)B3_CAP - st0);
                }
            }
            if (sa[0] | sa[1]) {
                const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long *>(cursor + b_lo), (unsigned long long)sa[0] | ((unsigned long long)sa[1] << 32));
                res_a = (uint32_t)old & 0x7FFFFFFFu;
                res_b = (uint32_t)(old >> 32) & 0x7FFFFFFFu;
            }
        }
    } else if (threadIdx.x < 64) {
        // 128 counters: lane l takes the bins 2 l and 2 l + 1 (run starts in bin order); their two reservations are ONE 64-bit returning atomic
        // when the pair is aligned (round 6: the memory-side atomic unit is the resource, A.1 item 1 — 1.4 M reservations per step become 0.7 M)
        const uint32_t lane = threadIdx.x;
        const uint32_t ca = cnt[2 * lane], cb = cnt[2 * lane + 1];
        const uint32_t incl = cn_wave_incl_scan(ca + cb);
        const uint32_t oa = incl - (ca + cb), ob = oa + ca;
        start[2 * lane] = oa; start[2 * lane + 1] = ob;
        if (lane == 63) *s_total = incl;
        if (dense_lvl) {
            if (2 * lane < nch && !(abl & 8)) runs[(size_t)(bin0 + 2 * lane) * nb + pb] = ca | (oa << 16);
            if (2 * lane + 1 < nch && !(abl & 8)) runs[(size_t)(bin0 + 2 * lane + 1) * nb + pb] = cb | (ob << 16);
        } else if (!(abl & 1)) {
            // reserve room in the bin's region for what the staging area holds of the run (the rest, if any, is already in the block's region)
            const uint32_t sa = (2 * lane < nch && oa < B3_CAP) ? min(ca, (uint32_t)B3_CAP - oa) : 0u;
            const uint32_t sb = (2 * lane + 1 < nch && ob < B3_CAP) ? min(cb, (uint32_t)B3_CAP - ob) : 0u;
            if ((bin0 & 1u) == 0) {
                if (sa | sb) {
                    const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long *>(cursor + bin0 + 2 * lane),
                                                             (unsigned long long)sa | ((unsigned long long)sb << 32));
                    res_a = (uint32_t)old & 0x7FFFFFFFu;
                    res_b = (uint32_t)(old >> 32) & 0x7FFFFFFFu;
                }
            } else {
                if (sa) res_a = atomicAdd(&cursor[bin0 + 2 * lane], sa) & 0x7FFFFFFFu;
                if (sb) res_b = atomicAdd(&cursor[bin0 + 2 * lane + 1], sb) & 0x7FFFFFFFu;
            }
        }
    }
    // bare barrier: only the LDS writes above (start[], *s_total) must have landed; __syncthreads() would also wait for the returning atomics
    // (vmcnt(0)) — they have all of phase 2 to come back
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // ---- phase 2: the records, sorted by bin, into the staging area
    uint2 *__restrict__ region = dslab + ((size_t)slot * nb + pb) * B3_REGION;        // the block's private region on this level
#pragma unroll
    for (int i = 0; i < PPT; i++) {
        if (!ok[i] || (abl & 4)) continue;
        const uint32_t fxq = min((uint32_t)(fx[i] * 65536.0f), 65535u);
        union { __half2 h; uint32_t u; } v;
        auto put = [&](uint32_t c, uint32_t ticket, uint32_t word, uint32_t val) {
            const uint32_t sl = start[c] + ticket;
            if (sl < B3_CAP) { s_rec[sl] = make_uint2(word, val); if (!wide) s_bin[sl] = (uint8_t)c; }
            else region[sl] = make_uint2(word, val);                                // beyond the staging capacity: straight into the block's region
        };
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t a0 = i0[i][q], a1 = i1[i][q];
            const uint32_t c0 = bin_of(a0);
            if (paired(a0, a1)) {
                const uint32_t t = 31u - (uint32_t)__clz((int)(a0 ^ a1));
                v.h = __floats2half2_rn(wyz[i][q] * g0[i], wyz[i][q] * g1[i]);
                put(c0, tk[i][q], local_of(a0) | (t << 12) | (fxq << 16), v.u);
            } else {
                const float w0 = (1 - fx[i]) * wyz[i][q], w1 = fx[i] * wyz[i][q];
                v.h = __floats2half2_rn(w0 * g0[i], w0 * g1[i]);
                put(c0, tk[i][q] & 0xFFFFu, local_of(a0) | (B2_SINGLE << 12), v.u);
                v.h = __floats2half2_rn(w1 * g0[i], w1 * g1[i]);
                put(bin_of(a1), tk[i][q] >> 16, local_of(a1) | (B2_SINGLE << 12), v.u);
            }
        }
    }
    if (!dense_lvl && wide) {
        if (threadIdx.x <= NBN / 2) {                                               // (the pair's two bins; the arithmetic of the narrow form below)
            const uint32_t capb = plan.capb, b_lo = ((bin0 >> 1) + threadIdx.x) * 2u;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint32_t c = b_lo + h - bin0;
                if (c >= nch) continue;
                const uint32_t n = cnt[c], st0 = start[c], res = h ? res_b : res_a;
                const uint32_t staged = st0 >= B3_CAP ? 0u : min(n, (uint32_t)B3_CAP - st0);
                const uint32_t room = res >= capb ? 0u : capb - res;
                const uint32_t fit = min(min(n, room), staged);
                gdst[c] = res - st0;
                cnt[c] = fit;
                if (!(abl & 8)) runs[(size_t)(bin0 + c) * nb + pb] = (n - fit) | ((st0 + fit) << 16);
                if (n > fit) atomicOr(&cursor[bin0 + c], 0x80000000u);
            }
        }
    } else if (!dense_lvl && threadIdx.x < 64) {                                    // the reservations have had phase 2 to come back
        // what of the run fits the bin's region goes there; the rest SPILLS: it stays in the block's region (where the staging order puts it anyway)
        // and the run table says so — the accumulate workgroup of an overflowed bin walks those runs after its region.  Records beyond the staging
        // capacity (already in the block's region) count as spilled whatever the cursor says.
        const uint32_t capb = plan.capb;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t c = 2 * threadIdx.x + h;
            const uint32_t n = cnt[c], st0 = start[c], res = h ? res_b : res_a;
            const uint32_t staged = st0 >= B3_CAP ? 0u : min(n, (uint32_t)B3_CAP - st0);
            const uint32_t room = res >= capb ? 0u : capb - res;
            const uint32_t fit = min(min(n, room), staged);
            gdst[c] = res - st0;                                                    // staging slot -> position in the bin's region (wrapping arithmetic)
            cnt[c] = fit;                                                           // (the counts are not needed any more: records of the run that go to the region)
            if (c < nch && !(abl & 8)) runs[(size_t)(bin0 + c) * nb + pb] = (n - fit) | ((st0 + fit) << 16);
            if (c < nch && n > fit) atomicOr(&cursor[bin0 + c], 0x80000000u);       // the bin has spilled runs: its accumulate workgroup must walk the run table
        }
    }
    __syncthreads();
    const uint32_t total = (abl & 2) ? 0u : min(*s_total, (uint32_t)B3_CAP);
    if (dense_lvl) {
        for (uint32_t sl = threadIdx.x; sl < total; sl += B3_THREADS) region[sl] = s_rec[sl];
    } else if (wide) {
        // bin by bin, four bins per wave at a time (a run is ~16 records: sixteen lanes each); a bin's staged records are the slots
        // [start, next start) below `total`.  (Measured on the narrow levels too — one bin per wave, no bin-id bytes in the staging area: emit
        // 469 -> 485 us, fitted 266 -> 304: the flat slot-by-slot copy below stays for them, profiles/r06_scatter_walk_bybin_ab.txt.)
        const uint32_t capb = plan.capb, l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;
        for (uint32_t c = grp; c < nch; c += B3_THREADS / 16) {
            const uint32_t st0 = start[c], en = min(c + 1 < NBN ? start[c + 1] : *s_total, total), fit = cnt[c], gd = gdst[c];
            for (uint32_t sl = st0 + l16; sl < en; sl += 16) {
                if (sl - st0 < fit) hslab[(size_t)(bin0 + c) * capb + (gd + sl)] = s_rec[sl];
                else region[sl] = s_rec[sl];
            }
        }
    } else {
        const uint32_t capb = plan.capb;
        for (uint32_t sl = threadIdx.x; sl < total; sl += B3_THREADS) {
            const uint32_t c = s_bin[sl];
            const uint32_t pos = gdst[c] + sl;                                      // (wrapping 32-bit arithmetic: gdst = reservation - run start)
            if (sl - start[c] < cnt[c]) hslab[(size_t)(bin0 + c) * capb + pos] = s_rec[sl];
            else region[sl] = s_rec[sl];
        }
    }
}

// per bin: dense level — exclusive prefix of the run counts (low half of the run-table words) over the point blocks -> pre, and the bin total;
// hashed level — the total is what the bin's region holds (its cursor, capped), and only a bin that overflowed (cursor > capacity) needs the
// prefix of its SPILLED runs
__global__ void __launch_bounds__(BN_SCAN_THREADS) k_bin3_totals(const uint32_t *__restrict__ runs, uint32_t *__restrict__ pre, const uint32_t *__restrict__ cursor,
                                                                 uint32_t *__restrict__ bin_total, const Bin3Plan plan, uint32_t n_slots) {
    constexpr uint32_t NW = BN_SCAN_THREADS / 64;
    __shared__ uint32_t wave_tot[NW];
    __shared__ uint32_t carry;
    const uint32_t bin = blockIdx.x, nb = plan.p.nb;
    uint32_t slot = 0;
    while (slot + 1 < n_slots && plan.p.bin_first[slot + 1] <= bin) slot++;
    const bool hashed = plan.dense_slot[slot] == 0xFFFFFFFFu;
    if (hashed) {
        const uint32_t cw = cursor[bin], c = cw & 0x7FFFFFFFu;                      // bit 31: some block spilled a run of this bin
        if (!(cw >> 31)) {
            if (threadIdx.x == 0) bin_total[bin] = min(c, plan.capb);
            return;
        }
    }
    const uint32_t *h = runs + (size_t)bin * nb;
    uint32_t *o = pre + (size_t)bin * nb;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint32_t s0 = 0; s0 < nb; s0 += BN_SCAN_THREADS * 4) {
        const uint32_t i = s0 + tid * 4;
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = i + k < nb ? (h[i + k] & 0xFFFFu) : 0;
        const uint32_t mine = v[0] + v[1] + v[2] + v[3];
        const uint32_t incl = cn_wave_incl_scan(mine);
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        uint32_t wbase = 0, tot = 0;
#pragma unroll
        for (uint32_t w = 0; w < NW; w++) {
            const uint32_t t = wave_tot[w];
            if (w < wave) wbase += t;
            tot += t;
        }
        const uint32_t base = carry;
        uint32_t run = base + wbase + incl - mine;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (i + k < nb) o[i + k] = run;
            run += v[k];
        }
        __syncthreads();
        if (tid == 0) carry = base + tot;
        __syncthreads();
    }
    // (a hashed bin's record index space: what its region holds, then its spilled runs)
    if (tid == 0) bin_total[bin] = hashed ? min(cursor[bin] & 0x7FFFFFFFu, plan.capb) + carry : carry;
}

// ---- the table's optimiser step inside the scatter (round 6, cnerf_grid_backward_adam): the flush below is where a table entry's gradient of this
// backward pass becomes final — one owner per bin, or k_bin3_reduce_split for a split one — and k_adam_scaled would read it back 0.4 ms later, together
// with p / m / v, at the HBM roofline (62 us for the benchmark table).  Applied here the update rides on kernels that are bound by LDS atomics, and the
// gradient never makes the round trip.  Same arithmetic as k_adam_scaled (misc.hip), value for value: the parameters after a step are bit-identical.
struct B3Adam {
    float *p, *m, *v;
    __half *ph;
    const float *state;                              // the loss scaler's {scale, growth_tracker, found_inf, good_steps}
    float lr, beta1, beta2, eps, extra_inv;
    int zero_grad, on;
    const float *cst;                                // {gscale, step_size, rsqrt_bc2, skip}: written by k_bin_scan_bins of the same backward pass
};
struct B3AdamConst { float gscale, step_size, rsqrt_bc2; int skip; };
__device__ __forceinline__ B3AdamConst b3_adam_const(const B3Adam &ad) {           // (k_adam_scaled's prologue: see k_bin_scan_bins)
    B3AdamConst c;
    c.gscale = ad.cst[0]; c.step_size = ad.cst[1]; c.rsqrt_bc2 = ad.cst[2]; c.skip = ad.cst[3] != 0.0f;
    return c;
}
// gg = the final gradient of four consecutive floats at float index `idx` (a multiple of 4), pp / mm / vv = parameter and moments there;
// d4 = their slot in the gradient table
__device__ __forceinline__ void b3_adam_update(const B3Adam &ad, const B3AdamConst &c, size_t idx, float4 gg, float4 pp, float4 mm, float4 vv, float4 *d4) {
    float *pa = &pp.x, *ga = &gg.x, *ma = &mm.x, *va = &vv.x;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float gk = ga[k] * c.gscale;
        ma[k] = ad.beta1 * ma[k] + (1.0f - ad.beta1) * gk;
        va[k] = ad.beta2 * va[k] + (1.0f - ad.beta2) * gk * gk;
        pa[k] -= c.step_size * ma[k] / (sqrtf(va[k]) * c.rsqrt_bc2 + ad.eps);
    }
    *reinterpret_cast<float4 *>(ad.p + idx) = pp;
    *reinterpret_cast<float4 *>(ad.m + idx) = mm;
    *reinterpret_cast<float4 *>(ad.v + idx) = vv;
    *d4 = ad.zero_grad ? make_float4(0, 0, 0, 0) : gg;
    if (ad.ph) {
        union { __half2 h[2]; uint2 u; } o;
        o.h[0] = __floats2half2_rn(pp.x, pp.y);
        o.h[1] = __floats2half2_rn(pp.z, pp.w);
        *reinterpret_cast<uint2 *>(ad.ph + idx) = o.u;
    }
}
__device__ __forceinline__ void b3_adam_apply(const B3Adam &ad, const B3AdamConst &c, size_t idx, float4 gg, float4 *d4) {
    if (c.skip) {                                    // non-finite gradients somewhere in this step: no update, the gradients are still cleared
        *d4 = ad.zero_grad ? make_float4(0, 0, 0, 0) : gg;
        return;
    }
    b3_adam_update(ad, c, idx, gg, *reinterpret_cast<const float4 *>(ad.p + idx), *reinterpret_cast<const float4 *>(ad.m + idx),
                   *reinterpret_cast<const float4 *>(ad.v + idx), d4);
}

// flush of a finished LDS image: sole owner -> read-modify-write of the gradient table; a split bin parks its fixed-point image.
// hbits = 0xFF: dense level (the bin is a contiguous chunk); else a hashed level's interleaved bin.
__device__ __forceinline__ void b3_flush(const long long *acc, const GridLevels &lv, const Bin2Plan &plan, uint32_t slot, uint32_t bin, uint32_t nseg, uint32_t gseg,
                                         float *__restrict__ grad_grid, long long *__restrict__ partial, uint32_t hbits, const B3Adam &ad,
                                         const B3AdamConst &adc) {
    const uint32_t level = lv.order[slot];
    const uint32_t cb = bin - plan.bin_first[slot];
    const uint32_t e0 = hbits == 0xFFu ? cb << BN_CHUNK_LOG2 : 0u;
    const uint32_t n_entries = hbits == 0xFFu ? min(BN_CHUNK, lv.size[level] - e0) : min(BN_CHUNK, lv.size[level] >> hbits);
    float *__restrict__ dst = grad_grid + (size_t)lv.offset[level] * 2;
    if (nseg == 1 && ad.on && !adc.skip) {
        // the optimiser step rides on the flush: a full bin is two groups of four floats per thread — every load of both (gradient, parameter,
        // two moments) is issued before the first use (one memory round trip per workgroup instead of two)
        constexpr int NIT = BN_CHUNK / 2 / 1024;
        float4 gg[NIT], pp[NIT], mm[NIT], vv[NIT];
        auto index_of = [&](uint32_t j) {                                          // float index of the group (recomputed at the use: registers)
            const uint32_t e = hbits == 0xFFu ? e0 + 2 * j : b3_entry_of(2 * j, cb, hbits);
            return ((size_t)lv.offset[level] + e) * 2;
        };
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const uint32_t j = it * 1024 + threadIdx.x;
            if (j < n_entries / 2) {
                const size_t idx = index_of(j);
                gg[it] = *reinterpret_cast<const float4 *>(grad_grid + idx);
                pp[it] = *reinterpret_cast<const float4 *>(ad.p + idx);
                mm[it] = *reinterpret_cast<const float4 *>(ad.m + idx);
                vv[it] = *reinterpret_cast<const float4 *>(ad.v + idx);
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const uint32_t j = it * 1024 + threadIdx.x;
            if (j >= n_entries / 2) continue;
            const size_t idx = index_of(j);
            float4 g = gg[it];
            g.x += bn_acc_to_float<__half>(acc[j * 2]); g.y += bn_acc_to_float<__half>(acc[BN_CHUNK + j * 2]);
            g.z += bn_acc_to_float<__half>(acc[j * 2 + 1]); g.w += bn_acc_to_float<__half>(acc[BN_CHUNK + j * 2 + 1]);
            b3_adam_update(ad, adc, idx, g, pp[it], mm[it], vv[it], reinterpret_cast<float4 *>(grad_grid + idx));
        }
    } else if (nseg == 1) {
        for (uint32_t j = threadIdx.x; j < n_entries / 2; j += 1024) {             // two entries x two channels per thread
            const uint32_t e = hbits == 0xFFu ? e0 + 2 * j : b3_entry_of(2 * j, cb, hbits);
            float4 *d4 = reinterpret_cast<float4 *>(dst + (size_t)e * 2);
            float4 g = *d4;
            g.x += bn_acc_to_float<__half>(acc[j * 2]); g.y += bn_acc_to_float<__half>(acc[BN_CHUNK + j * 2]);
            g.z += bn_acc_to_float<__half>(acc[j * 2 + 1]); g.w += bn_acc_to_float<__half>(acc[BN_CHUNK + j * 2 + 1]);
            if (ad.on) b3_adam_apply(ad, adc, ((size_t)lv.offset[level] + e) * 2, g, d4);      // (a skipped step: clears the gradient)
            else *d4 = g;
        }
    } else {
        long long *__restrict__ img = partial + (size_t)gseg * (BN_CHUNK * 2);
        for (uint32_t j = threadIdx.x; j < n_entries * 2; j += 1024) img[j] = acc[(j & 1) * BN_CHUNK + (j >> 1)];      // interleaved (local entry, channel) order
    }
}

// sum the fixed-point partial images of the split bins into the gradient table (contiguous chunks on dense levels, interleaved bins on hashed / wrapped ones)
// The crowded bins of the dense levels split into dozens of segments (level 0 of the benchmark table: two bins of 4 M records, ~57 segments each): one
// thread summing all of a group's partials was a serial chain of that many dependent load rounds on a handful of workgroups (35 us).  Round 6: a wave
// takes every fourth segment of 64 groups (64-bit integer sums: any order gives the same bits), the four waves meet in LDS; the workgroups walk the
// list of split bins that k_bin_scan_bins leaves behind (one workgroup per bin and slice of EVERY bin was 200 k empty workgroups on the bear table):
// 27 us (profiles/r06_reduce_split_ab.txt).
#define B3_RS_GROUPS 64                             // groups of four values (two local entries x two channels) per workgroup
#ifndef B3_RS_BINS
#define B3_RS_BINS 256                              // grid.x: workgroups striding over the split-bin list
#endif
__global__ void __launch_bounds__(256) k_bin3_reduce_split(const long long *__restrict__ partial, const uint32_t *__restrict__ seg_first,
                                                           const GridLevels lv, const Bin3Plan plan, float *__restrict__ grad_grid, uint32_t n_slots,
                                                           const uint32_t *__restrict__ split_list, const B3Adam ad) {
    __shared__ long long s_sum[3][B3_RS_GROUPS][4];
    const uint32_t n_split = split_list ? split_list[0] : gridDim.x;                // (no list: one workgroup column per bin)
    if (blockIdx.x >= n_split) return;
    B3AdamConst adc = {0.0f, 0.0f, 0.0f, 0};
    if (ad.on) adc = b3_adam_const(ad);
    for (uint32_t b = blockIdx.x; b < n_split; b += gridDim.x) {                    // (workgroup-uniform trip count: the barrier below is safe)
    const uint32_t bin = split_list ? split_list[1 + b] : b;
    const uint32_t s0 = seg_first[bin], nseg = seg_first[bin + 1] - s0;
    if (nseg <= 1) continue;
    uint32_t slot = 0;
    while (slot + 1 < n_slots && plan.p.bin_first[slot + 1] <= bin) slot++;
    const uint32_t level = lv.order[slot];
    const bool dense_lvl = plan.dense_slot[slot] != 0xFFFFFFFFu;
    const uint32_t hb = plan.hbits[slot], cb = bin - plan.p.bin_first[slot];
    const uint32_t e0 = dense_lvl ? cb << BN_CHUNK_LOG2 : 0u;
    const uint32_t n_entries = dense_lvl ? min(BN_CHUNK, lv.size[level] - e0) : min(BN_CHUNK, lv.size[level] >> hb);
    if (blockIdx.y * B3_RS_GROUPS >= n_entries / 2) continue;                      // (workgroup-uniform)
    const uint32_t lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const uint32_t j = blockIdx.y * B3_RS_GROUPS + lane;                           // group of four values (two local entries x two channels)
    const bool ok = j < n_entries / 2;
    long long sum[4] = {0, 0, 0, 0};
    // the finishing wave's read-modify-write operands (gradient; parameter and moments when the optimiser step rides along): issued in front of the sums
    const uint32_t e_dst = dense_lvl ? e0 + 2 * j : b3_entry_of(2 * j, cb, hb);
    const size_t idx = ((size_t)lv.offset[level] + e_dst) * 2;
    const bool fin = !sl && ok, upd = fin && ad.on && !adc.skip;
    float4 g_old = make_float4(0, 0, 0, 0), pp = g_old, mm = g_old, vv = g_old;
    if (fin) g_old = *reinterpret_cast<const float4 *>(grad_grid + idx);
    if (upd) {
        pp = *reinterpret_cast<const float4 *>(ad.p + idx);
        mm = *reinterpret_cast<const float4 *>(ad.m + idx);
        vv = *reinterpret_cast<const float4 *>(ad.v + idx);
    }
    if (ok) {
        const long long *__restrict__ src = partial + (size_t)s0 * (BN_CHUNK * 2) + (size_t)j * 4;
#pragma unroll 8
        for (uint32_t s = sl; s < nseg; s += 4) {
            const longlong2 v0 = reinterpret_cast<const longlong2 *>(src + (size_t)s * (BN_CHUNK * 2))[0];
            const longlong2 v1 = reinterpret_cast<const longlong2 *>(src + (size_t)s * (BN_CHUNK * 2))[1];
            sum[0] += v0.x; sum[1] += v0.y; sum[2] += v1.x; sum[3] += v1.y;
        }
    }
    if (sl) {
#pragma unroll
        for (int k = 0; k < 4; k++) s_sum[sl - 1][lane][k] = sum[k];
    }
    __syncthreads();
    if (!sl && ok) {
#pragma unroll
        for (int w = 0; w < 3; w++)
#pragma unroll
            for (int k = 0; k < 4; k++) sum[k] += s_sum[w][lane][k];
        float4 *dst = reinterpret_cast<float4 *>(grad_grid + idx);
        float4 g = g_old;
        g.x += bn_acc_to_float<__half>(sum[0]); g.y += bn_acc_to_float<__half>(sum[1]);
        g.z += bn_acc_to_float<__half>(sum[2]); g.w += bn_acc_to_float<__half>(sum[3]);
        if (upd) b3_adam_update(ad, adc, idx, g, pp, mm, vv, dst);
        else if (ad.on) *dst = ad.zero_grad ? make_float4(0, 0, 0, 0) : g;              // (a skipped step: clears the gradient)
        else *dst = g;
    }
    __syncthreads();                                                                 // (s_sum is reused by the next bin of this workgroup)
    }
}

// The records of a bin segment that live in RUNS of the point blocks' private regions (a dense level's bins; the spill of a hashed bin): the
// part [begin, end) of the bin's run index space, starting at block pb_first.  Runs are anything from a handful of records (a block that grazes
// the chunk's slab of a level-4 table) to thousands, so the runs of 64 blocks at a time are flattened: the first wave clips each run to the
// segment and leaves its source position and the exclusive prefix of the clipped lengths in LDS; thread t then takes records B3_WALK t .. B3_WALK (t + 1) - 1 of
// every 1024 B3_WALK of the flat index space (a 6-step search in the 64 prefix words for the first, a short walk for the rest) — that many loads in
// flight per thread, full lanes whatever the run lengths, and runs of same-entry records (neighbouring samples of a ray on a coarse level)
// become successive atomics of one lane instead of same-address lanes of one instruction.
template <int W>
__device__ __forceinline__ void b3_walk_runs(long long *acc, uint32_t *s_c, uint32_t *tiles, const uint32_t *__restrict__ rt, const uint32_t *__restrict__ pt,
                                             const uint2 *__restrict__ lvl_slab, uint32_t nb, uint32_t pb_first, uint32_t begin, uint32_t end) {
    const uint32_t lane = threadIdx.x & 63;
    // B3_WALK_BLOCKS runs at a time (round 6: 256 — with 64, a level whose (block, bin) runs are short gave the 8192 record slots of an iteration
    // a few thousand records and paid two barriers per 64 blocks: the reference field's level 4 has 156 bins, ~53 records per run)
    constexpr uint32_t CB = B3_WALK_BLOCKS, KPL = CB / 64;
    uint32_t *s_pre = tiles, *s_base = tiles + CB + 1;                              // [CB + 1] exclusive prefix of the clipped run lengths (+ total), [CB] first record of each
    for (uint32_t pb0 = pb_first; pb0 < nb; pb0 += CB) {
        if (threadIdx.x < 64) {
            uint32_t len[KPL], base[KPL], loc = 0, p_last = 0xFFFFFFFFu;
#pragma unroll
            for (uint32_t k = 0; k < KPL; k++) {
                const uint32_t pbl = pb0 + lane * KPL + k;
                uint32_t p_j = 0xFFFFFFFFu, r_j = 0;
                if (pbl < nb) { p_j = pt[pbl]; r_j = rt[pbl]; }
                uint32_t lo = 0, hi = 0;
                if (p_j < end) {
                    const uint32_t c_j = r_j & 0xFFFFu;
                    lo = max(p_j, begin) - p_j;
                    hi = max(min(p_j + c_j, end), p_j) - p_j;
                    if (hi < lo) hi = lo;
                }
                len[k] = hi - lo;
                base[k] = pbl * B3_REGION + (r_j >> 16) + lo;
                loc += len[k];
                p_last = p_j;
            }
            const uint32_t incl = cn_wave_incl_scan(loc);
            uint32_t run = incl - loc;
#pragma unroll
            for (uint32_t k = 0; k < KPL; k++) {
                s_pre[lane * KPL + k] = run;
                s_base[lane * KPL + k] = base[k];
                run += len[k];
            }
            if (lane == 63) { s_pre[CB] = incl; s_c[1] = p_last >= end ? 1u : 0u; } // 1 = the segment ends inside this chunk
        }
        __syncthreads();
        const uint32_t total = s_pre[CB], last = s_c[1];
        // (Round 6, measured and dropped: a STRIDED assignment — lane l of a wave takes sample l of one (y, z) row of W neighbouring rays, so that
        // the lanes of an instruction are 64 different depths and a lane's W records share a cell across rays — 369 -> 389 us at random init and
        // 194 -> 383 us on a fitted field: profiles/r06_gather_scatter_ab.txt.  Consecutive records per lane it stays.)
        for (uint32_t f0 = threadIdx.x * W; f0 < total; f0 += 1024 * W) {
            uint32_t j = 0;                                                         // largest j with s_pre[j] <= f0
#pragma unroll
            for (uint32_t step = CB / 2; step; step >>= 1)
                if (s_pre[j + step] <= f0) j += step;
            uint32_t idx[W];
            bool ok4[W];
#pragma unroll
            for (int u = 0; u < W; u++) {
                const uint32_t f = f0 + u;
                ok4[u] = f < total;
                while (ok4[u] && f >= s_pre[j + 1]) j++;                            // (empty runs are stepped over; j stays < 64 while f < total)
                idx[u] = ok4[u] ? s_base[j] + (f - s_pre[j]) : s_base[0];
            }
            uint2 v[W];
#pragma unroll
            for (int u = 0; u < W; u++) v[u] = lvl_slab[idx[u]];              // (unconditional on a valid index: masked at use)
            // Consecutive records of a run come from consecutive samples of a ray (wave-aggregated tickets), and on a coarse level a ray
            // spends a dozen samples in one cell: records with the same entry pair are summed in registers — integer sums, so the result is
            // the one of separate atomics, bit for bit — and go to LDS as one update.  Neighbouring lanes still meet on such an entry
            // (same-address atomics serialise); a longer in-lane stretch is fewer of them.
            B3Pending c;
            c.key = 0xFFFFFFFFu;
#pragma unroll
            for (int u = 0; u < W; u++)
                if (ok4[u]) b3_push_record(acc, c, v[u]);
            b3_flush_pending(acc, c);
        }
        __syncthreads();
        if (last) break;
    }
}

__global__ void __launch_bounds__(1024, 8) k_bin3_accum(const uint2 *__restrict__ hslab, const uint2 *__restrict__ dslab, const uint32_t *__restrict__ runs,
                                                     const uint32_t *__restrict__ pre, const uint32_t *__restrict__ cursor, const uint32_t *__restrict__ bin_base,
                                                     const uint32_t *__restrict__ seg_first, const GridLevels lv, const Bin3Plan plan,
                                                     float *__restrict__ grad_grid, long long *__restrict__ partial, const uint32_t *__restrict__ seg_bin,
                                                     uint32_t n_slots, uint32_t seg_records, uint32_t only, const B3Adam ad) {
    extern __shared__ __attribute__((aligned(16))) unsigned char bn_lds[];   // [2][BN_CHUNK] accumulators, scratch words, the tile list (one LDS object)
    long long *acc = reinterpret_cast<long long *>(bn_lds);
    uint32_t *s_w = reinterpret_cast<uint32_t *>(bn_lds + sizeof(long long) * BN_CHUNK * 2);
    const uint32_t gseg = blockIdx.x;
    if (gseg >= seg_first[plan.p.total_bins]) return;
#ifdef CNERF_TUNING
    if (only) {                                                          // timing aid (results wrong): 1 = hashed bins only, 2 = dense bins only
        uint32_t sl_ = 0;
        const uint32_t b_ = seg_bin[gseg];
        while (sl_ + 1 < n_slots && plan.p.bin_first[sl_ + 1] <= b_) sl_++;
        if ((plan.dense_slot[sl_] == 0xFFFFFFFFu) != (only == 1)) return;
    }
#else
    (void)only;
#endif
    const uint32_t nb = plan.p.nb;
    if (threadIdx.x == 0) {
        const uint32_t bin = seg_bin[gseg];
        uint32_t slot = 0;
        while (slot + 1 < n_slots && plan.p.bin_first[slot + 1] <= bin) slot++;
        uint32_t lo = 0;
        const bool hashed = plan.dense_slot[slot] == 0xFFFFFFFFu;
        const uint32_t cw = hashed ? cursor[bin] : 0u;
        const uint32_t treg = hashed ? min(cw & 0x7FFFFFFFu, plan.capb) : 0u;      // records in the bin's region (hashed); the run space starts behind them
        uint32_t begin = (gseg - seg_first[bin]) * seg_records;
        if (!hashed || ((cw >> 31) && begin > treg)) {
            // first run that reaches into [begin, ...): the last block whose prefix is <= begin
            begin -= treg;
            const uint32_t *p = pre + (size_t)bin * nb;
            uint32_t hi = nb;                                                  // invariant: p[lo] <= begin (p[0] = 0)
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (p[mid] <= begin) lo = mid; else hi = mid;
            }
        }
        s_w[0] = bin; s_w[1] = lo; s_w[2] = slot; s_w[3] = treg;
    }
    for (uint32_t i = threadIdx.x; i < sizeof(long long) * BN_CHUNK * 2 / 16; i += 1024) reinterpret_cast<uint4 *>(bn_lds)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const uint32_t bin = s_w[0], pb_first = s_w[1], slot = s_w[2], treg = s_w[3];
    const uint32_t seg = gseg - seg_first[bin], nseg = seg_first[bin + 1] - seg_first[bin];
    const uint32_t total = bin_base[bin + 1] - bin_base[bin];
    const uint32_t begin = seg * seg_records, end = min(begin + seg_records, total);
    uint32_t *s_c = s_w + 4, *tiles = s_w + 8;                                // {tiles in the chunk, segment ends here}; [B3_MAXT][2]: first record, count
    const uint32_t *__restrict__ rt = runs + (size_t)bin * nb, *__restrict__ pt = pre + (size_t)bin * nb;
    const uint2 *__restrict__ lvl_slab = dslab + (size_t)slot * nb * B3_REGION;
    if (plan.dense_slot[slot] == 0xFFFFFFFFu) {
        // ---- hashed level: the bin's records are one contiguous range of its region (the second form's stream: 16-byte loads, two records per lane)
        const uint2 *__restrict__ slab = hslab + (size_t)bin * plan.capb;           // (capb is even: the region starts 16-byte aligned)
        uint32_t b2 = min(begin, treg), e2 = min(end, treg);                          // the segment's part of the region
        if ((b2 & 1u) && b2 < e2) { if (threadIdx.x == 0) b2_add_record(acc, slab[b2]); b2++; }
        if ((e2 & 1u) && b2 < e2) { e2--; if (threadIdx.x == 0) b2_add_record(acc, slab[e2]); }
        const uint4 *__restrict__ slab2 = reinterpret_cast<const uint4 *>(slab);
        const uint32_t pend = e2 >> 1;
        constexpr int UNR = 4;
        const bool crowded = nseg > 1;
        uint32_t ib = b2 >> 1;
        for (; ib + UNR * 1024 <= pend; ib += UNR * 1024) {
            uint4 r[UNR];
#pragma unroll
            for (int u = 0; u < UNR; u++) r[u] = slab2[crowded ? ib + threadIdx.x * UNR + u : ib + u * 1024 + threadIdx.x];
#pragma unroll
            for (int u = 0; u < UNR; u++) {
                b2_add_record(acc, make_uint2(r[u].x, r[u].y));
                b2_add_record(acc, make_uint2(r[u].z, r[u].w));
            }
        }
        for (uint32_t i = ib + threadIdx.x; i < pend; i += 1024) {
            const uint4 r = slab2[i];
            b2_add_record(acc, make_uint2(r.x, r.y));
            b2_add_record(acc, make_uint2(r.z, r.w));
        }
        // some runs of this bin did not fit — its region is full (a sample distribution that crowds a few entries of a coarse hashed level) or a
        // block's staging area was: the bin's record index space continues behind the region with the spilled runs, where the point blocks left them
        if (end > treg) {                                                            // (block-uniform)
            __syncthreads();
            b3_walk_runs<B3_WALK>(acc, s_c, tiles, rt, pt, lvl_slab, nb, pb_first, max(begin, treg) - treg, end - treg);
        }
    } else {
        b3_walk_runs<B3_WALK>(acc, s_c, tiles, rt, pt, lvl_slab, nb, pb_first, begin, end);
    }
    __syncthreads();
    B3AdamConst adc = {0.0f, 0.0f, 0.0f, 0};
    if (ad.on) adc = b3_adam_const(ad);
    b3_flush(acc, lv, plan.p, slot, bin, nseg, gseg, grad_grid, partial, plan.dense_slot[slot] != 0xFFFFFFFFu ? 0xFFu : (uint32_t)plan.hbits[slot], ad, adc);
}

// ------------------------------------------------------------------------------------------------ host side
static inline uint64_t bn_align(uint64_t x) { return (x + 255) & ~(uint64_t)255; }

static void bn_plan(const GridLevels &lv, uint32_t nl, uint32_t B, BinPlan &plan) {
    plan.nb = cn_div_up(B, BN_PTS);
    uint32_t acc = 0;
    for (uint32_t l = 0; l < nl; l++) {
        plan.bin_first[l] = acc;
        acc += cn_div_up(lv.size[l], BN_CHUNK);
    }
    for (uint32_t l = nl; l <= GE_MAX_LEVELS; l++) plan.bin_first[l] = acc;
    plan.total_bins = acc;
}

static uint64_t bn_layout(const BinPlan &plan, uint32_t B, uint32_t nl, int dtype, BinWs *ws, void *base) {
    const uint64_t rec = dtype == CNERF_F16 ? 8 : 12;
    uint64_t off = 0;
    const uint64_t o_hist = off; off = bn_align(off + (uint64_t)plan.total_bins * plan.nb * 4);
    const uint64_t o_base = off; off = bn_align(off + (uint64_t)(plan.total_bins + 1) * 4);
    const uint64_t o_seg = off; off = bn_align(off + (uint64_t)(plan.total_bins + 1) * 4);
    const uint64_t o_rec = off; off = bn_align(off + (uint64_t)B * nl * 8 * rec);
    if (ws) {
        char *p = (char *)base;
        ws->hist = (uint32_t *)(p + o_hist);
        ws->bin_base = (uint32_t *)(p + o_base);
        ws->seg_first = (uint32_t *)(p + o_seg);
        ws->records = p + o_rec;
    }
    return off;
}


// ---- second form, host side
static int b2_env(const char *name, int dflt) { return cn_tune_env(name, dflt); }
static bool b2_enabled(int dtype) {
    static int v1 = -1;
    if (v1 < 0) v1 = b2_env("CNERF_BIN_V1", 0);
    return dtype == CNERF_F16 && !v1;
}

// points per block and level in the hist / emit sweeps
// Records per accumulate workgroup: 1/8 above the expected size of a hashed level's bin (4 pair records per sample over 128 bins), so
// that those bins keep one owner each (plain read-modify-write flush) while the crowded bins of the small dense levels split into
// workgroups of about the same length (uniform durations pack the last round of workgroups better: 0.40 -> 0.38 ms).
static uint32_t b2_seg(uint32_t B, uint32_t max_chunks = 128) {
    const uint64_t hashed_bin = (uint64_t)B * 4 / (max_chunks ? max_chunks : 1);
    const uint64_t s = hashed_bin + hashed_bin / 8;
    // small tables (few bins per level) would otherwise get a handful of million-record workgroups: cap, and let those bins split
    // ... and beyond 2^18 records a typical hashed bin is divided into EQUAL segments (round 6; the flat 2^17 cap gave every hashed bin of a 32768-ray
    // batch a second segment of a few thousand records — two 64 KiB partial images and a reduction per bin for nothing: 3.91 -> 3.80 ms per step;
    // one segment per bin at 65536 rays, on the other hand, costs the accumulate 13 %: profiles/r06_reduce_split_ab.txt)
    static const uint64_t cap = (uint64_t)b2_env("CNERF_B3_SEG_CAP", 1 << 18);
    if (s < B2_SEG_MIN) return B2_SEG_MIN;
    const uint64_t k = (s + cap - 1) / cap;
    return (uint32_t)((s + k - 1) / k);
}

static uint32_t b2_max_chunks(const Bin2Plan &plan, uint32_t nl) {
    uint32_t m = 1;
    for (uint32_t s = 0; s < nl; s++) m = m > plan.bin_first[s + 1] - plan.bin_first[s] ? m : plan.bin_first[s + 1] - plan.bin_first[s];
    return m;
}


static void b2_plan(const GridLevels &lv, uint32_t nl, uint32_t B, Bin2Plan &plan) {
    plan.nb = cn_div_up(B, B3_PTS);                 // (point blocks of the third form; the first form plans with BinPlan)
    uint32_t acc = 0;
    for (uint32_t s = 0; s < nl; s++) {
        plan.bin_first[s] = acc;
        acc += cn_div_up(lv.size[lv.order[s]], BN_CHUNK);
    }
    for (uint32_t s = nl; s <= GE_MAX_LEVELS; s++) plan.bin_first[s] = acc;
    plan.total_bins = acc;
}

// ---- third form, host side
struct Bin3Ws {
    uint32_t *runs, *pre, *cursor, *bin_base, *seg_first, *seg_bin, *split_list;
    float *adam_const;
    uint2 *hslab, *dslab;
    long long *partial;
    uint64_t max_seg, cursor_bytes;
};

// does ge_index take the strided (un-hashed, un-wrapped) sum on this level?  (ge_level_mode's GE_MODE_DENSE: every dimension's stride fits the
// table; with align_corners the grid has `resolution` lines per axis instead of `resolution + 1`, and its boundary corner wraps once)
static bool b3_is_dense(const GridLevels &lv, uint32_t level, int ac) {
    const uint64_t r1 = (uint64_t)lv.resolution[level] + (ac ? 0 : 1);
    return r1 * r1 * r1 <= lv.size[level];
}

// -> false when a level is neither dense nor a power-of-two hashed / wrapped level made of whole 4096-entry bins (or, round 6, of ONE bin of at
// most 4096 entries: the small tables): such tables take the first form
static bool b3_plan(const GridLevels &lv, uint32_t nl, uint32_t B, Bin3Plan &plan, uint32_t &n_dense, int ac) {
    b2_plan(lv, nl, B, plan.p);
    plan.p.nb = cn_div_up(B, B3_PTS);
    n_dense = 0;
    uint64_t cap = 0;
    bool ok = true;
    for (uint32_t s = 0; s < GE_MAX_LEVELS; s++) { plan.dense_slot[s] = 0xFFFFFFFFu; plan.hbits[s] = 0; }
    for (uint32_t s = 0; s < nl; s++) {
        const uint32_t level = lv.order[s], size = lv.size[level];
        if (b3_is_dense(lv, level, ac)) { plan.dense_slot[s] = n_dense++; continue; }
        const uint32_t nch = plan.p.bin_first[s + 1] - plan.p.bin_first[s];
        // interleaved bins need 2^k bins of exactly 4096 entries — or a single bin that IS the (power-of-two, >= 32-entry) level
        if ((size & (size - 1)) != 0 || (nch & (nch - 1)) != 0 || !(size == nch * BN_CHUNK || (nch == 1 && size >= 32))) ok = false;
        uint32_t hb = 0;
        while ((1u << hb) < nch) hb++;
        plan.hbits[s] = (uint8_t)hb;
        const uint64_t mean = (uint64_t)B * 17 / 4 / nch;                          // four pair records per sample (+ the unpaired ones: at most one in 16 as two singles), spread evenly by the interleave
        const uint64_t c = mean + mean / 2 + 8192;
        cap = cap > c ? cap : c;
    }
    plan.capb = (uint32_t)((cap + 1) & ~(uint64_t)1);
    return ok;
}

// the third form serves float16 records on tables of at most B2S_MAX_CHUNKS bins per level (the staging area's counters); larger tables
// (T = 2^20, 2^21: the reference's bear table) keep the second form with its histogram
// (gridtype 1 = tiled: its wrapping levels are x-contiguous, the interleave does not balance them — second form)
static bool b3_enabled(const GridLevels &lv, uint32_t nl, uint32_t B, int dtype, uint32_t gridtype, int ac) {
    static const int on = b2_env("CNERF_B3", 1);
    static const int wide_on = b2_env("CNERF_B3_WIDE", 1);                     // tuning builds: 0 = tables of more than 128 bins per level keep the second form
    (void)gridtype;                                                            // (round 6: tiled levels that wrap take the interleaved bins like hashed ones)
    if (!on || !b2_enabled(dtype)) return false;
    Bin3Plan plan;
    uint32_t nd;
    if (!b3_plan(lv, nl, B, plan, nd, ac)) return false;
    const uint32_t mc = b2_max_chunks(plan.p, nl);
    if (mc > (wide_on ? (uint32_t)B3_WIDE_CHUNKS : (uint32_t)B2S_MAX_CHUNKS)) return false;
    if ((uint64_t)plan.p.total_bins * plan.capb >= 0xF0000000ull) return false;           // record positions inside the bin-major slab stay 32-bit in the accumulate
    return (uint64_t)nl * plan.p.nb * B3_REGION < 0xF0000000ull && (uint64_t)B * 8 < 0x7FFFFFFFull;
}

static uint64_t b3_layout(const Bin3Plan &plan, uint32_t n_dense, uint32_t B, uint32_t nl, Bin3Ws *ws, void *base) {
    const Bin2Plan &p2 = plan.p;
    uint64_t off = 0;
    const uint64_t o_runs = off; off = bn_align(off + (uint64_t)p2.total_bins * p2.nb * 4);
    const uint64_t o_pre = off; off = bn_align(off + (uint64_t)p2.total_bins * p2.nb * 4);
    const uint64_t o_cur = off; off = bn_align(off + ((uint64_t)p2.total_bins + 2) * 4);       // (+ the partner word of a 64-bit pair reservation past the last bin)
    const uint64_t o_base = off; off = bn_align(off + (uint64_t)(p2.total_bins + 1) * 4);
    const uint64_t o_seg = off; off = bn_align(off + (uint64_t)(p2.total_bins + 1) * 4);
    const uint64_t o_split = off; off = bn_align(off + (uint64_t)(p2.total_bins + 1) * 4);
    const uint64_t o_adamc = off; off = bn_align(off + 16);                                   // B3AdamConst of this backward pass (k_bin_scan_bins)    // count + the bins that were split (k_bin_scan_bins -> k_bin3_reduce_split)
    const uint64_t h_records = (uint64_t)p2.total_bins * plan.capb;               // bin-major regions (the dense levels' bins leave theirs unused)
    const uint64_t d_records = (uint64_t)nl * p2.nb * B3_REGION;                   // the point blocks' private regions: every record of a dense level, the spill of a hashed one
    (void)n_dense;
    const uint64_t o_h = off; off = bn_align(off + h_records * 8);
    const uint64_t o_d = off; off = bn_align(off + d_records * 8);
    const uint64_t max_seg = (uint64_t)p2.total_bins + cn_div_up64((uint64_t)B * nl * 8, b2_seg(B, b2_max_chunks(p2, nl)));
    const uint64_t o_segbin = off; off = bn_align(off + max_seg * 4);
    const uint64_t o_part = off; off = bn_align(off + max_seg * BN_CHUNK * 2 * 8);
    if (ws) {
        char *p = (char *)base;
        ws->runs = (uint32_t *)(p + o_runs);
        ws->pre = (uint32_t *)(p + o_pre);
        ws->cursor = (uint32_t *)(p + o_cur);
        ws->cursor_bytes = (uint64_t)p2.total_bins * 4;
        ws->bin_base = (uint32_t *)(p + o_base);
        ws->seg_first = (uint32_t *)(p + o_seg);
        ws->split_list = (uint32_t *)(p + o_split);
        ws->adam_const = (float *)(p + o_adamc);
        ws->hslab = (uint2 *)(p + o_h);
        ws->dslab = (uint2 *)(p + o_d);
        ws->seg_bin = (uint32_t *)(p + o_segbin);
        ws->partial = (long long *)(p + o_part);
        ws->max_seg = max_seg;
    }
    return off;
}

__global__ void __launch_bounds__(256) k_bin3_zero(uint32_t *__restrict__ p, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0;
}

// cnerf_grid_backward_adam: the optimiser step armed for the next backward pass into `g` (one shot)
static CnerfGridAdam g_grid_adam;
static bool g_grid_adam_armed = false;
static int g_grid_adam_consumed = 0;
void bn_grid_adam_arm(const CnerfGridAdam *cfg) {
    g_grid_adam_armed = cfg != nullptr;
    if (cfg) { g_grid_adam = *cfg; g_grid_adam_consumed = 0; }
}
int bn_grid_adam_consumed() { const int c = g_grid_adam_consumed; g_grid_adam_consumed = 0; return c; }
// any backward pass into the armed gradient table disarms; only one that covers the WHOLE table through the third form applies the step
static B3Adam b3_take_adam(float *gemb, const GridLevels &lv, uint32_t nl, bool third_form) {
    B3Adam ad = {};
    if (!g_grid_adam_armed || g_grid_adam.g != gemb) return ad;
    g_grid_adam_armed = false;
    const uint64_t covered = ((uint64_t)lv.offset[nl - 1] + lv.size[nl - 1]) * 2;
    if (!third_form || covered != g_grid_adam.n) return ad;
    ad.p = g_grid_adam.p; ad.m = g_grid_adam.m; ad.v = g_grid_adam.v; ad.ph = reinterpret_cast<__half *>(g_grid_adam.p_half);
    ad.state = g_grid_adam.scaler_state;
    ad.lr = g_grid_adam.lr; ad.beta1 = g_grid_adam.beta1; ad.beta2 = g_grid_adam.beta2; ad.eps = g_grid_adam.eps; ad.extra_inv = g_grid_adam.extra_inv;
    ad.zero_grad = g_grid_adam.zero_grad; ad.on = 1;
    g_grid_adam_consumed = 1;
    return ad;
}

static int b3_backward(const __half *grad, const float *inputs, const GridLevels &lv, float *gemb, uint32_t B, uint32_t nl, uint32_t gridtype, int ac,
                       uint32_t interp, void *workspace, hipStream_t st) {
    if (nl == 0) return CNERF_OK;
    Bin3Plan plan;
    uint32_t n_dense;
    if (!b3_plan(lv, nl, B, plan, n_dense, ac)) return CNERF_EINVAL;
    Bin3Ws ws;
    b3_layout(plan, n_dense, B, nl, &ws, workspace);
    static const int emit_pad = cn_tune_env("CNERF_B3_EMIT_LDS_PAD", 0);        // tuning builds: extra LDS bytes per emit workgroup (occupancy experiments)
    const uint32_t lds_n = B3_CAP * 9 + B2S_MAX_CHUNKS * 12 + 16, lds_w = B3_CAP * 8 + B3_WIDE_CHUNKS * 12 + 16;
    const uint32_t emit_lds = (lds_n > lds_w ? lds_n : lds_w) + (uint32_t)emit_pad, acc_lds = BN_CHUNK * 2 * sizeof(long long) + 32 + B3_MAXT * 8;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bin3_emit), hipFuncAttributeMaxDynamicSharedMemorySize, emit_lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bin3_accum), hipFuncAttributeMaxDynamicSharedMemorySize, acc_lds);
        attr_set = true;
    }
    const Bin2Plan &p2 = plan.p;
    const uint32_t seg = b2_seg(B, b2_max_chunks(p2, nl));
    B3Adam ad = b3_take_adam(gemb, lv, nl, true);
    ad.cst = ws.adam_const;
    hipLaunchKernelGGL(k_bin3_zero, dim3(cn_div_up(p2.total_bins, 256)), dim3(256), 0, st, ws.cursor, p2.total_bins);    // (a kernel, not a memset node: hipGraph capture)
    cn_stage(0, st);
    hipLaunchKernelGGL(k_bin3_emit, dim3(p2.nb * nl), dim3(B3_THREADS), emit_lds, st, grad, inputs, lv, plan, ws.runs, ws.cursor, ws.hslab, ws.dslab, B, gridtype, ac,
                       interp, gemb, nl, (uint32_t)b2_env("CNERF_B3_EMIT_ABL", 0), g_cn_found_inf);
    cn_stage(1, st);
    hipLaunchKernelGGL(k_bin3_totals, dim3(p2.total_bins), dim3(BN_SCAN_THREADS), 0, st, (const uint32_t *)ws.runs, ws.pre, (const uint32_t *)ws.cursor, ws.bin_base,
                       plan, nl);
    static const uint32_t rs_bins = (uint32_t)cn_tune_env("CNERF_B3_RS_BINS", B3_RS_BINS);
    static const int rs_list = cn_tune_env("CNERF_B3_RSLIST", 1);                 // tuning builds: 0 = one workgroup column per bin (profiles/r06_reduce_split_ab.txt)
    hipLaunchKernelGGL(k_bin_scan_bins, dim3(1), dim3(1024), 0, st, ws.bin_base, ws.bin_base, ws.seg_first, p2.total_bins, seg, ws.seg_bin, ws.split_list, ad.on,
                       ad.state, ad.lr, ad.beta1, ad.beta2, ad.extra_inv, ad.on ? ws.adam_const : (float *)nullptr);
    hipLaunchKernelGGL(k_bin3_accum, dim3((uint32_t)ws.max_seg), dim3(1024), acc_lds, st, (const uint2 *)ws.hslab, (const uint2 *)ws.dslab, (const uint32_t *)ws.runs,
                       (const uint32_t *)ws.pre, (const uint32_t *)ws.cursor, (const uint32_t *)ws.bin_base, (const uint32_t *)ws.seg_first, lv, plan, gemb, ws.partial,
                       (const uint32_t *)ws.seg_bin, nl, seg, (uint32_t)b2_env("CNERF_B3_ONLY", 0), ad);
    cn_stage(2, st);
    hipLaunchKernelGGL(k_bin3_reduce_split, dim3(!rs_list || p2.total_bins < rs_bins ? p2.total_bins : rs_bins, BN_CHUNK * 2 / 4 / B3_RS_GROUPS), dim3(256), 0, st,
                       (const long long *)ws.partial, ws.seg_first, lv, plan, gemb, nl, rs_list ? (const uint32_t *)ws.split_list : (const uint32_t *)nullptr, ad);
    cn_stage(3, st);
#ifdef CNERF_TUNING
    static const int dbg = cn_tune_env("CNERF_B3_DEBUG", 0);                     // tuning builds: split statistics of every dbg-th call (synchronises)
    static int dbg_calls = 0;
    if (dbg && (++dbg_calls % dbg) == 0) {
        uint32_t n_split = 0, n_seg = 0;
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(&n_split, ws.split_list, 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&n_seg, ws.seg_first + p2.total_bins, 4, hipMemcpyDeviceToHost);
        std::vector<uint32_t> sf(p2.total_bins + 1), bb(p2.total_bins + 1);
        (void)hipMemcpy(sf.data(), ws.seg_first, sf.size() * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(bb.data(), ws.bin_base, bb.size() * 4, hipMemcpyDeviceToHost);
        fprintf(stderr, "[b3] call %d: bins %u segments %u split bins %u seg %u | per slot (bins, segments, records, max bin):", dbg_calls, p2.total_bins, n_seg, n_split, seg);
        for (uint32_t sl = 0; sl < nl; sl++) {
            uint32_t mx = 0;
            for (uint32_t b_ = p2.bin_first[sl]; b_ < p2.bin_first[sl + 1]; b_++) mx = std::max(mx, bb[b_ + 1] - bb[b_]);
            fprintf(stderr, " L%u(%u,%u,%u,%u)", lv.order[sl], p2.bin_first[sl + 1] - p2.bin_first[sl], sf[p2.bin_first[sl + 1]] - sf[p2.bin_first[sl]],
                    bb[p2.bin_first[sl + 1]] - bb[p2.bin_first[sl]], mx);
        }
        fprintf(stderr, "\n");
    }
#endif
    return cn_launch_status();
}

// does the binned backward of this shape use a plan prepared ahead of time (histogram + scans on the sample coordinates)?  The third form
// counts inside its emit kernel: nothing to prepare.
// (the query has no align_corners argument: a plan is "needed" unless the third form takes the shape either way — a plan prepared for a launch
// that then does not use it is wasted work, never wrong)
bool bn_needs_plan(uint32_t B, uint32_t nl, const GridLevels &lv, int dtype, uint32_t gridtype) {
    return !(b3_enabled(lv, nl, B, dtype, gridtype, 0) && b3_enabled(lv, nl, B, dtype, gridtype, 1));
}

// used by gridencoder.hip
bool bn_eligible(uint32_t B, uint32_t D, uint32_t C, uint32_t nl, const GridLevels &lv) {
    if (D != 3 || C != 2 || nl == 0) return false;
    for (uint32_t l = 0; l < nl; l++)
        if (cn_div_up(lv.size[l], BN_CHUNK) > BN_MAX_CHUNKS || (lv.size[l] & 7)) return false;
    return (uint64_t)B * nl * 8 < 0xF0000000ull;          // record positions are 32-bit
}

// (the size does not depend on the grid type — the entry point that asks has none — so it covers whichever form the launch will take)
uint64_t bn_workspace_bytes(uint32_t B, uint32_t nl, const GridLevels &lv, int dtype) {
    uint64_t b3 = 0;
    for (int ac = 0; ac < 2; ac++) {                                             // (the query has no align_corners argument either: the larger of the two layouts)
        if (!b3_enabled(lv, nl, B, dtype, 0u, ac)) continue;
        Bin3Plan p3;
        uint32_t nd;
        b3_plan(lv, nl, B, p3, nd, ac);
        const uint64_t b = b3_layout(p3, nd, B, nl, nullptr, nullptr);
        b3 = b > b3 ? b : b3;
    }
    BinPlan plan;
    bn_plan(lv, nl, B, plan);
    const uint64_t b1 = bn_layout(plan, B, nl, dtype, nullptr, nullptr);
    return b1 > b3 ? b1 : b3;
}

// phase 1 (needs the sample coordinates only): histogram + scans.  phase 2 (needs the gradients): emit + accumulate.
// The two phases may be issued separately (cnerf_grid_encode_backward_prepare) so that phase 1 overlaps the forward / field backward.
static int bn_phase1(const float *inputs, const GridLevels &lv, uint32_t B, uint32_t nl, uint32_t gridtype, int ac, uint32_t interp, int dtype,
                     void *workspace, hipStream_t st) {
    BinPlan plan;
    bn_plan(lv, nl, B, plan);
    BinWs ws;
    bn_layout(plan, B, nl, dtype, &ws, workspace);
    const dim3 grid1(plan.nb * nl);
    hipLaunchKernelGGL(k_bin_hist, grid1, dim3(BN_THREADS), 0, st, inputs, lv, plan, ws.hist, B, gridtype, ac, interp);
    hipLaunchKernelGGL(k_bin_scan_blocks, dim3(plan.total_bins), dim3(BN_SCAN_THREADS), 0, st, ws.hist, ws.bin_base, plan.nb);
    hipLaunchKernelGGL(k_bin_scan_bins, dim3(1), dim3(1024), 0, st, ws.bin_base, ws.bin_base, ws.seg_first, plan.total_bins, BN_SEG, (uint32_t *)nullptr);
    return cn_launch_status();
}

template <typename T>
static int bn_phase2(const T *grad, const float *inputs, const GridLevels &lv, float *gemb, uint32_t B, uint32_t nl, uint32_t gridtype, int ac,
                     uint32_t interp, int dtype, void *workspace, hipStream_t st) {
    BinPlan plan;
    bn_plan(lv, nl, B, plan);
    BinWs ws;
    bn_layout(plan, B, nl, dtype, &ws, workspace);
    const dim3 grid1(plan.nb * nl);
    hipLaunchKernelGGL((k_bin_emit<T>), grid1, dim3(BN_THREADS), 0, st, grad, inputs, lv, plan, ws.hist, ws.bin_base, (BinRec<T> *)ws.records, B,
                       gridtype, ac, interp, gemb, g_cn_found_inf);
    // upper bound of accumulate workgroups: every bin may add one partial segment
    const uint64_t max_seg = (uint64_t)plan.total_bins + cn_div_up64((uint64_t)B * nl * 8, BN_SEG);
    const uint32_t lds_bytes = BN_CHUNK * 2 * sizeof(typename BinAcc<T>::type) + 16;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bin_accum<T>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        attr_set = true;
    }
    hipLaunchKernelGGL((k_bin_accum<T>), dim3((uint32_t)max_seg), dim3(1024), lds_bytes, st, (const BinRec<T> *)ws.records,
                       ws.bin_base, ws.seg_first, lv, plan, gemb);
    return cn_launch_status();
}

// The plan in pieces (cnerf_grid_encode_backward_prepare_rows / _finish) was the second form's: with that form gone (round 6) no shape offers it —
// bn_hist_block_points() = 0 makes the entry points report *prepared = 0 before they get here.
int bn_prepare_rows(const float *, const GridLevels &, uint32_t, uint32_t, uint32_t, int, uint32_t, int, void *, hipStream_t, uint32_t, uint32_t) { return CNERF_EINVAL; }
int bn_prepare_finish(const GridLevels &, uint32_t, uint32_t, int, void *, hipStream_t) { return CNERF_EINVAL; }
uint32_t bn_hist_block_points(int) { return 0u; }

int bn_prepare(const float *inputs, const GridLevels &lv, uint32_t B, uint32_t nl, uint32_t gridtype, int ac, uint32_t interp, int dtype,
               void *workspace, hipStream_t st) {
    return bn_phase1(inputs, lv, B, nl, gridtype, ac, interp, dtype, workspace, st);
}

int bn_backward(const void *grad, const float *inputs, const GridLevels &lv, float *gemb, uint32_t B, uint32_t nl, uint32_t gridtype, int ac,
                uint32_t interp, int dtype, void *workspace, hipStream_t st, bool prepared) {
    if (b3_enabled(lv, nl, B, dtype, gridtype, ac)) return b3_backward((const __half *)grad, inputs, lv, gemb, B, nl, gridtype, ac, interp, workspace, st);
    if (nl) (void)b3_take_adam(gemb, lv, nl, false);                                 // (the first form does not carry the optimiser step: disarm)
    if (!prepared) {
        const int rc = bn_phase1(inputs, lv, B, nl, gridtype, ac, interp, dtype, workspace, st);
        if (rc) return rc;
    }
    if (dtype == CNERF_F16) return bn_phase2<__half>((const __half *)grad, inputs, lv, gemb, B, nl, gridtype, ac, interp, dtype, workspace, st);
    return bn_phase2<float>((const float *)grad, inputs, lv, gemb, B, nl, gridtype, ac, interp, dtype, workspace, st);
}
