// Fused attention forward for the UNet (self-attention over 4096 / 1024 / 256 / 64 tokens, cross-attention over 77): softmax(q k^T / sqrt(d)) v per
// head without materialising the score matrix (537 MB per layer at 64x64 latents otherwise).
// One wave owns 32 queries, lane = query: the transposed score tile S^T = K Q^T comes out of v_mfma_f32_32x32x16_f16 with the
// 32 keys of the tile spread over the lane's 16 accumulator registers (x 2 half-waves), so the online-softmax row statistics are
// per-lane scalars (one cross-half exchange per key tile for the running maximum), and the probabilities — still in their C
// registers — are directly the B fragments of the second product O^T += V^T P^T (same K-slot permutation trick as the NeRF
// field kernels, field_common.h).  V is consumed either transposed ([C][tokens], produced by cnerf_sd_transpose: the cross-attention values,
// which depend on the prompt only and are transposed once per prompt) or — round 4, VROW — as it comes out of the projection ([tokens][C]):
// its key tile is staged row-major in LDS and the V^T fragments are transposing reads (ds_read_b64_tr_b16) of it, which removes the
// k_transpose launch in front of every self-attention.  K and V^T tiles are read through L2 (they are shared by all query
// blocks of a head); head dims 40 / 80 / 160 are padded to the 16- / 32-wide MFMA shapes with zero fragments.
#include "common.h"
#include "../../include/customnerf_sd.h"

typedef _Float16 at_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 at_h4 __attribute__((ext_vector_type(4)));
typedef float at_f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int at_rho(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

typedef short at_s4 __attribute__((__vector_size__(4 * sizeof(short))));
// row pitch (halves) of the row-major V tile: 32 DT channels, padded so that the pitch is 64 bytes modulo 128 — the four key rows a transposing
// read touches then fall into four different 64-byte bank quarters (conflict-free)
__host__ __device__ constexpr uint32_t at_vrow_halves(int dt) { return (32u * dt * 2u) % 128u == 64u ? 32u * dt : 32u * dt + 32u; }

// exchange between the two half-waves on the vector ALU (v_permlane32_swap: no LDS round trip in the middle of the softmax chain)
__device__ __forceinline__ float at_half_max(float x) {
    const uint32_t u = __builtin_bit_cast(uint32_t, x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);          // r[0] = (low half, low half), r[1] = (high half, high half)
    return fmaxf(__builtin_bit_cast(float, (uint32_t)r[0]), __builtin_bit_cast(float, (uint32_t)r[1]));
}

// KS = ceil(d / 16) k-steps of the score product, DT = ceil(d / 32) output tiles, KT = 32-key sub-tiles per iteration (one workgroup barrier, one
// running-maximum update and one staging round per 32 KT keys); VROW: V is [tokens][C] (ldv = its row pitch), else V^T [C][tokens]
template <int KS, int DT, int KT, bool VROW>
__global__ void __launch_bounds__(256) k_sd_attention(const _Float16 *__restrict__ Q, const _Float16 *__restrict__ K, const _Float16 *__restrict__ VT,
                                                      _Float16 *__restrict__ O, uint32_t H, uint32_t Tq, uint32_t Tk, uint32_t d, uint32_t ldq, uint64_t sq,
                                                      uint32_t ldk, uint64_t sk, uint32_t ldv, uint64_t sv, uint32_t ldo, uint64_t so, float scale_log2e, int causal) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hi = lane >> 5, li = lane & 31;
    // workgroups are dealt to the 8 XCDs round-robin by linear id: renumber so that the query blocks of one (batch, head) run on ONE XCD and its
    // K / V stream (0.65–1.7 MB of lines at 4096 keys) is served by that XCD's 4 MB L2 instead of by every L2 at once
    uint32_t bx = blockIdx.x, by = blockIdx.y;
    {
        const uint32_t total = gridDim.x * gridDim.y;
        if (total % CN_NXCD == 0) {
            const uint32_t id = blockIdx.y * gridDim.x + blockIdx.x, nid = (id % CN_NXCD) * (total / CN_NXCD) + id / CN_NXCD;
            by = nid / gridDim.x;
            bx = nid - by * gridDim.x;
        }
    }
    const uint32_t b = by / H, h = by - b * H;
    const uint32_t q0 = (bx * 4 + wave) * 32;
    const uint32_t qi = q0 + li;
    const bool q_ok = qi < Tq;                               // waves past the last query keep running (workgroup barriers below), results unused
    const _Float16 *qrow = Q + sq * b + (size_t)qi * ldq + h * d;
    const _Float16 *kbase = K + sk * b + h * d;
    const _Float16 *vbase = VROW ? VT + sv * b + h * d : VT + sv * b + (size_t)(h * d) * ldv;

    // Q fragments (B operand of S^T = K Q^T): lane = query, k = 16 s + 8 hi + j
    at_h8 qf[KS];
#pragma unroll
    for (int s = 0; s < KS; s++) {
        const uint32_t c = 16 * s + 8 * hi;
        at_h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (q_ok && c < d) v = *reinterpret_cast<const at_h8 *>(qrow + c);          // d % 8 == 0
        qf[s] = v;
    }
    at_f16v o[DT];
#pragma unroll
    for (int t = 0; t < DT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) o[t][r] = 0.0f;
    float m = -INFINITY, l = 0.0f;                     // running max in raw score units (identical in both half-waves) and this half's partial sum
    // Head dims below the padded 32 DT channels (40 -> 64, 80 -> 96) leave a spare zero channel in the V tile: it carries ones, so that the softmax
    // denominator is one more row of O^T += V^T P^T (rescaled with it) instead of 32 conversions + adds per key tile on the vector ALU.
    const bool sum_mfma = d < 32u * DT;

    // The four waves of a workgroup share the K / V^T tiles of their (batch, head) through LDS, stored in MFMA fragment order
    // ([fragment][lane] x 16 B: conflict-free ds_read_b128), double-buffered; every thread stages N_ITEMS / 256 16-byte items per tile.
    // (Each wave reading its own fragments from L2 cost 4x the L2 traffic: 1.8 GB per 4096-token layer.)
    constexpr int NFK = KT * KS, NFV = KT * 2 * DT;    // fragments per iteration: of K, of V^T (2 per 32-channel tile and sub-tile)
    constexpr uint32_t VROWH = at_vrow_halves(DT), VCH = 4 * DT;                  // VROW: row pitch of the V tile, 16-byte chunks per row
    constexpr uint32_t KEYS = 32 * KT;
    constexpr int N_ITEMS = VROW ? NFK * 64 + (int)(KEYS * VCH) : (NFK + NFV) * 64;
    constexpr int PER = (N_ITEMS + 255) / 256;         // staged 16-byte items per thread
    __shared__ __attribute__((aligned(16))) at_h8 tile[2][VROW ? NFK * 64 : (NFK + NFV) * 64];
    __shared__ __attribute__((aligned(16))) _Float16 vtile[2][VROW ? KEYS * VROWH : 8];
    // causal (CLIP text encoder): query q attends to keys <= q; the workgroup's key range ends with its last query
    const uint32_t q_end = min(bx * 128 + 128, Tq);
    const uint32_t n_kt = causal ? min((Tk + KEYS - 1) / KEYS, (q_end + KEYS - 1) / KEYS) : (Tk + KEYS - 1) / KEYS;
    // register staging two iterations deep: a tile's loads are issued two iterations before its LDS commit (one iteration is ~0.5 us of work, an
    // L2 / MALL round trip more than that: with one tile of lookahead every iteration waited for its loads)
    at_h8 stage_a[PER], stage_b[PER];
    auto fetch = [&](uint32_t kt, at_h8 (&stage)[PER]) __attribute__((always_inline)) {
        const uint32_t key0 = kt * KEYS;
#pragma unroll
        for (int it = 0; it < PER; it++) {
            const uint32_t item = it * 256 + threadIdx.x, f = item >> 6, ln = item & 63, fl = ln & 31, fh = ln >> 5;
            at_h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (f < (uint32_t)NFK) {
                const uint32_t u = f / KS, s = f - u * KS, krow = key0 + 32 * u + fl, c = 16 * s + 8 * fh;
                if (krow < Tk && c < d) v = *reinterpret_cast<const at_h8 *>(kbase + (size_t)krow * ldk + c);
            } else if (VROW) {
                const uint32_t j = item - NFK * 64, row = j / VCH, c = 8 * (j - row * VCH);
                if (j < KEYS * VCH && key0 + row < Tk && c < d) v = *reinterpret_cast<const at_h8 *>(vbase + (size_t)(key0 + row) * ldv + c);
                else if (sum_mfma && c == d) v[0] = (_Float16)1.0f;                       // the ones channel (masked keys have p = 0)
            } else if (f < (uint32_t)(NFK + NFV)) {
                const uint32_t g = f - NFK, u = g / (2 * DT), g2 = g - u * 2 * DT, t = g2 >> 1, s2 = g2 & 1, dd = 32 * t + fl;
                if (dd < d && key0 + 32 * u < Tk) {                                    // V^T rows are padded to whole 32-key tiles (at_launch checks ldv)
                    const _Float16 *vp = vbase + (size_t)dd * ldv + key0 + 32 * u + 4 * fh + 16 * s2;
                    const at_h4 x = *reinterpret_cast<const at_h4 *>(vp), y = *reinterpret_cast<const at_h4 *>(vp + 8);
                    v = at_h8{x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
                } else if (sum_mfma && dd == d) {
                    v = at_h8{1, 1, 1, 1, 1, 1, 1, 1};                                  // the ones channel
                }
            }
            stage[it] = v;
        }
    };
    auto commit = [&](uint32_t buf, const at_h8 (&stage)[PER]) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < PER; it++) {
            const uint32_t item = it * 256 + threadIdx.x;
            if (VROW) {
                if (item < (uint32_t)NFK * 64) tile[buf][item] = stage[it];
                else if (item < (uint32_t)N_ITEMS) {
                    const uint32_t j = item - NFK * 64, row = j / VCH, c = 8 * (j - row * VCH);
                    *reinterpret_cast<at_h8 *>(&vtile[buf][row * VROWH + c]) = stage[it];
                }
            } else if (item < (uint32_t)(NFK + NFV) * 64) tile[buf][item] = stage[it];
        }
    };
    fetch(0, stage_a);
    commit(0, stage_a);
    if (1 < n_kt) fetch(1, stage_b);
    __syncthreads();
    // one iteration: `st_same` held tile kt (committed an iteration ago) and takes tile kt + 2, `st_next` holds tile kt + 1
    auto key_tile = [&](uint32_t kt, at_h8 (&st_same)[PER], const at_h8 (&st_next)[PER]) __attribute__((always_inline)) {
        const uint32_t key0 = kt * KEYS, buf = kt & 1;
        if (kt + 2 < n_kt) fetch(kt + 2, st_same);
        // ---- S^T sub-tiles: rows = keys (A operand: lane = key li), cols = queries
        at_f16v sacc[KT];
#pragma unroll
        for (int u = 0; u < KT; u++) {
#pragma unroll
            for (int r = 0; r < 16; r++) sacc[u][r] = 0.0f;
#pragma unroll
            for (int s = 0; s < KS; s++) sacc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(tile[buf][(u * KS + s) * 64 + lane], qf[s], sacc[u], 0, 0, 0);
        }
        // ---- online softmax over this lane's 16 KT keys (+ the partner half's).  The running maximum is kept in raw score units and the
        // 1/sqrt(d) log2(e) scale rides on the exponent's fused multiply-add: max3 + fma + exp2 + packed convert per key, nothing else.
        float tmax = -INFINITY;
        if (!causal && key0 + KEYS <= Tk) {                      // interior tile (workgroup-uniform): no masking work
#pragma unroll
            for (int u = 0; u < KT; u++)
#pragma unroll
                for (int r = 0; r < 16; r++) tmax = fmaxf(tmax, sacc[u][r]);
        } else {
#pragma unroll
            for (int u = 0; u < KT; u++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const uint32_t key = key0 + 32 * u + at_rho(r, hi);
                    sacc[u][r] = (key < Tk && !(causal && key > qi)) ? sacc[u][r] : -INFINITY;
                    tmax = fmaxf(tmax, sacc[u][r]);
                }
        }
        tmax = at_half_max(tmax);
        const float m_new = fmaxf(fmaxf(m, tmax), -1e30f);   // stays finite even when a causal tile holds no key for this query yet
        const bool moved = m_new != m;
        const float corr = moved ? __builtin_amdgcn_exp2f((m - m_new) * scale_log2e) : 1.0f;
        m = m_new;
        const float nm = -m_new * scale_log2e;
        float psum = 0.0f;
        at_h8 pf[KT][2];
#pragma unroll
        for (int u = 0; u < KT; u++)
#pragma unroll
            for (int s = 0; s < 2; s++) {
                at_h8 f;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float p = __builtin_amdgcn_exp2f(cn_fma(sacc[u][8 * s + j], scale_log2e, nm));       // v_exp_f32: exp2(-inf) = 0 for masked keys
                    const _Float16 ph = (_Float16)p;
                    f[j] = ph;
                    if (!sum_mfma) psum += (float)ph;    // the denominator sums what the numerator uses
                }
                pf[u][s] = f;
            }
        if (!sum_mfma) l = l * corr + psum;
        if (__any(moved)) {                                      // the running maxima settle after a few tiles: skip the rescale when no lane moved
#pragma unroll
            for (int t = 0; t < DT; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) o[t][r] *= corr;
        }
        // ---- O^T += V^T P^T : A = V^T fragment (lane = channel row, K-slots = keys in C-register order: two runs of 4 keys)
#pragma unroll
        for (int u = 0; u < KT; u++)
#pragma unroll
            for (int t = 0; t < DT; t++)
#pragma unroll
                for (int s = 0; s < 2; s++) {
                    at_h8 vf;
                    if (VROW) {
                        // lane (channel 32 t + (l & 31), half hi) wants keys 4 hi + 16 s + {0..3} and + 8: two transposing reads of the row-major tile.
                        // Address role of lane i in its 16-lane group G: key row i >> 2, chunk i & 3 of the group's 16 channels.
                        const uint32_t i16 = lane & 15, G = (lane >> 4) & 1;
                        const _Float16 *vp = &vtile[buf][(32 * u + 4 * hi + 16 * s + (i16 >> 2)) * VROWH + 32 * t + 16 * G + 4 * (i16 & 3)];
                        union { at_h8 h; at_s4 q[2]; } f;
                        f.q[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((at_s4 __attribute__((address_space(3))) *)vp);
                        f.q[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((at_s4 __attribute__((address_space(3))) *)(vp + 8 * VROWH));
                        vf = f.h;
                    } else {
                        vf = tile[buf][(NFK + u * 2 * DT + 2 * t + s) * 64 + lane];
                    }
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[u][s], o[t], 0, 0, 0);
                }
        if (kt + 1 < n_kt) commit(buf ^ 1, st_next);
        __syncthreads();
    };
    for (uint32_t kt = 0; kt < n_kt; kt += 2) {
        key_tile(kt, stage_a, stage_b);
        if (kt + 1 < n_kt) key_tile(kt + 1, stage_b, stage_a);
    }
    if (sum_mfma) {                                   // channel d sits in C row d % 32 = register 4 ((d % 32) / 8) of the low half-wave, column = query
        float ls = 0.0f;
#pragma unroll
        for (int t = 0; t < DT; t++)
#pragma unroll
            for (int g = 0; g < 4; g++)
                if ((uint32_t)(32 * t + 8 * g) == d) ls = o[t][4 * g];
        l = __shfl(ls, li, 64);
    } else {
        l += __shfl_xor(l, 32, 64);
    }
    const float inv = 1.0f / l;
    if (!q_ok) return;
    _Float16 *orow = O + so * b + (size_t)qi * ldo + h * d;
#pragma unroll
    for (int t = 0; t < DT; t++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const uint32_t dd = 32 * t + 8 * g + 4 * hi;            // rows rho(4g .. 4g+3, hi) are 4 consecutive channels
            if (dd < d) {
                at_h4 v;
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = (_Float16)(o[t][4 * g + e] * inv);
                *reinterpret_cast<at_h4 *>(orow + dd) = v;          // d % 4 == 0
            }
        }
}

// ---- LDS-DMA form for row-major V and the head dims the UNet / CLIP encoder use (40, 64, 80, 160: compile-time D) -----------------------------
// The register-staged loop above spends a third of its time in its fetch: per-item bounds branches, and a compiler-placed vmcnt(0) in front of the
// first MFMA of every iteration (the two-deep staging registers are written under exec masks, so every load is waited for where it is issued).
// Here K fragments and V rows go global -> LDS directly (global_load_lds_dwordx4, no staging registers, no ds_write pass): every wave issues NIW
// branch-free instructions per iteration — a lane whose source lies outside the problem (key >= Tk, pad channels) reads a 16-byte constant of zeros,
// the lane of the ones channel reads {1, 0, ...} — into a ring of three LDS buffers, two iterations ahead, with a counted vmcnt before the iteration's
// one barrier.  The V tile is row-major with a pitch of D / 8 (+ 1: the ones channel) 16-byte chunks; the transposing fragment reads of channels
// beyond the pitch run into the next key row — finite values in accumulator rows that are never stored.
// compile-time loops (immediate offsets of the inline-assembly LDS reads below)
#include <utility>
template <typename F, int... Is>
__device__ __forceinline__ void at_static_for_impl(F &&f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void at_static_for(F &&f) { at_static_for_impl(static_cast<F &&>(f), std::make_integer_sequence<int, N>{}); }

// The transposing reads of the DMA kernel are inline assembly: the compiler orders every LDS read it can see behind ALL outstanding LDS-DMA
// (s_waitcnt vmcnt(0) — it cannot tell the ring's buffers apart), which would make each iteration wait for the batch it has just issued.
template <int OFF>
__device__ __forceinline__ at_s4 at_tr_read(uint32_t addr) {
    at_s4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
__device__ __forceinline__ void at_lds_wait(at_s4 &a, at_s4 &b, at_s4 &c, at_s4 &d) {      // the reads have landed before anything consumes them
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

__device__ __attribute__((aligned(16))) const _Float16 at_const[16] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0};

template <int D, int KT>
__global__ void __launch_bounds__(256) k_sd_attention_dma(const _Float16 *__restrict__ Q, const _Float16 *__restrict__ K, const _Float16 *__restrict__ V,
                                                          _Float16 *__restrict__ O, uint32_t H, uint32_t Tq, uint32_t Tk, uint32_t ldq, uint64_t sq,
                                                          uint32_t ldk, uint64_t sk, uint32_t ldv, uint64_t sv, uint32_t ldo, uint64_t so, float scale_log2e, int causal) {
    constexpr int KS = (D + 15) / 16, DT = (D + 31) / 32;
    constexpr bool ONES = D < 32 * DT;                          // spare channel: the softmax denominator rides on the PV product
    constexpr uint32_t PCH = D / 8 + (ONES ? 1 : 0);            // 16-byte chunks per V row in LDS
    constexpr uint32_t KEYS = 32 * KT;
    constexpr int NFK = KT * KS;                                // K fragments (1 KiB each) per iteration
    constexpr int NVI = (KEYS * PCH + 63) / 64;                 // DMA instructions of the V tile
    constexpr int NI = NFK + NVI, NIW = (NI + 3) / 4;           // per iteration, per wave (slots beyond NI write zeros to a dump area)
    constexpr uint32_t KB = NFK * 1024, VB = NVI * 1024 + 128;  // bytes per buffer (V: whole instructions + over-read slack)
    constexpr uint32_t BUF = KB + VB;
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * BUF + 1024];
    unsigned char *dump = lds + 3 * BUF;

    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hi = lane >> 5, li = lane & 31;
    uint32_t bx = blockIdx.x, by = blockIdx.y;                  // one (batch, head) per XCD at a time (see k_sd_attention)
    {
        const uint32_t total = gridDim.x * gridDim.y;
        if (total % CN_NXCD == 0) {
            const uint32_t id = blockIdx.y * gridDim.x + blockIdx.x, nid = (id % CN_NXCD) * (total / CN_NXCD) + id / CN_NXCD;
            by = nid / gridDim.x;
            bx = nid - by * gridDim.x;
        }
    }
    const uint32_t b = by / H, h = by - b * H;
    const uint32_t q0 = (bx * 4 + wave) * 32;
    const uint32_t qi = q0 + li;
    const bool q_ok = qi < Tq;
    const _Float16 *qrow = Q + sq * b + (size_t)qi * ldq + h * D;
    const _Float16 *kbase = K + sk * b + h * D;
    const _Float16 *vbase = V + sv * b + h * D;

    at_h8 qf[KS];
#pragma unroll
    for (int s = 0; s < KS; s++) {
        const uint32_t c = 16 * s + 8 * hi;
        at_h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (q_ok && c < (uint32_t)D) v = *reinterpret_cast<const at_h8 *>(qrow + c);
        qf[s] = v;
    }
    at_f16v o[DT];
#pragma unroll
    for (int t = 0; t < DT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) o[t][r] = 0.0f;
    float m = -INFINITY, l = 0.0f;

    // ---- this wave's DMA slots: item = wave + 4 i.  Per lane: source pointer at key tile 0, its advance per iteration, the key row it reads (for
    // the Tk test), the constant it reads instead when the source is outside the problem; per slot (uniform): the LDS destination inside a buffer.
    const _Float16 *src[NIW];
    const _Float16 *alt[NIW];
    uint32_t rowk[NIW], step[NIW], dst[NIW];
    const uint32_t wave_u = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
    for (int i = 0; i < NIW; i++) {
        const uint32_t item = wave_u + 4 * i;
        alt[i] = at_const;
        if (item < (uint32_t)NFK) {                              // K fragment f = u KS + s: lane (fl, fh) holds K[32 u + fl][16 s + 8 fh ..]
            const uint32_t u = item / KS, sfrag = item - u * KS, c = 16 * sfrag + 8 * hi;
            rowk[i] = c < (uint32_t)D ? 32 * u + li : 0xFFFFFFFFu;
            src[i] = kbase + (size_t)(32 * u + li) * ldk + c;
            step[i] = KEYS * ldk;
            dst[i] = item * 1024;
        } else if (item < (uint32_t)NI) {                        // V rows: 16-byte chunk idx = row PCH + ch of the row-major tile
            const uint32_t idx = (item - NFK) * 64 + lane, row = idx / PCH, ch = idx - row * PCH;
            const bool data = row < KEYS && ch < (uint32_t)D / 8;
            rowk[i] = data ? row : 0xFFFFFFFFu;
            if (ONES && row < KEYS && ch == (uint32_t)D / 8) alt[i] = at_const + 8;
            src[i] = vbase + (size_t)row * ldv + 8 * ch;
            step[i] = KEYS * ldv;
            dst[i] = KB + (item - NFK) * 1024;
        } else {                                                 // filler slot: every wave issues NIW instructions per iteration (counted vmcnt)
            rowk[i] = 0xFFFFFFFFu;
            src[i] = at_const;
            step[i] = 0;
            dst[i] = 0xFFFFFFFFu;
        }
    }
    for (uint32_t i = threadIdx.x; i < 3 * 128 / 4; i += 256) {  // the over-read slack behind each V tile
        const uint32_t bsel = i / 32, w = i - bsel * 32;
        reinterpret_cast<uint32_t *>(lds + bsel * BUF + KB + NVI * 1024)[w] = 0;
    }
    auto issue = [&](uint32_t kt, uint32_t boff) __attribute__((always_inline)) {
        const uint32_t key0 = kt * KEYS;
#pragma unroll
        for (int i = 0; i < NIW; i++) {
            const bool ok = rowk[i] != 0xFFFFFFFFu && key0 + rowk[i] < Tk;
            const _Float16 *p = ok ? src[i] + (size_t)kt * step[i] : alt[i];
            unsigned char *dp = dst[i] == 0xFFFFFFFFu ? dump : lds + boff + dst[i];
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p, (__attribute__((address_space(3))) void *)dp, 16, 0, 0);
        }
    };
    const uint32_t q_end = min(bx * 128 + 128, Tq);
    const uint32_t n_kt = causal ? min((Tk + KEYS - 1) / KEYS, (q_end + KEYS - 1) / KEYS) : (Tk + KEYS - 1) / KEYS;
    // per-lane bases of the fragment reads
    const uint32_t i16 = lane & 15, G = (lane >> 4) & 1;
    const uint32_t vlane = (4 * hi + (i16 >> 2)) * PCH * 16 + (16 * G + 4 * (i16 & 3)) * 2;
    const uint32_t lds_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds;

    issue(0, 0);
    if (1 < n_kt) issue(1, BUF);
    uint32_t b_cur = 0, b_nxt = BUF, b_nx2 = 2 * BUF;
    for (uint32_t kt = 0; kt < n_kt; kt++) {
        // tile kt has landed (this wave's share: at most the newest batch is still in flight), then everybody's; the barrier also says that
        // every wave is done reading tile kt - 1, whose buffer the next batch overwrites
        if (kt + 1 < n_kt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 2 < n_kt) issue(kt + 2, b_nx2);
        const uint32_t key0 = kt * KEYS;
        const at_h8 *ktile = reinterpret_cast<const at_h8 *>(lds + b_cur);
        // ---- S^T sub-tiles
        at_f16v sacc[KT];
#pragma unroll
        for (int u = 0; u < KT; u++) {
#pragma unroll
            for (int r = 0; r < 16; r++) sacc[u][r] = 0.0f;
#pragma unroll
            for (int s = 0; s < KS; s++) sacc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ktile[(u * KS + s) * 64 + lane], qf[s], sacc[u], 0, 0, 0);
        }
        // ---- online softmax (as in k_sd_attention)
        float tmax = -INFINITY;
        if (!causal && key0 + KEYS <= Tk) {
#pragma unroll
            for (int u = 0; u < KT; u++)
#pragma unroll
                for (int r = 0; r < 16; r++) tmax = fmaxf(tmax, sacc[u][r]);
        } else {
#pragma unroll
            for (int u = 0; u < KT; u++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const uint32_t key = key0 + 32 * u + at_rho(r, hi);
                    sacc[u][r] = (key < Tk && !(causal && key > qi)) ? sacc[u][r] : -INFINITY;
                    tmax = fmaxf(tmax, sacc[u][r]);
                }
        }
        tmax = at_half_max(tmax);
        const float m_new = fmaxf(fmaxf(m, tmax), -1e30f);
        const bool moved = m_new != m;
        const float corr = moved ? __builtin_amdgcn_exp2f((m - m_new) * scale_log2e) : 1.0f;
        m = m_new;
        const float nm = -m_new * scale_log2e;
        float psum = 0.0f;
        at_h8 pf[KT][2];
#pragma unroll
        for (int u = 0; u < KT; u++)
#pragma unroll
            for (int s = 0; s < 2; s++) {
                at_h8 f;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float p = __builtin_amdgcn_exp2f(cn_fma(sacc[u][8 * s + j], scale_log2e, nm));
                    const _Float16 ph = (_Float16)p;
                    f[j] = ph;
                    if (!ONES) psum += (float)ph;
                }
                pf[u][s] = f;
            }
        if (!ONES) l = l * corr + psum;
        if (__any(moved)) {
#pragma unroll
            for (int t = 0; t < DT; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) o[t][r] *= corr;
        }
        // ---- O^T += V^T P^T: transposing reads of the row-major tile (lane i of its 16-lane group: key row i >> 2, chunk i & 3 of 16 channels)
        const uint32_t va = lds_addr + b_cur + KB + vlane;
        at_static_for<KT>([&](auto U) {
            constexpr int u = decltype(U)::value;
            at_s4 fr[DT][2][2];
            at_static_for<DT>([&](auto T) {
                constexpr int t = decltype(T)::value;
                at_static_for<2>([&](auto S2) {
                    constexpr int s2 = decltype(S2)::value;
                    constexpr int off = (32 * u + 16 * s2) * (int)PCH * 16 + 64 * t;
                    fr[t][s2][0] = at_tr_read<off>(va);
                    fr[t][s2][1] = at_tr_read<off + 8 * (int)PCH * 16>(va);
                });
            });
#pragma unroll
            for (int t = 0; t < DT; t++) {
                at_lds_wait(fr[t][0][0], fr[t][0][1], fr[t][1][0], fr[t][1][1]);
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++) {
                    union { at_h8 h; at_s4 q[2]; } f;
                    f.q[0] = fr[t][s2][0];
                    f.q[1] = fr[t][s2][1];
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.h, pf[u][s2], o[t], 0, 0, 0);
                }
            }
        });
        const uint32_t tb = b_cur;
        b_cur = b_nxt; b_nxt = b_nx2; b_nx2 = tb;
    }
    if (ONES) {                                       // channel D sits in C row D % 32 = register 4 ((D % 32) / 8) of the low half-wave, column = query
        l = __shfl(o[D / 32][4 * ((D % 32) / 8)], li, 64);
    } else {
        l += __shfl_xor(l, 32, 64);
    }
    const float inv = 1.0f / l;
    if (!q_ok) return;
    _Float16 *orow = O + so * b + (size_t)qi * ldo + h * D;
#pragma unroll
    for (int t = 0; t < DT; t++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const uint32_t dd = 32 * t + 8 * g + 4 * hi;
            if (dd < (uint32_t)D) {
                at_h4 v;
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = (_Float16)(o[t][4 * g + e] * inv);
                *reinterpret_cast<at_h4 *>(orow + dd) = v;
            }
        }
}

static bool at_use_dma() { static const int v = cn_tune_env("CNERF_ATTN_DMA", 1); return v != 0; }

template <bool VROW>
static int at_launch(const void *q, const void *k, const void *vT, void *out, uint32_t B, uint32_t H, uint32_t Tq, uint32_t Tk, uint32_t d, uint32_t ldq,
                     uint64_t sq, uint32_t ldk, uint64_t sk, uint32_t ldv, uint64_t sv, uint32_t ldo, uint64_t so, int causal, void *stream) {
    if (B == 0 || H == 0 || Tq == 0 || Tk == 0 || d == 0 || (d & 7) || d > 160) return CNERF_EINVAL;
    if (VROW) {
        if ((ldq & 7) || (ldk & 7) || (ldv & 7) || (ldo & 3) || (sv & 7)) return CNERF_EINVAL;                     // V rows are read in 16-byte chunks
    } else if ((ldq & 7) || (ldk & 7) || (ldv & 3) || (ldo & 3) || ldv < ((Tk + 31) / 32) * 32) return CNERF_EINVAL;  // V^T rows are read in whole 32-key tiles
    if ((sq & 7) || (sk & 7) || (sv & 3) || (so & 3)) return CNERF_EINVAL;
    if (!q || !k || !vT || !out) return CNERF_ENULL;
    if ((((uintptr_t)q) | ((uintptr_t)k)) & 15 || (((uintptr_t)vT) | ((uintptr_t)out)) & 7) return CNERF_EINVAL;
    if (VROW && (((uintptr_t)vT) & 15)) return CNERF_EINVAL;
    const dim3 grid(cn_div_up(Tq, 128), B * H), block(256);
    const float scale_log2e = 1.4426950408889634f / sqrtf((float)d);
    hipStream_t st = CN_STREAM(stream);
#define AT_LAUNCH(KS, DT, KT)                                                                                                                       \
    hipLaunchKernelGGL((k_sd_attention<KS, DT, KT, VROW>), grid, block, 0, st, (const _Float16 *)q, (const _Float16 *)k, (const _Float16 *)vT, (_Float16 *)out, H, Tq, Tk, \
                       d, ldq, sq, ldk, sk, ldv, sv, ldo, so, scale_log2e, causal)
#define AT_LAUNCH_DMA(D, KT)                                                                                                                         \
    hipLaunchKernelGGL((k_sd_attention_dma<D, KT>), grid, block, 0, st, (const _Float16 *)q, (const _Float16 *)k, (const _Float16 *)vT, (_Float16 *)out, H, Tq, Tk, \
                       ldq, sq, ldk, sk, ldv, sv, ldo, so, scale_log2e, causal)
    if (VROW && at_use_dma()) {                                  // row-major V at the head dims of the UNet / the text encoder: the LDS-DMA form
        if (d == 40) { AT_LAUNCH_DMA(40, 2); return cn_launch_status(); }
        if (d == 64) { AT_LAUNCH_DMA(64, 2); return cn_launch_status(); }
        if (d == 80) { AT_LAUNCH_DMA(80, 1); return cn_launch_status(); }
        if (d == 160) { AT_LAUNCH_DMA(160, 1); return cn_launch_status(); }
    }
#undef AT_LAUNCH_DMA
    const uint32_t ks = (d + 15) / 16, dt = (d + 31) / 32;
    // 64 keys per iteration where the registers allow two waves per SIMD with it (head dims <= 64), 32 above
    if (ks <= 3 && dt <= 2) AT_LAUNCH(3, 2, 2);
    else if (ks <= 4 && dt <= 2) AT_LAUNCH(4, 2, 2);
    else if (ks <= 5 && dt <= 3) AT_LAUNCH(5, 3, 1);
    else AT_LAUNCH(10, 5, 1);
#undef AT_LAUNCH
    return cn_launch_status();
}

extern "C" {

int cnerf_sd_attention(const void *q, const void *k, const void *vT, void *out, uint32_t B, uint32_t H, uint32_t Tq, uint32_t Tk, uint32_t d, uint32_t ldq,
                       uint64_t sq, uint32_t ldk, uint64_t sk, uint32_t ldv, uint64_t sv, uint32_t ldo, uint64_t so, int causal, void *stream) {
    return at_launch<false>(q, k, vT, out, B, H, Tq, Tk, d, ldq, sq, ldk, sk, ldv, sv, ldo, so, causal, stream);
}

int cnerf_sd_attention_v(const void *q, const void *k, const void *v, void *out, uint32_t B, uint32_t H, uint32_t Tq, uint32_t Tk, uint32_t d, uint32_t ldq,
                         uint64_t sq, uint32_t ldk, uint64_t sk, uint32_t ldv, uint64_t sv, uint32_t ldo, uint64_t so, int causal, void *stream) {
    return at_launch<true>(q, k, v, out, B, H, Tq, Tk, d, ldq, sq, ldk, sk, ldv, sv, ldo, so, causal, stream);
}

}  // extern "C"
