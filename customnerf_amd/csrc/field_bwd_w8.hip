// Fused field backward for gfx950 (fp16, enc_pad == 32), round 4: k_field_bwd_w8 — TWO busy waves per SIMD.
//
// Why a new kernel: k_field_bwd_x2 (field_bwd_x2.hip) keeps the 24 persistent 32x32 weight-gradient tiles of a wave pair in the registers of
// the two waves that also run the MLP chains — 246 VGPRs + 256 AGPRs per lane, ONE wave per SIMD, and a lone wave issues 43 % of the time
// (profiles/r03_field_bwd_pmc.txt: vector ALU 31 %, matrix pipe 29 %, LDS 13 %, nothing overlapping).  Here a tile stream is a four-stage
// pipeline over FOUR waves, every one below 256 registers, two per SIMD, each carrying a comparable share of chain AND weight-gradient work:
//
//   Af  geometry forward   tile i in phase i       n0, n1, n2                                   20 MFMA 16x16x32 + dW_n2 (tile i - 3), dW_rO, dW_dO (tile i - 2)  (8 MFMA 32x32x16)
//   Bf  heads forward      tile i in phase i + 1   d0 | r0, dO | rO, output-layer gradients      24 MFMA 16x16x32 + dW_d0 (tile i - 2)                              (4)
//   Bb  heads backward     tile i in phase i + 2   dO^T | rO^T, d0^T + r0^T -> dz_3              24 MFMA 16x16x32 + dW_r0 (same tile)                               (6)
//   Ab  geometry backward  tile i in phase i + 3   n2^T, n1^T, n0^T -> d/d(grid features)        20 MFMA 16x16x32 + dW_n1, dW_n0 (same tile)                        (6)
// The forward stages are the short ones, so they take the weight-gradient products whose operands are rings written in EARLIER phases:
// those MFMAs do not depend on the wave's own chain and fill the matrix pipe while its first fetches are in flight.
//
// (The first version of this file had two chain waves and two "shadow" waves that only accumulated weight gradients: correct, but the
// shadows issued 50 instructions per phase against 340 of a chain wave — still one busy wave per SIMD, 470 us against the 450 us of
// k_field_bwd_x2.)  Every wave's weight-gradient operands are images it published itself in the same phase (DS operations of one wave
// execute in order: no barrier, single buffer) or images an earlier stage published in an earlier phase (ring buffers below, one
// workgroup barrier per phase).  The tile is SIXTEEN samples: the images of a 32-sample tile with these lifetimes, for two streams, do
// not fit beside the weights; at 16 samples they take 112 KiB + 44 KiB of weight fragments.  Chains therefore run on
// v_mfma_f32_16x16x32_f16 (16 outputs x 16 samples x 32 inputs), the weight gradients on v_mfma_f32_32x32x16_f16 with the 16 samples of a
// tile on the contraction index: one MFMA per 32x32 tile of dW per sample tile.
//
// Layouts (lane l: n = l & 15, g = l >> 4):
//   chain MFMA   A = weights: lane (row n, g) holds W[16 m + n][K-slots 8 g .. 8 g + 7]; B = activations: lane (sample n, g) holds the same
//                K-slots of its sample; C: lane (sample n, g) holds output rows 16 m + 4 g + r, r < 4.  The C registers of output tiles
//                2 s, 2 s + 1 ARE the next layer's B fragment of K-step s: K-slot (s, g, j) := feature 32 s + 16 (j >> 2) + 4 g + (j & 3)
//                ("C order"); the weight fragments are staged in that column order once per workgroup.  Grid features and direction
//                features arrive in natural order: K-slot (s, g, j) = feature 32 s + 8 g + j.
//   W^T          (backward chain) fragments are transposing reads (ds_read_b64_tr_b16) of the forward fragment store: four consecutive
//                output rows sit 16 bytes apart in one (m, s) block, the sixteen input features of a 16-lane group in four 8-byte chunks.
//   images       [K-step][g][n] 16-byte slots = the B fragments as the chain holds them, the slot's bits 6..7 XORed with g (conflict-free
//                transposing reads); a weight-gradient operand = "feature slot f = 8 g + j" of 8 consecutive samples, two transposing reads.
//
// OUTCOME (round 4, profiles/r04_field_bwd_w8.txt): bit-for-bit deterministic, parity green (tests/test_gpu_field.py with the kernel switched on),
// 1.78-1.91 waves per SIMD by SQ_WAVE_CYCLES, no scratch, 256 VGPRs / 0 AGPRs — and 430 us per 2.1 M samples standalone against 450 us of
// k_field_bwd_x2, 2.17 against 2.15 ms per training step in place: NOT faster.  The 16-sample tile doubles the LDS instruction count per
// sample (one weight fragment per MFMA, two 8-byte transposing reads per W^T fragment — 8-byte reads from one or two waves per SIMD reach a
// fraction of the LDS rate), the LDS is busy 45 % of the time with 24 % of that in bank conflicts, and every stage is still a chain of
// dependent LDS round trips: three of the four waves take ~3 k cycles per 16-sample phase for ~220 instructions each.  The release library
// therefore keeps k_field_bwd_x2; this kernel is compiled into tuning builds only (make TUNING=1, CNERF_FIELD_W8_BWD=1).
#include "field_bwd_common.h"

#ifdef CNERF_TUNING
typedef float w8_f4 __attribute__((ext_vector_type(4)));
typedef short w8_s4 __attribute__((__vector_size__(4 * sizeof(short))));

#define W8_THREADS 512
#define W8_TILE 16
#define W8_K 1024                                  // one K-step image: 64 lanes x 16 B (32 features x 16 samples)

// weight fragment store (halves): n0 4x1, n1 4x2, n2 4x2, d0 4x2, dO 1x2, r0 4x3, rO 1x2 blocks of 512 halves
#define W8_W_N0 0
#define W8_W_N1 (W8_W_N0 + 4 * 512)
#define W8_W_N2 (W8_W_N1 + 8 * 512)
#define W8_W_D0 (W8_W_N2 + 8 * 512)
#define W8_W_DO (W8_W_D0 + 8 * 512)
#define W8_W_R0 (W8_W_DO + 2 * 512)
#define W8_W_RO (W8_W_R0 + 12 * 512)
#define W8_W_END (W8_W_RO + 2 * 512)               // 22528 halves = 44 KiB

// per-stream LDS (bytes, dynamic).  Ring depths follow the lifetimes: written in phase w, last read in phase r -> r - w + 1 buffers.
#define W8_AF 0                                    // 4 x 5 KiB  x0 1, h1 2, h2 2: Af (phase i) -> Ab (phase i + 3)
#define W8_AF_X0 0
#define W8_AF_H1 (1 * W8_K)
#define W8_AF_H2 (3 * W8_K)
#define W8_AF_BYTES (5 * W8_K)
#define W8_FEA (W8_AF + 4 * W8_AF_BYTES)           // 4 x 2 KiB  Af (i) -> Bf (i + 1), Bb's dW (i + 2), Bf's dW_d0 (i + 3)
#define W8_BF (W8_FEA + 4 * 2 * W8_K)              // 2 x 7 KiB  dir 1, hd 2, hr 2, bro 1, bdo 1: Bf (i + 1) -> Bb, Af's dW (i + 2)
#define W8_BF_DIR 0
#define W8_BF_HD (1 * W8_K)
#define W8_BF_HR (3 * W8_K)
#define W8_BF_BRO (5 * W8_K)
#define W8_BF_BDO (6 * W8_K)
#define W8_BF_BYTES (7 * W8_K)
#define W8_Z3 (W8_BF + 2 * W8_BF_BYTES)            // 2 x 2 KiB  Bb (i + 2) -> Ab, Af's dW_n2 (i + 3)
#define W8_ZD (W8_Z3 + 2 * 2 * W8_K)               // 2 x 2 KiB  Bb (i + 2) -> Bf's dW_d0 (i + 3)
#define W8_BB (W8_ZD + 2 * 2 * W8_K)               // 2 KiB      zr: private to Bb
#define W8_BB_ZR 0
#define W8_AB (W8_BB + 2 * W8_K)                   // 4 KiB      z2 2, z1 2: private to Ab
#define W8_AB_Z2 0
#define W8_AB_Z1 (2 * W8_K)
#define W8_SCR (W8_AB + 4 * W8_K)                  // 256 B: direction-feature scratch of Bf
#define W8_PIPE_BYTES (W8_SCR + 256)               // 57 600 B per stream

struct W8Off {
    uint32_t n0, n1, n2, d0, dO, r0, rO, total;
};
__host__ __device__ __forceinline__ W8Off w8_offsets(const FieldDims &dm) {
    W8Off o;
    uint32_t p = 0;
    o.n0 = p; p += FLD_HID * dm.enc_pad;
    o.n1 = p; p += (dm.n_hidden_geo == 2) ? 4096 : 0;
    o.n2 = p; p += 4096;
    o.d0 = p; p += 4096;
    o.dO = p; p += 16 * 64;
    o.r0 = p; p += 64 * 96;
    o.rO = p; p += 16 * 64;
    o.total = p;
    return o;
}

__device__ __forceinline__ w8_f4 w8_mfma16(cn_h8 a, cn_h8 b, w8_f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ cn_f16v w8_mfma32(cn_h8 a, cn_h8 b, cn_f16v c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

__host__ __device__ __forceinline__ int w8_col_c(int s, int g, int j) { return 32 * s + 16 * (j >> 2) + 4 * g + (j & 3); }     // C order
__host__ __device__ __forceinline__ int w8_col_n(int s, int g, int j) { return 32 * s + 8 * g + j; }                         // natural order

// forward-order staging.  KIND 0: natural columns, 1: C-order columns, 2: the colour layer ([27 direction features | 64 fea] parameter
// columns; K-steps 0..1 = fea in C order, K-step 2 = the direction features in natural order)
template <int KIND>
__device__ __forceinline__ void w8_stage_layer(_Float16 *dst, const float *__restrict__ W, uint32_t rows, uint32_t in_stride, uint32_t M, uint32_t S,
                                               uint32_t n_valid_cols) {
    const uint32_t total = M * S * 512;
    for (uint32_t i = threadIdx.x; i < total; i += W8_THREADS) {
        const uint32_t j = i & 7, lane = (i >> 3) & 63, ms = i >> 9;
        const uint32_t s = ms % S, m = ms / S;
        const uint32_t row = 16 * m + (lane & 15), g = lane >> 4;
        int col;
        if (KIND == 0) col = w8_col_n(s, g, j);
        else if (KIND == 1) col = w8_col_c(s, g, j);
        else {
            if (s < 2) col = FLD_NDIR + w8_col_c(s, g, j);
            else {
                col = w8_col_n(0, g, j);
                if (col >= FLD_NDIR) col = -1;
            }
        }
        float v = 0.0f;
        if (row < rows && col >= 0 && (uint32_t)col < n_valid_cols) v = W[(size_t)row * in_stride + col];
        // swizzle: row bit 3 ^= g >> 1.  The four slot groups g_f = 0..3 a transposing read touches are 256 bytes apart (same banks); with
        // the XOR they alias in pairs only (two-way instead of four-way conflicts), and the forward 16-byte reads stay conflict-free
        dst[(i & ~(8u << 3)) | ((i ^ ((g >> 1) << 6)) & (8u << 3))] = (_Float16)v;
    }
}

__device__ __forceinline__ w8_s4 w8_tr(const unsigned char *p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((w8_s4 __attribute__((address_space(3))) *)p);
}
__device__ __forceinline__ cn_h8 w8_pack(w8_s4 lo, w8_s4 hi) {
    union { cn_h8 h; w8_s4 s[2]; } f;
    f.s[0] = lo; f.s[1] = hi;
    return f.h;
}

// Weight fragments are fetched into registers one stage ahead of the MFMAs that use them (explicit prefetch: a lone chain wave would
// otherwise wait for every LDS read right in front of its MFMA).  Fragment order of every loader: f[s * M + m].
// forward order: W(m, s) of a layer staged with S K-steps per output tile
template <int M, int S, int S_USED>
__device__ __forceinline__ void w8_load_F(const unsigned char *wf, uint32_t lane16, cn_h8 *f) {
#pragma unroll
    for (int s = 0; s < S_USED; s++)
#pragma unroll
        for (int m = 0; m < M; m++) f[s * M + m] = *reinterpret_cast<const cn_h8 *>(wf + (m * S + s) * 1024 + lane16);
}
// W^T(mp, sp): input tile mp (16 input features in natural numbering), K-step sp = output rows 32 sp .. + 31 in C order.  Lane role in its
// 16-lane group (i = l & 15): row i >> 2 of four consecutive output rows, chunk q = i & 3 of the sixteen input features.  C-order input
// columns (every layer but n0): input feature 16 mp + i sits in K-step mp >> 1 at slot (g_f = i >> 2, j = (i & 3) + 4 (mp & 1)), i.e. chunk
// q = slot group q of the (m, s) block, byte 8 (mp & 1) of the 16-byte slot.
template <int MP, int SP, int S>
__device__ __forceinline__ void w8_load_T(const unsigned char *wf, uint32_t lane_off, cn_h8 *f) {
#pragma unroll
    for (int sp = 0; sp < SP; sp++)
#pragma unroll
        for (int mp = 0; mp < MP; mp++) {
            const uint32_t c = (mp >> 1) * 1024 + 8 * (mp & 1);
            f[sp * MP + mp] = w8_pack(w8_tr(wf + lane_off + (2 * sp * S) * 1024 + c), w8_tr(wf + lane_off + ((2 * sp + 1) * S) * 1024 + c));
        }
}
// the same for a 16-row output layer (dO, rO): one output tile, K-slots j >= 4 are empty
template <int MP, int S>
__device__ __forceinline__ void w8_load_T16(const unsigned char *wf, uint32_t lane_off, cn_h8 *f) {
    const w8_s4 zero = {0, 0, 0, 0};
#pragma unroll
    for (int mp = 0; mp < MP; mp++) f[mp] = w8_pack(w8_tr(wf + lane_off + (mp >> 1) * 1024 + 8 * (mp & 1)), zero);
}
// natural input columns (n0, S = 1): input feature 16 mp + i = slot (g_f = 2 mp + (i >> 3), j = i & 7): chunk q -> g_f = 2 mp + (q >> 1), byte
// 8 (q & 1); the store's swizzle depends on g_f >> 1 = mp: one lane offset per mp
template <int MP, int SP>
__device__ __forceinline__ void w8_load_Tn(const unsigned char *wf, uint32_t lane_off_n0, uint32_t lane_off_n1, cn_h8 *f) {
#pragma unroll
    for (int sp = 0; sp < SP; sp++)
#pragma unroll
        for (int mp = 0; mp < MP; mp++) {
            const uint32_t lo = mp ? lane_off_n1 : lane_off_n0;
            f[sp * MP + mp] = w8_pack(w8_tr(wf + lo + (2 * sp) * 1024 + mp * 512), w8_tr(wf + lo + (2 * sp + 1) * 1024 + mp * 512));
        }
}
// acc[m] += f(m, s) b[s]
template <int M, int S>
__device__ __forceinline__ void w8_mm(const cn_h8 *f, const cn_h8 *b, w8_f4 (&acc)[M]) {
#pragma unroll
    for (int s = 0; s < S; s++)
#pragma unroll
        for (int m = 0; m < M; m++) acc[m] = w8_mfma16(f[s * M + m], b[s], acc[m]);
}

__device__ __forceinline__ uint32_t w8_cvt_pk(float a, float b) {
    const cn_h2 v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ uint32_t w8_pk_relu(uint32_t v) {
    uint32_t r;
    asm("v_pk_max_i16 %0, %1, 0" : "=v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ uint32_t w8_pk_posmask(uint32_t act) {      // 0xffff per half of `act` (post-ReLU halves) that is > 0 (see field_bwd_x2.hip)
    uint32_t t, m;
    asm("v_pk_sub_i16 %0, 0, %1" : "=v"(t) : "v"(act));
    asm("v_pk_ashrrev_i16 %0, %1, %2" : "=v"(m) : "v"(0x000F000Fu), "v"(t));
    return m;
}
// C registers of four 16-row tiles -> the two B fragments of the next layer.  MODE 0: plain, 1: ReLU, 2: masked with [act > 0]
template <int MODE>
__device__ __forceinline__ void w8_c_to_b(const w8_f4 (&acc)[4], const cn_h8 *act, cn_h8 *b) {
#pragma unroll
    for (int s = 0; s < 2; s++) {
        union { cn_h8 h; uint32_t w[4]; } f, a;
        if (MODE == 2) a.h = act[s];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const w8_f4 &c = acc[2 * s + (k >> 1)];
            uint32_t v = w8_cvt_pk(c[2 * (k & 1)], c[2 * (k & 1) + 1]);
            if (MODE == 1) v = w8_pk_relu(v);
            if (MODE == 2) v &= w8_pk_posmask(a.w[k]);
            f.w[k] = v;
        }
        b[s] = f.h;
    }
}
template <int M>
__device__ __forceinline__ void w8_zero(w8_f4 (&acc)[M]) {
#pragma unroll
    for (int m = 0; m < M; m++) acc[m] = w8_f4{0.0f, 0.0f, 0.0f, 0.0f};
}

// image slot of lane (g, n) in K-step s
__device__ __forceinline__ uint32_t w8_slot(uint32_t lane) { return ((lane >> 4) << 8) + (((lane & 15) << 4) ^ ((lane >> 4) << 6)); }
template <int NS>
__device__ __forceinline__ void w8_publish(unsigned char *img, uint32_t slot, const cn_h8 *x) {
#pragma unroll
    for (int s = 0; s < NS; s++) *reinterpret_cast<cn_h8 *>(img + s * W8_K + slot) = x[s];
}
template <int NS>
__device__ __forceinline__ void w8_fetch(const unsigned char *img, uint32_t slot, cn_h8 *x) {
#pragma unroll
    for (int s = 0; s < NS; s++) x[s] = *reinterpret_cast<const cn_h8 *>(img + s * W8_K + slot);
}
// Operand of a weight-gradient MFMA (32x32x16, the tile's 16 samples on the contraction index) out of K-step u of an image: lane (hi, f)
// receives samples 8 hi .. 8 hi + 7 of feature slot f = 8 g_f + j_f.  Address role of lane i in its 16-lane group G: sample row i >> 2,
// chunk q = i & 3 -> slot g_f = 2 (G & 1) + (q >> 1), byte 8 (q & 1).
struct W8Op { uint32_t c0, c1; };
__device__ __forceinline__ W8Op w8_lane_off_op(uint32_t l) {
    const uint32_t G = l >> 4, i = l & 15, ra = i >> 2, q = i & 3, hi = G >> 1;
    const uint32_t gf = 2 * (G & 1) + (q >> 1);
    W8Op o;
    o.c0 = (gf << 8) + ((((8 * hi + ra) << 4)) ^ (gf << 6)) + 8 * (q & 1);
    o.c1 = (gf << 8) + ((((8 * hi + 4 + ra) << 4)) ^ (gf << 6)) + 8 * (q & 1);
    return o;
}
__device__ __forceinline__ cn_h8 w8_op(const unsigned char *img, W8Op lo, int u) { return w8_pack(w8_tr(img + u * W8_K + lo.c0), w8_tr(img + u * W8_K + lo.c1)); }
// the gradient-side operand of a tile outside the stream's range is forced to zero with a wave-uniform select: control flow around the
// accumulating MFMAs makes the compiler copy the accumulator tiles (160 v_mov per phase)
__device__ __forceinline__ cn_h8 w8_op_if(bool on, const unsigned char *img, W8Op lo, int u) {
    union { cn_h8 h; uint32_t w[4]; } f;
    f.h = w8_op(img, lo, u);
#pragma unroll
    for (int k = 0; k < 4; k++) f.w[k] = on ? f.w[k] : 0u;
    return f.h;
}

__device__ __forceinline__ void w8_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// measurement aid (CNERF_W8_ABLATE bit 5, tuning builds): per wave, cycles spent working / waiting at the phase barrier
#define W8_T0() const unsigned long long t0_ = (ablate & 32) ? __builtin_readcyclecounter() : 0ull
#define W8_T1() do { if (ablate & 32) { const unsigned long long t1_ = __builtin_readcyclecounter(); w8_barrier(); \
                                        const unsigned long long t2_ = __builtin_readcyclecounter(); tw_ += t1_ - t0_; tb_ += t2_ - t1_; } else w8_barrier(); } while (0)

// one 32x32 weight-gradient tile -> its place in the partial row.  Rows = feature slots of the Z operand (lane index of the MFMA's A side: the C
// layout's row), columns = feature slots of the activation operand (lane & 31).  ZC / AC: slot -> feature in C order (else natural).
template <bool ZC, bool AC>
__device__ __forceinline__ void w8_store(float *__restrict__ part, uint32_t dst, uint32_t stride, uint32_t col0, uint32_t M, uint32_t N, uint32_t uz, uint32_t ua,
                                         uint32_t lane, const cn_f16v &acc) {
    const uint32_t fa = lane & 31, hi = lane >> 5;
    const uint32_t col = AC ? (uint32_t)w8_col_c(ua, fa >> 3, fa & 7) : 32 * ua + fa;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const uint32_t fz = (uint32_t)fld_rho(r, (int)hi);
        const uint32_t row = ZC ? (uint32_t)w8_col_c(uz, fz >> 3, fz & 7) : 32 * uz + fz;
        if (row < M && col < N) part[dst + (size_t)row * stride + col0 + col] = acc[r];
    }
}
__device__ __forceinline__ void w8_zero16(cn_f16v &a) {
#pragma unroll
    for (int r = 0; r < 16; r++) a[r] = 0.0f;
}

// ================================================================================================ the kernel
template <int NGEO>
__global__ void __launch_bounds__(W8_THREADS) k_field_bwd_w8(const void *__restrict__ enc, const float *__restrict__ xyz, const float *__restrict__ dirs,
                                                             uint32_t dir_group, uint32_t P_, FieldDims dm, const float *__restrict__ pnet,
                                                             const float *__restrict__ pden, const float *__restrict__ prgb,
                                                             const float *__restrict__ g_sigma, const float *__restrict__ g_rgbc,
                                                             void *__restrict__ grad_enc, float *__restrict__ partials, uint32_t ablate_arg) {
#ifdef CNERF_TUNING
    const uint32_t ablate = ablate_arg;                       // measurement aid, tuning builds only: 1 A-bwd, 2 A-fwd, 4 B, 8 A', 16 B' switched off (results wrong)
#else
    constexpr uint32_t ablate = 0;
    (void)ablate_arg;
#endif
    __shared__ __attribute__((aligned(16))) _Float16 w8_w[W8_W_END];
    extern __shared__ __attribute__((aligned(16))) unsigned char w8_lds[];
    const W8Off po = w8_offsets(dm);
    const uint32_t in_r0 = FLD_HID + FLD_DIR;
    const float *n0 = pnet, *n1 = pnet + FLD_HID * dm.enc_pad;
    const float *n2 = n1 + (NGEO == 2 ? FLD_HID * FLD_HID : 0);
    const float *d0 = pden, *dO = pden + FLD_HID * FLD_HID;
    const float *r0 = prgb, *rO = prgb + FLD_HID * in_r0;

    w8_stage_layer<0>(w8_w + W8_W_N0, n0, FLD_HID, dm.enc_pad, 4, 1, dm.enc_pad);
    if (NGEO == 2) w8_stage_layer<1>(w8_w + W8_W_N1, n1, FLD_HID, FLD_HID, 4, 2, FLD_HID);
    w8_stage_layer<1>(w8_w + W8_W_N2, n2, FLD_HID, FLD_HID, 4, 2, FLD_HID);
    w8_stage_layer<1>(w8_w + W8_W_D0, d0, FLD_HID, FLD_HID, 4, 2, FLD_HID);
    w8_stage_layer<1>(w8_w + W8_W_DO, dO, 16, FLD_HID, 1, 2, FLD_HID);
    w8_stage_layer<2>(w8_w + W8_W_R0, r0, FLD_HID, in_r0, 4, 3, in_r0);
    w8_stage_layer<1>(w8_w + W8_W_RO, rO, 16, FLD_HID, 1, 2, FLD_HID);
    // the image rings start out as zeros: stages that run ahead of / behind the stream's range read rings nobody has written yet, and although
    // their gradient-side operands are forced to zero, 0 x (uninitialised NaN pattern) would still poison an accumulator
    for (uint32_t i = threadIdx.x; i < 2 * W8_PIPE_BYTES / 16; i += W8_THREADS) reinterpret_cast<uint4 *>(w8_lds)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // waves w and w + 4 share a SIMD (dispatch order 0 -> 2 -> 1 -> 3): the short forward stages sit beside the long backward ones
    const uint32_t pipe = (wave >> 1) & 1;
    const uint32_t role = wave < 4 ? (wave & 1) : 3 - (wave & 1);                  // 0 Af, 1 Bf, 2 Bb, 3 Ab: (Af, Ab) and (Bf, Bb) share a SIMD
    unsigned char *px = w8_lds + pipe * W8_PIPE_BYTES;
    const unsigned char *wb = reinterpret_cast<const unsigned char *>(w8_w);
    const uint32_t n_tiles = (P_ + W8_TILE - 1) / W8_TILE;
    const uint32_t G = gridDim.x * 2, gp = blockIdx.x * 2 + pipe;
    const uint32_t n_iter = (n_tiles + G - 1) / G;                                 // workgroup-uniform
    const uint32_t n_phase = n_iter + 3;                                           // tile i leaves the pipeline in phase i + 3
    float *part = partials + (size_t)gp * po.total;
    const uint32_t lane16 = (lane * 16) ^ ((g >> 1) << 7), slot = w8_slot(lane);       // forward fragment slot of this lane (swizzled: w8_stage_layer)
    const uint32_t i16 = lane & 15, tq = i16 & 3, trow = (4 * g + (i16 >> 2)) * 16;
    const uint32_t lane_off_T = tq * 256 + (trow ^ ((tq >> 1) << 7));                                          // C-order input columns: g_f = q
    const uint32_t lane_off_Tn0 = (tq >> 1) * 256 + trow + 8 * (tq & 1);                                       // natural columns (n0), mp = 0: g_f = q >> 1
    const uint32_t lane_off_Tn1 = (tq >> 1) * 256 + (trow ^ 128u) + 8 * (tq & 1);                              // mp = 1: g_f = 2 + (q >> 1)
    const W8Op lop = w8_lane_off_op(lane);
    [[maybe_unused]] unsigned long long tw_ = 0, tb_ = 0;
    // Tiles outside [0, n_iter) are computed on clamped inputs and ignored by every consumer: the chains are straight-line code, only the
    // accumulations and the global stores test the tile index.

    if (role == 0) {
        // ======================================================================== Af: geometry forward of tile p
        auto enc_request = [&](uint32_t tile, uint32_t (&raw)[4]) __attribute__((always_inline)) {
            const uint32_t *e = reinterpret_cast<const uint32_t *>(enc) + min(tile * W8_TILE + n, P_ - 1);
#pragma unroll
            for (int jj = 0; jj < 4; jj++) raw[jj] = e[(size_t)min(4 * g + jj, dm.L - 1) * P_];
        };
        cn_f16v wn2[2][2], wro[2], wdo[2];
#pragma unroll
        for (int a = 0; a < 2; a++) {
            w8_zero16(wro[a]); w8_zero16(wdo[a]);
#pragma unroll
            for (int b = 0; b < 2; b++) w8_zero16(wn2[a][b]);
        }
        uint32_t N0[4], N1[4];
        enc_request(gp, N0);
#pragma unroll
        for (int jj = 0; jj < 4; jj++) N1[jj] = 0;
        const unsigned char *w_n0 = wb + 2 * W8_W_N0, *w_n1 = wb + 2 * W8_W_N1, *w_n2 = wb + 2 * W8_W_N2;
        auto phase = [&](uint32_t p, uint32_t (&xcur)[4], uint32_t (&xnext)[4]) __attribute__((always_inline)) {
            W8_T0();
            asm volatile("" ::: "memory");
            enc_request(gp + (p + 1) * G, xnext);
            unsigned char *my = px + W8_AF + (p & 3) * W8_AF_BYTES;
            const bool nv = p < n_iter && (gp + p * G) * W8_TILE + n < P_;
            cn_h8 wa[4], wb1[8], wb2[8];
            w8_load_F<4, 1, 1>(w_n0, lane16, wa);
            if (NGEO == 2) w8_load_F<4, 2, 2>(w_n1, lane16, wb1);
            {   // weight gradients out of earlier phases' rings (independent of this phase's chain): dW_n2 of tile p - 3, dW_rO / dW_dO of tile p - 2
                const bool on3 = p >= 3 && p - 3 < n_iter, on2 = p >= 2 && p - 2 < n_iter;
                const unsigned char *z3i = px + W8_Z3 + ((p + 1) & 1) * 2 * W8_K;                       // (p - 3) & 1
                const unsigned char *hl = px + W8_AF + ((p + 1) & 3) * W8_AF_BYTES + ((NGEO == 2) ? W8_AF_H2 : W8_AF_H1);
                const unsigned char *bf = px + W8_BF + (p & 1) * W8_BF_BYTES;                          // (p - 2) & 1
                const cn_h8 z0 = w8_op_if(on3, z3i, lop, 0), z1 = w8_op_if(on3, z3i, lop, 1), a0 = w8_op(hl, lop, 0), a1 = w8_op(hl, lop, 1);
                const cn_h8 zro = w8_op_if(on2, bf + W8_BF_BRO, lop, 0), zdo = w8_op_if(on2, bf + W8_BF_BDO, lop, 0);
                const cn_h8 ahr0 = w8_op(bf + W8_BF_HR, lop, 0), ahr1 = w8_op(bf + W8_BF_HR, lop, 1);
                const cn_h8 ahd0 = w8_op(bf + W8_BF_HD, lop, 0), ahd1 = w8_op(bf + W8_BF_HD, lop, 1);
                wn2[0][0] = w8_mfma32(z0, a0, wn2[0][0]); wn2[0][1] = w8_mfma32(z0, a1, wn2[0][1]);
                wn2[1][0] = w8_mfma32(z1, a0, wn2[1][0]); wn2[1][1] = w8_mfma32(z1, a1, wn2[1][1]);
                wro[0] = w8_mfma32(zro, ahr0, wro[0]); wro[1] = w8_mfma32(zro, ahr1, wro[1]);
                wdo[0] = w8_mfma32(zdo, ahd0, wdo[0]); wdo[1] = w8_mfma32(zdo, ahd1, wdo[1]);
            }
            cn_h8 x0[1], h1[2], h2[2];
            {
                union { cn_h8 h; uint32_t u[4]; } f;
#pragma unroll
                for (int jj = 0; jj < 4; jj++) f.u[jj] = (nv && (4 * g + jj) < dm.L) ? xcur[jj] : 0u;
                x0[0] = f.h;
            }
            w8_publish<1>(my + W8_AF_X0, slot, x0);
            w8_f4 acc[4];
            w8_zero(acc);
            __builtin_amdgcn_sched_barrier(0);
            w8_mm<4, 1>(wa, x0, acc);
            w8_load_F<4, 2, 2>(w_n2, lane16, wb2);
            __builtin_amdgcn_sched_barrier(0);
            w8_c_to_b<1>(acc, nullptr, h1);
            w8_publish<2>(my + W8_AF_H1, slot, h1);
            if (NGEO == 2) {
                w8_zero(acc);
                __builtin_amdgcn_sched_barrier(0);
                w8_mm<4, 2>(wb1, h1, acc);
                __builtin_amdgcn_sched_barrier(0);
                w8_c_to_b<1>(acc, nullptr, h2);
                w8_publish<2>(my + W8_AF_H2, slot, h2);
            }
            w8_zero(acc);
            __builtin_amdgcn_sched_barrier(0);
            w8_mm<4, 2>(wb2, (NGEO == 2) ? h2 : h1, acc);
            __builtin_amdgcn_sched_barrier(0);
            cn_h8 fea[2];
            w8_c_to_b<0>(acc, nullptr, fea);
            w8_publish<2>(px + W8_FEA + (p & 3) * 2 * W8_K, slot, fea);
            W8_T1();
        };
        uint32_t p = 0;
        for (; p + 1 < n_phase; p += 2) {
            phase(p, N0, N1);
            phase(p + 1, N1, N0);
        }
        if (p < n_phase) phase(p, N0, N1);
#pragma unroll
        for (int a = 0; a < 2; a++) {
            w8_store<false, true>(part, po.rO, 64, 0, dm.n_rgb_out, 64, 0, a, lane, wro[a]);
            w8_store<false, true>(part, po.dO, 64, 0, 1, 64, 0, a, lane, wdo[a]);
#pragma unroll
            for (int b = 0; b < 2; b++) w8_store<true, true>(part, po.n2, 64, 0, 64, 64, a, b, lane, wn2[a][b]);
        }
        // the padding rows of the two output layers (the partial row is summed entry by entry: it is written in full, never zero-filled)
        for (uint32_t i = dm.n_rgb_out * 64 + lane; i < 16 * 64; i += 64) part[po.rO + i] = 0.0f;
        for (uint32_t i = 64 + lane; i < 16 * 64; i += 64) part[po.dO + i] = 0.0f;
    } else if (role == 1) {
        // ======================================================================== Bf: heads forward of tile p - 1, output-layer gradients; dW_d0 of tile p - 3
        cn_f16v wd0[2][2];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) w8_zero16(wd0[a][b]);
        struct BIn { float x, y, z, gs, dx, dy, dz; float4 gc; };
        auto load_in = [&](uint32_t tile) __attribute__((always_inline)) {
            BIn r;
            const uint32_t p = min(tile * W8_TILE + n, P_ - 1);
            r.x = xyz[(size_t)p * 3]; r.y = xyz[(size_t)p * 3 + 1]; r.z = xyz[(size_t)p * 3 + 2];
            r.gs = g_sigma[p];
            r.gc = *reinterpret_cast<const float4 *>(g_rgbc + (size_t)p * 4);
            const float *dp = dirs + (size_t)(p / dir_group) * 3;
            r.dx = dp[0]; r.dy = dp[1]; r.dz = dp[2];
            return r;
        };
        BIn I0 = load_in(gp), I1 = I0;
        const bool dir_uniform = (dir_group % W8_TILE) == 0;
        unsigned char *scr = px + W8_SCR;
        const unsigned char *w_d0 = wb + 2 * W8_W_D0, *w_dO = wb + 2 * W8_W_DO, *w_r0 = wb + 2 * W8_W_R0, *w_rO = wb + 2 * W8_W_RO;
        auto phase = [&](uint32_t p, const BIn &cur, BIn &nxt) __attribute__((always_inline)) {
            W8_T0();
            asm volatile("" ::: "memory");
            if (p >= 1) nxt = load_in(gp + p * G);
            const uint32_t tile = gp + (p - 1) * G;                                 // (wraps at p == 0: outside the range, ignored)
            const bool valid = p >= 1 && p - 1 < n_iter && tile * W8_TILE + n < P_;
            unsigned char *my = px + W8_BF + ((p + 1) & 1) * W8_BF_BYTES;
            // ---- stage 0: fea, direction features, fragments of d0 / r0
            cn_h8 x3[3];                                                            // [fea K-step 0, fea K-step 1, direction features]
            w8_fetch<2>(px + W8_FEA + ((p + 3) & 3) * 2 * W8_K, slot, x3);          // (p - 1) & 3
            cn_h8 wd[8], wr[12];
            w8_load_F<4, 2, 2>(w_d0, lane16, wd);
            w8_load_F<4, 3, 3>(w_r0, lane16, wr);
            {   // dW_d0 = dz_d x fea of tile p - 3, both out of rings written in earlier phases
                const bool on3 = p >= 3 && p - 3 < n_iter;
                const unsigned char *zdi = px + W8_ZD + ((p + 1) & 1) * 2 * W8_K, *fe3 = px + W8_FEA + ((p + 1) & 3) * 2 * W8_K;
                const cn_h8 z0 = w8_op_if(on3, zdi, lop, 0), z1 = w8_op_if(on3, zdi, lop, 1), f0 = w8_op(fe3, lop, 0), f1 = w8_op(fe3, lop, 1);
                wd0[0][0] = w8_mfma32(z0, f0, wd0[0][0]); wd0[0][1] = w8_mfma32(z0, f1, wd0[0][1]);
                wd0[1][0] = w8_mfma32(z1, f0, wd0[1][0]); wd0[1][1] = w8_mfma32(z1, f1, wd0[1][1]);
            }
            {
                // direction features (frequency encoding, nerf/base.py:42-60) in natural order.  One direction per tile (run() path): lanes
                // q < 27 evaluate feature q and the 32 halves go through a 64-byte scratch of this wave; otherwise every lane evaluates the
                // eight features 8 g .. 8 g + 7 of its own sample.  The component is picked arithmetically (a select chain over the
                // adjacent struct fields becomes an indexed load and drags the whole struct into scratch memory).
                const float ddx = cur.dx, ddy = cur.dy, ddz = cur.dz;
                auto feat = [&](uint32_t q) __attribute__((always_inline)) {
                    const uint32_t qq = q >= 3 ? q - 3 : 0, k = qq / 6, r = qq % 6, c = q < 3 ? q : (r < 3 ? r : r - 3);
                    const float d = ddx * (c == 0 ? 1.0f : 0.0f) + ddy * (c == 1 ? 1.0f : 0.0f) + ddz * (c == 2 ? 1.0f : 0.0f);
                    const float a = d * (float)(1u << k);
                    const float sn = __sinf(a), cs = __cosf(a);
                    const float sc = r < 3 ? sn : cs;
                    return q < 3 ? d : (q < FLD_NDIR ? sc : 0.0f);
                };
                if (dir_uniform) {
                    const float v = feat(lane & 31);
                    if (lane < 32) reinterpret_cast<_Float16 *>(scr)[lane] = (_Float16)v;
                    x3[2] = *reinterpret_cast<const cn_h8 *>(scr + 16 * g);
                } else {
                    cn_h8 f;
#pragma unroll
                    for (int j = 0; j < 8; j++) f[j] = (_Float16)feat(8 * g + j);
                    x3[2] = f;
                }
            }
            w8_publish<1>(my + W8_BF_DIR, slot, x3 + 2);
            cn_h8 wdo_f[2], wro_f[2];
            w8_load_F<1, 2, 2>(w_dO, lane16, wdo_f);
            w8_load_F<1, 2, 2>(w_rO, lane16, wro_f);
            __builtin_amdgcn_sched_barrier(0);
            // ---- stage 1: d0 fea | r0 [fea, dir]
            w8_f4 accd[4], accr[4];
            w8_zero(accd); w8_zero(accr);
            w8_mm<4, 2>(wd, x3, accd);
            w8_mm<4, 3>(wr, x3, accr);
            __builtin_amdgcn_sched_barrier(0);
            // ---- stage 2: output layers
            cn_h8 hd[2], hr[2];
            w8_c_to_b<1>(accd, nullptr, hd);
            w8_c_to_b<1>(accr, nullptr, hr);
            w8_publish<2>(my + W8_BF_HD, slot, hd);
            w8_publish<2>(my + W8_BF_HR, slot, hr);
            w8_f4 outd[1], outr[1];
            w8_zero(outd); w8_zero(outr);
            __builtin_amdgcn_sched_barrier(0);
            w8_mm<1, 2>(wdo_f, hd, outd);
            w8_mm<1, 2>(wro_f, hr, outr);
            __builtin_amdgcn_sched_barrier(0);
            // ---- stage 3: output-layer gradients (sigmoid', clamped exp': provider_utils.py:26-29): rows 0..3 live in the g == 0 lanes
            cn_h8 bro = Prec<true>::zero(), bdo = Prec<true>::zero();
            {
                const float raw = (float)(_Float16)outd[0][0];
                const float x = cur.x, y = cur.y, z = cur.z;
                const float gg = 5.0f * __expf(-(x * x + y * y + z * z) * (1.0f / 0.08f));
                const bool on = valid && g == 0;
                bdo[0] = (_Float16)(on ? cur.gs * __expf(fminf(fmaxf(raw + gg, -15.0f), 15.0f)) : 0.0f);
                const float gcv[4] = {cur.gc.x, cur.gc.y, cur.gc.z, cur.gc.w};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const float sg = (float)(_Float16)__builtin_amdgcn_rcpf(1.0f + __expf(-outr[0][k]));
                    bro[k] = (_Float16)((on && k < (int)dm.n_rgb_out) ? gcv[k] * sg * (1.0f - sg) : 0.0f);
                }
            }
            w8_publish<1>(my + W8_BF_BRO, slot, &bro);
            w8_publish<1>(my + W8_BF_BDO, slot, &bdo);
            W8_T1();
        };
        uint32_t p = 0;
        for (; p + 1 < n_phase; p += 2) {
            phase(p, I1, I0);
            phase(p + 1, I0, I1);
        }
        if (p < n_phase) phase(p, I1, I0);
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) w8_store<true, true>(part, po.d0, 64, 0, 64, 64, a, b, lane, wd0[a][b]);
    } else if (role == 2) {
        // ======================================================================== Bb: heads backward of tile p - 2, dW_r0 (same tile)
        cn_f16v wrd[2], wrf[2][2];
#pragma unroll
        for (int a = 0; a < 2; a++) {
            w8_zero16(wrd[a]);
#pragma unroll
            for (int b = 0; b < 2; b++) w8_zero16(wrf[a][b]);
        }
        const unsigned char *w_d0 = wb + 2 * W8_W_D0, *w_dO = wb + 2 * W8_W_DO, *w_r0 = wb + 2 * W8_W_R0, *w_rO = wb + 2 * W8_W_RO;
        unsigned char *mine = px + W8_BB;
        for (uint32_t p = 0; p < n_phase; p++) {
            W8_T0();
            asm volatile("" ::: "memory");
            const bool on_tile = p >= 2 && p - 2 < n_iter;
            const unsigned char *bf = px + W8_BF + (p & 1) * W8_BF_BYTES;          // (p - 2) & 1
            const unsigned char *fe = px + W8_FEA + ((p + 2) & 3) * 2 * W8_K;      // (p - 2) & 3
            unsigned char *zdo_ = px + W8_ZD + (p & 1) * 2 * W8_K;                 // (p - 2) & 1
            // ---- stage 0: output-layer gradients, the forward's activations (masks), every transposed fragment of the phase
            cn_h8 hd[2], hr[2], bro[1], bdo[1];
            w8_fetch<1>(bf + W8_BF_BDO, slot, bdo);
            w8_fetch<1>(bf + W8_BF_BRO, slot, bro);
            cn_h8 wdoT[4], wroT[4];
            w8_load_T16<4, 2>(w_dO, lane_off_T, wdoT);
            w8_load_T16<4, 2>(w_rO, lane_off_T, wroT);
            w8_fetch<2>(bf + W8_BF_HD, slot, hd);
            w8_fetch<2>(bf + W8_BF_HR, slot, hr);
            cn_h8 wdT[8], wrT[8];
            w8_load_T<4, 2, 2>(w_d0, lane_off_T, wdT);
            w8_load_T<4, 2, 3>(w_r0, lane_off_T, wrT);
            // the activation operands of dW_r0 = dz_r x [dir | fea]: rings of earlier phases
            const cn_h8 f0 = w8_op(fe, lop, 0), f1 = w8_op(fe, lop, 1), dd = w8_op(bf + W8_BF_DIR, lop, 0);
            w8_f4 accd[4], accr[4];
            w8_zero(accd); w8_zero(accr);
            __builtin_amdgcn_sched_barrier(0);
            // ---- stage 1: dO^T | rO^T
            w8_mm<4, 1>(wdoT, bdo, accd);
            w8_mm<4, 1>(wroT, bro, accr);
            __builtin_amdgcn_sched_barrier(0);
            // ---- stage 2: d(fea) = d0^T dz_d + r0^T dz_r
            cn_h8 zd[2], zr[2];
            w8_c_to_b<2>(accr, hr, zr);
            w8_publish<2>(mine + W8_BB_ZR, slot, zr);
            w8_c_to_b<2>(accd, hd, zd);
            w8_publish<2>(zdo_, slot, zd);
            w8_f4 dfea[4];
            w8_zero(dfea);
            // dz_r read back transposed for the weight gradients (in-order DS: after this wave's own publish)
            const cn_h8 z0 = w8_op_if(on_tile, mine + W8_BB_ZR, lop, 0), z1 = w8_op_if(on_tile, mine + W8_BB_ZR, lop, 1);
            __builtin_amdgcn_sched_barrier(0);
            w8_mm<4, 2>(wrT, zr, dfea);
            w8_mm<4, 2>(wdT, zd, dfea);
            wrd[0] = w8_mfma32(z0, dd, wrd[0]); wrd[1] = w8_mfma32(z1, dd, wrd[1]);
            wrf[0][0] = w8_mfma32(z0, f0, wrf[0][0]); wrf[0][1] = w8_mfma32(z0, f1, wrf[0][1]);
            wrf[1][0] = w8_mfma32(z1, f0, wrf[1][0]); wrf[1][1] = w8_mfma32(z1, f1, wrf[1][1]);
            __builtin_amdgcn_sched_barrier(0);
            cn_h8 z3[2];
            w8_c_to_b<0>(dfea, nullptr, z3);
            w8_publish<2>(px + W8_Z3 + (p & 1) * 2 * W8_K, slot, z3);               // (p - 2) & 1
            W8_T1();
        }
        for (uint32_t i = lane; i < 64 * (96 - FLD_NDIR - 64); i += 64) {
            const uint32_t row = i / (96 - FLD_NDIR - 64), col = FLD_NDIR + 64 + i % (96 - FLD_NDIR - 64);
            part[po.r0 + row * 96 + col] = 0.0f;
        }
#pragma unroll
        for (int a = 0; a < 2; a++) {
            w8_store<true, false>(part, po.r0, 96, 0, 64, FLD_NDIR, a, 0, lane, wrd[a]);
#pragma unroll
            for (int b = 0; b < 2; b++) w8_store<true, true>(part, po.r0, 96, FLD_NDIR, 64, 64, a, b, lane, wrf[a][b]);
        }
    } else {
        // ======================================================================== Ab: geometry backward of tile p - 3, dW_n1, dW_n0 (same tile)
        cn_f16v wn1[2][2], wn0[2];
#pragma unroll
        for (int a = 0; a < 2; a++) {
            w8_zero16(wn0[a]);
#pragma unroll
            for (int b = 0; b < 2; b++) w8_zero16(wn1[a][b]);
        }
        const unsigned char *w_n0 = wb + 2 * W8_W_N0, *w_n1 = wb + 2 * W8_W_N1, *w_n2 = wb + 2 * W8_W_N2;
        unsigned char *mine = px + W8_AB;
        for (uint32_t p = 0; p < n_phase; p++) {
            W8_T0();
            asm volatile("" ::: "memory");
            const bool on_tile = p >= 3 && p - 3 < n_iter;
            const uint32_t np = (gp + (p - 3) * G) * W8_TILE + n;
            const bool valid = on_tile && np < P_;
            const unsigned char *af = px + W8_AF + ((p + 1) & 3) * W8_AF_BYTES;    // (p - 3) & 3
            const unsigned char *z3i = px + W8_Z3 + ((p + 1) & 1) * 2 * W8_K;      // (p - 3) & 1
            // ---- stage 0: dz_3, the forward's activations (masks + weight-gradient operands), every transposed fragment of the phase
            cn_h8 z3[2], h1[2], h2[2];
            w8_fetch<2>(z3i, slot, z3);
            cn_h8 wT2[8], wT1[8], wTn[4];
            w8_load_T<4, 2, 2>(w_n2, lane_off_T, wT2);
            if (NGEO == 2) {
                w8_fetch<2>(af + W8_AF_H2, slot, h2);
                w8_load_T<4, 2, 2>(w_n1, lane_off_T, wT1);
            }
            w8_fetch<2>(af + W8_AF_H1, slot, h1);
            w8_load_Tn<2, 2>(w_n0, lane_off_Tn0, lane_off_Tn1, wTn);
            const cn_h8 a10 = w8_op(af + W8_AF_H1, lop, 0), a11 = w8_op(af + W8_AF_H1, lop, 1), ax0 = w8_op(af + W8_AF_X0, lop, 0);
            w8_f4 acc[4];
            w8_zero(acc);
            __builtin_amdgcn_sched_barrier(0);
            // ---- stage 1: n2^T dz3
            w8_mm<4, 2>(wT2, z3, acc);
            __builtin_amdgcn_sched_barrier(0);
            cn_h8 z1f[2];
            if (NGEO == 2) {
                // ---- stage 2: n1^T dz2 (+ dW_n1 = dz2 x h1)
                cn_h8 z2[2];
                w8_c_to_b<2>(acc, h2, z2);
                w8_publish<2>(mine + W8_AB_Z2, slot, z2);
                w8_zero(acc);
                const cn_h8 z0 = w8_op_if(on_tile, mine + W8_AB_Z2, lop, 0), z1 = w8_op_if(on_tile, mine + W8_AB_Z2, lop, 1);
                __builtin_amdgcn_sched_barrier(0);
                w8_mm<4, 2>(wT1, z2, acc);
                wn1[0][0] = w8_mfma32(z0, a10, wn1[0][0]); wn1[0][1] = w8_mfma32(z0, a11, wn1[0][1]);
                wn1[1][0] = w8_mfma32(z1, a10, wn1[1][0]); wn1[1][1] = w8_mfma32(z1, a11, wn1[1][1]);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- stage 3: n0^T dz1 -> d(loss)/d(grid features) (+ dW_n0 = dz1 x x0)
            w8_c_to_b<2>(acc, h1, z1f);
            w8_publish<2>(mine + W8_AB_Z1, slot, z1f);
            w8_f4 denc[2];
            w8_zero(denc);
            const cn_h8 y0 = w8_op_if(on_tile, mine + W8_AB_Z1, lop, 0), y1 = w8_op_if(on_tile, mine + W8_AB_Z1, lop, 1);
            __builtin_amdgcn_sched_barrier(0);
            w8_mm<2, 2>(wTn, z1f, denc);
            wn0[0] = w8_mfma32(y0, ax0, wn0[0]); wn0[1] = w8_mfma32(y1, ax0, wn0[1]);
            __builtin_amdgcn_sched_barrier(0);
            if (valid) {
#pragma unroll
                for (int mp = 0; mp < 2; mp++) {
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const uint32_t level = 8 * mp + 2 * g + h;
                        if (level < dm.L) reinterpret_cast<uint32_t *>(grad_enc)[(size_t)level * P_ + np] = w8_cvt_pk(denc[mp][2 * h], denc[mp][2 * h + 1]);
                    }
                }
            }
            W8_T1();
        }
#pragma unroll
        for (int a = 0; a < 2; a++) {
            w8_store<true, false>(part, po.n0, dm.enc_pad, 0, 64, dm.enc_pad, a, 0, lane, wn0[a]);
            if (NGEO == 2) {
#pragma unroll
                for (int b = 0; b < 2; b++) w8_store<true, true>(part, po.n1, 64, 0, 64, 64, a, b, lane, wn1[a][b]);
            }
        }
    }
#ifdef CNERF_TUNING
    if ((ablate & 32) && lane == 0) {                         // role-timing dump: rows 510 / 511 of the workspace (blocks <= 255 then)
        unsigned long long *tt = reinterpret_cast<unsigned long long *>(partials + (size_t)510 * po.total) + ((size_t)blockIdx.x * 8 + wave) * 2;
        tt[0] = tw_; tt[1] = tb_;
    }
#endif
}

// ------------------------------------------------------------------------------------------------ host entry (called from field_bwd_fused.hip)
void ff_reduce_partials(const float *partials, uint32_t n_partials, uint32_t total, uint32_t n_net, uint32_t n_den, float *g_net, float *g_den, float *g_rgb,
                        hipStream_t st);

bool w8_eligible(const FieldDims &dm) {
    static const int on = cn_tune_env("CNERF_FIELD_W8_BWD", 0);   // tuning builds only: 1 selects this kernel instead of k_field_bwd_x2
    return on && dm.enc_pad == 32 && (dm.n_hidden_geo == 1 || dm.n_hidden_geo == 2);
}

int w8_launch(const void *enc, const float *xyz, const float *dirs, uint32_t dir_group, uint32_t P_, const FieldDims &dm, const float *pnet, const float *pden,
              const float *prgb, const float *g_sigma, const float *g_rgbc, void *grad_enc, float *g_net, float *g_den, float *g_rgb, void *workspace,
              uint32_t max_partials, hipStream_t st) {
    const uint32_t lds_bytes = 2 * W8_PIPE_BYTES;                               // dynamic part; the weight fragments are a 44 KiB static array
    const uint32_t n_tiles = cn_div_up(P_, W8_TILE);
    uint32_t blocks = cn_div_up(n_tiles, 2);
    if (blocks > max_partials / 2) blocks = max_partials / 2;                   // one partial-gradient row per pipeline, two pipelines per workgroup
    if (blocks > 256) blocks = 256;
    const W8Off po = w8_offsets(dm);
    float *partials = reinterpret_cast<float *>(workspace);
    static const int ablate = cn_tune_env("CNERF_W8_ABLATE", 0);               // tuning builds only; the release kernel folds the mask to 0
    if ((ablate & 32) && blocks > 255) blocks = 255;                           // rows 510 / 511 of the workspace hold the timing dump
#define W8_LAUNCH(KERN)                                                                                                                    \
    {                                                                                                                                      \
        auto kern = KERN;                                                                                                                  \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);            \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(W8_THREADS), lds_bytes, st, enc, xyz, dirs, dir_group, P_, dm, pnet, pden, prgb, g_sigma, g_rgbc, \
                           grad_enc, partials, (uint32_t)ablate);                                                                          \
    }
    if (dm.n_hidden_geo == 2) W8_LAUNCH(k_field_bwd_w8<2>) else W8_LAUNCH(k_field_bwd_w8<1>)
    int rc = cn_launch_status();
    if (rc) return rc;
    ff_reduce_partials(partials, blocks * 2, po.total, po.d0, po.r0 - po.d0, g_net, g_den, g_rgb, st);
    return cn_launch_status();
}
#else   // release build: the kernel is not compiled in
bool w8_eligible(const FieldDims &) { return false; }
int w8_launch(const void *, const float *, const float *, uint32_t, uint32_t, const FieldDims &, const float *, const float *, const float *, const float *,
              const float *, void *, float *, float *, float *, void *, uint32_t, hipStream_t) { return CNERF_EINVAL; }
#endif
