// Fused field backward for gfx950 (fp16, enc_pad == 32) — the default backward of the fused field: k_field_bwd_x2.
//
// History of this kernel (measurements: profiles/r02_*):
//   * round 1, field_bwd_mma.hip: two producer/consumer wave pairs per workgroup, persistent weight-gradient tiles, operands transposed by
//     identity MFMAs.  508 registers per lane; with a budget above 256 the compiler selects the accumulation-register form of EVERY MFMA,
//     so each value of every layer output costs a v_accvgpr_read before the vector ALU may touch it; with the transposes' repacking the
//     kernel issued ~13 vector instructions per MFMA: VALU 48 % of its cycles, matrix pipe 28 %.  632 us per 2.1 M samples.
//   * round 2, first attempt (k_field_bwd_x4, removed again): four roles per workgroup (net chain, heads chain, two weight-gradient waves),
//     every wave below 256 registers (VGPR-destination MFMAs), operands handed over as LDS images.  Correct, but ONE tile in flight per
//     CU: the double-buffered images (104 KiB) beside the weights leave room for a single workgroup, and a chain wave alone on its SIMD
//     is latency-bound (4.5 k cycles per tile against 1.5 k cycles of MFMA issue).  700 us.
//   * this kernel: back to two pairs per workgroup, but the weight-gradient operands come from WAVE-LOCAL LDS images read with the
//     transposing read of CDNA4 (ds_read_b64_tr_b16) instead of identity MFMAs, the transposed weight copy is gone (W^T fragments are
//     transposing reads of the forward fragment store), ReLU / masks are packed 16-bit integer operations, every prefetch is
//     unconditional with statically named buffers (no exposed global latency), weights and images are separate LDS objects.  517 us.
#include "field_bwd_common.h"

typedef short x4_s4 __attribute__((__vector_size__(4 * sizeof(short))));
typedef short x4_s2 __attribute__((ext_vector_type(2)));

struct MmOff {
    uint32_t n0, n1, n2, d0, dO, r0, rO, total;
};
__host__ __device__ __forceinline__ MmOff x4_offsets(const FieldDims &dm) {
    MmOff o;
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

#define X4_K 1024                            // one K-step image: 64 lanes x 16 B (16 features x 32 samples)

// A fragment of W^T for the data-gradient chain, read out of the FORWARD fragment store of the layer (dst[((t S + s) 64 + lane) 8 + j] =
// W[32 t + (lane & 31)][col(s, lane >> 5, j)], field.hip) by two transposing reads.  Wanted: lane l = input feature 32 t' + (l & 31)
// (natural order inside the tile), K-slot (s', hi' = l >> 5, j') = output feature clayout(s', hi', j') — the order in which dz arrives.
// Those are the output rows li = 16 (s' & 1) + 8 c + 4 hi' + (j' & 3) (c = j' >> 2) of output tile t = s' >> 1: two runs of four
// consecutive rows, one transposing read each; a read's 16-lane group G = (l >> 4) covers input features 16 (G & 1) .. + 15 in four 8-byte
// chunks (address role of lane i = l & 15: row i >> 2, chunk q = i & 3), which for C-ordered columns sit at K-step 2 t' + (G & 1),
// half q & 1, element offset 4 (q >> 1) and for natural columns (first layer) at half q >> 1, offset 4 (q & 1).
// `lane_off` is that lane-dependent part (x4_lane_off_w), everything else is a compile-time immediate.
// The store is bank-swizzled like the images below (fb_stage_layer<.., SWZ = true>, x4_img_swz): one lane offset per read.
struct X4Img { uint32_t c0, c1; };
__device__ __forceinline__ uint32_t x4_img_swz(uint32_t byte_off) { return byte_off ^ (((byte_off >> 9) & 3u) << 6); }
template <int S>
__device__ __forceinline__ cn_h8 x4_frag_T(const unsigned char *layer, X4Img lane_off, int tp, int sp) {
    const uint32_t c = ((sp >> 1) * S + 2 * tp) * 1024 + (16 * (sp & 1)) * 16;
    union { cn_h8 h; x4_s4 s[2]; } f;
    f.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((x4_s4 __attribute__((address_space(3))) *)(layer + lane_off.c0 + c));
    f.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((x4_s4 __attribute__((address_space(3))) *)(layer + lane_off.c1 + c));
    return f.h;
}
__device__ __forceinline__ X4Img x4_lane_off_w(uint32_t l, bool natural_cols) {
    const uint32_t G = l >> 4, i = l & 15, rho = i >> 2, q = i & 3;
    const uint32_t hi = natural_cols ? (q >> 1) : (q & 1), j0 = natural_cols ? 4 * (q & 1) : 4 * (q >> 1);
    const uint32_t base = (G & 1) * 1024 + hi * 512 + (4 * (l >> 5) + rho) * 16;
    X4Img o;
    o.c0 = x4_img_swz(base) + j0 * 2;
    o.c1 = x4_img_swz(base + 8 * 16) + j0 * 2;
    return o;
}
// forward-order A fragment out of the swizzled store
__device__ __forceinline__ cn_h8 x4_load_frag(const _Float16 *base, uint32_t t, uint32_t S, uint32_t s, uint32_t lane) {
    return *reinterpret_cast<const cn_h8 *>(reinterpret_cast<const unsigned char *>(base) + (t * S + s) * 1024 + ((lane * 16) ^ ((((s & 1) * 2 + (lane >> 5)) << 6))));
}
// Operand fragment of a weight-gradient product out of a published image ([K-step][lane][8 halves], the B-fragment order of the chain):
// lane l = feature 32 u + (l & 31) of the image (natural numbering), K-slots = samples 16 kk + 8 (l >> 5) + j'.
// Bank swizzle of the images: the 32 lanes one transposing read serves together differ in K-step parity (1024 B) and half (512 B), both
// multiples of the 256-byte bank period — four lanes per bank.  The 16-byte slot index therefore carries (K-step parity, half) XORed into
// its bits 2..3 (x4_img_slot); the sample bit 2 of the second read then no longer adds as an immediate, so a lane holds one offset per read.
__device__ __forceinline__ X4Img x4_lane_off_img(uint32_t l, bool natural_slots) {
    const uint32_t G = l >> 4, i = l & 15, rho = i >> 2, q = i & 3;
    const uint32_t hi = natural_slots ? (q >> 1) : (q & 1), j0 = natural_slots ? 4 * (q & 1) : 4 * (q >> 1);
    const uint32_t base = (G & 1) * 1024 + hi * 512 + (8 * (l >> 5) + rho) * 16;
    X4Img o;
    o.c0 = x4_img_swz(base) + j0 * 2;
    o.c1 = x4_img_swz(base + 4 * 16) + j0 * 2;
    return o;
}
__device__ __forceinline__ cn_h8 x4_frag_img(const unsigned char *img, X4Img lane_off, int u, int kk) {
    const uint32_t c = 2 * u * 1024 + 16 * kk * 16;
    union { cn_h8 h; x4_s4 s[2]; } f;
    f.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((x4_s4 __attribute__((address_space(3))) *)(img + lane_off.c0 + c));
    f.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((x4_s4 __attribute__((address_space(3))) *)(img + lane_off.c1 + c));
    return f.h;
}

// da[t] += W^T(t, s) dz[s] with the W^T fragments read out of the forward store
// forward-order product without the scheduling fence of fb_gemm: below 256 registers there is room to let the compiler start a layer's
// fragment loads during the previous layer's epilogue (one wave per SIMD: nothing else hides the LDS latency)
template <int T, int NS>
__device__ __forceinline__ void x4_gemm(const _Float16 *wf, uint32_t S, uint32_t s0, const cn_h8 *b, uint32_t lane, cn_f16v (&acc)[T]) {
#pragma unroll
    for (int s = 0; s < NS; s++) {
#pragma unroll
        for (int t = 0; t < T; t++) acc[t] = Prec<true>::mfma(x4_load_frag(wf, t, S, s0 + s, lane), b[s], acc[t]);
    }
}

template <int T, int NS, int S>
__device__ __forceinline__ void x4_gemm_T(const unsigned char *layer, X4Img lane_off, const cn_h8 *b, cn_f16v (&acc)[T]) {
#pragma unroll
    for (int s = 0; s < NS; s++) {
#pragma unroll
        for (int t = 0; t < T; t++) acc[t] = Prec<true>::mfma(x4_frag_T<S>(layer, lane_off, t, s), b[s], acc[t]);
    }
}

// C registers of two 32-row tiles -> B fragments of the next layer, rounded to half; ReLU as a signed 16-bit integer max on the packed
// halves (positive halves order like their bit patterns; every non-positive value, -0 included, becomes +0).  The packed 16-bit integer
// operations are spelled as instructions: left to itself the compiler unpacks them into per-half compares and selects.
__device__ __forceinline__ uint32_t x4_cvt_pk(float a, float b) {
    const cn_h2 v = {(_Float16)a, (_Float16)b};                       // v_cvt_pk_f16_f32
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ uint32_t x4_pk_relu(uint32_t v) {
    uint32_t r;
    asm("v_pk_max_i16 %0, %1, 0" : "=v"(r) : "v"(v));
    return r;
}
// 0xffff per half of `act` (post-ReLU halves: bit patterns 0 .. 0x7fff) that is > 0, else 0: the sign of (0 - bits)
__device__ __forceinline__ uint32_t x4_pk_posmask(uint32_t act) {
    // (an inline constant in a packed instruction feeds its HIGH 16 bits — zero for a small integer — to the upper lane unless op_sel_hi
    // says otherwise; the shift count therefore travels in a register with both halves set)
    uint32_t t, m;
    asm("v_pk_sub_i16 %0, 0, %1" : "=v"(t) : "v"(act));
    asm("v_pk_ashrrev_i16 %0, %1, %2" : "=v"(m) : "v"(0x000F000Fu), "v"(t));
    return m;
}
template <bool RELU>
__device__ __forceinline__ void x4_c_to_b(const cn_f16v (&acc)[2], cn_h8 *b) {
#pragma unroll
    for (int u = 0; u < 2; u++) {
#pragma unroll
        for (int sub = 0; sub < 2; sub++) {
            union { cn_h8 h; uint32_t w[4]; } f;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t v = x4_cvt_pk(acc[u][8 * sub + 2 * j], acc[u][8 * sub + 2 * j + 1]);
                f.w[j] = RELU ? x4_pk_relu(v) : v;
            }
            b[u * 2 + sub] = f.h;
        }
    }
}
// dz = da * [act > 0]: act holds the forward's post-ReLU halves
__device__ __forceinline__ void x4_c_to_b_masked(const cn_f16v (&acc)[2], const cn_h8 *act, cn_h8 *b) {
#pragma unroll
    for (int u = 0; u < 2; u++) {
#pragma unroll
        for (int sub = 0; sub < 2; sub++) {
            union { cn_h8 h; uint32_t w[4]; } f, a;
            a.h = act[u * 2 + sub];
#pragma unroll
            for (int j = 0; j < 4; j++) f.w[j] = x4_cvt_pk(acc[u][8 * sub + 2 * j], acc[u][8 * sub + 2 * j + 1]) & x4_pk_posmask(a.w[j]);
            b[u * 2 + sub] = f.h;
        }
    }
}

// Phase barrier that leaves global loads in flight: __syncthreads() drains vmcnt too, which exposed the full latency of every prefetch
// (grid features, per-sample gradients, directions) once per phase — ~1-2.5 us per 32-sample tile, the whole kernel time.  Only this
// wave's LDS operations have to be complete before the other waves may read what it published.
__device__ __forceinline__ void x4_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Grid features of one sample as B fragments (natural order), split in two so that the loads are a PREFETCH: x4_enc_request issues them
// unconditionally (sample and level indices clamped into range — no `valid ? load : 0`, which the compiler turns into one exec-masked
// branch per level with the wait right behind each load), x4_enc_mask zeroes the padding when the fragments are consumed a phase later.
template <int SENC>
__device__ __forceinline__ void x4_enc_request(const void *__restrict__ enc, uint32_t P_, uint32_t L, uint32_t p, uint32_t hi, cn_h8 (&b)[SENC]) {
    const uint32_t *e = reinterpret_cast<const uint32_t *>(enc) + min(p, P_ - 1);
#pragma unroll
    for (int s = 0; s < SENC; s++) {
        union { cn_h8 h; uint32_t u[4]; } f;
#pragma unroll
        for (int jj = 0; jj < 4; jj++) f.u[jj] = e[(size_t)min(8 * s + 4 * hi + jj, L - 1) * P_];
        b[s] = f.h;
    }
}
template <int SENC>
__device__ __forceinline__ void x4_enc_mask(const cn_h8 (&raw)[SENC], uint32_t L, bool valid, uint32_t hi, cn_h8 (&b)[SENC]) {
#pragma unroll
    for (int s = 0; s < SENC; s++) {
        union { cn_h8 h; uint32_t u[4]; } f;
        f.h = raw[s];
#pragma unroll
        for (int jj = 0; jj < 4; jj++) f.u[jj] = (valid && (8 * s + 4 * hi + jj) < L) ? f.u[jj] : 0u;
        b[s] = f.h;
    }
}

template <int NS>
__device__ __forceinline__ void x4_publish(unsigned char *img, uint32_t lane, const cn_h8 *x) {
#pragma unroll
    for (int s = 0; s < NS; s++) *reinterpret_cast<cn_h8 *>(img + s * 1024 + ((lane * 16) ^ ((((s & 1) * 2 + (lane >> 5)) << 6)))) = x[s];
}
template <int NS>
__device__ __forceinline__ void x4_fetch(const unsigned char *img, uint32_t lane, cn_h8 *x) {
#pragma unroll
    for (int s = 0; s < NS; s++) x[s] = *reinterpret_cast<const cn_h8 *>(img + s * 1024 + ((lane * 16) ^ ((((s & 1) * 2 + (lane >> 5)) << 6))));
}

// one 32 x 32 weight-gradient tile: acc += Z(out block) . A(in block)^T over the 32 samples of the tile (two K = 16 MFMAs)
__device__ __forceinline__ void x4_dw(cn_f16v &acc, const cn_h8 (&z)[2], const cn_h8 (&a)[2]) {
    acc = Prec<true>::mfma(z[0], a[0], acc);
    acc = Prec<true>::mfma(z[1], a[1], acc);
}
__device__ __forceinline__ void x4_load_block(const unsigned char *img, X4Img lane_off, int u, cn_h8 (&f)[2]) {
    f[0] = x4_frag_img(img, lane_off, u, 0);
    f[1] = x4_frag_img(img, lane_off, u, 1);
}
__device__ __forceinline__ void x4_store(float *__restrict__ part, uint32_t dst, uint32_t stride, uint32_t col0, uint32_t M, uint32_t N, uint32_t mt, uint32_t nt,
                                         uint32_t li, uint32_t hi, const cn_f16v &acc) {
    const uint32_t col = 32 * nt + li;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const uint32_t row = 32 * mt + (uint32_t)fld_rho(r, (int)hi);
        if (row < M && col < N) part[dst + (size_t)row * stride + col0 + col] = acc[r];
    }
}
__device__ __forceinline__ void x4_zero(cn_f16v &a) {
#pragma unroll
    for (int r = 0; r < 16; r++) a[r] = 0.0f;
}

// measurement aid (CNERF_X2_ABLATE bit 5): per role, cycles between barriers spent working / waiting at the barrier, summed over the phases
#define X4_T0() const unsigned long long t0_ = (ablate & 32) ? __builtin_readcyclecounter() : 0ull
#define X4_T1() do { if (ablate & 32) { const unsigned long long t1_ = __builtin_readcyclecounter(); x4_barrier(); \
                                        const unsigned long long t2_ = __builtin_readcyclecounter(); tw_ += t1_ - t0_; tb_ += t2_ - t1_; } else x4_barrier(); } while (0)

// ================================================================================================ the kernel
// Two producer/consumer pairs per workgroup — two tiles per phase.  Wave A of a pair runs the geometry MLP (forward of tile p, then
// dz_3 -> dz_1 and d(loss)/d(grid features) of tile p-2), wave B the density + colour heads of tile p-1, forward and backward; fea and
// dz_3 cross between them through double-buffered 4 KiB LDS images, one workgroup barrier per phase.  Each wave also owns the weight
// gradients of its layers: with the 512-register budget the 24 persistent accumulator tiles live in accumulation registers, which only
// MFMAs ever touch, and their operands come from wave-local images: the wave publishes its activation / gradient fragments to its own LDS
// scratch and reads them back through ds_read_b64_tr_b16 (DS operations of one wave execute in order: no barrier, no double buffering).
// The independent dW MFMAs fill issue slots that the chain's dependent load -> MFMA -> convert sequence leaves empty.
#define X2_FEA 0                              // 2 x 4 KiB, A -> B
#define X2_Z3 (2 * 4 * X4_K)                  // 2 x 4 KiB, B -> A
#define X2_A (X2_Z3 + 2 * 4 * X4_K)           // wave A's images: x0 2, h1 4, h2 4, z2 4, z1 4 KiB
#define X2_A_X0 0
#define X2_A_H1 (2 * X4_K)
#define X2_A_H2 (6 * X4_K)
#define X2_A_Z2 (10 * X4_K)
#define X2_A_Z1 (14 * X4_K)
#define X2_B (X2_A + 18 * X4_K)               // wave B's images: hd 4, hr 4, dir 2, zr 4, zd 4, bro 2, bdo 2 KiB
#define X2_B_HD 0
#define X2_B_HR (4 * X4_K)
#define X2_B_DIR (8 * X4_K)
#define X2_B_ZR (10 * X4_K)
#define X2_B_ZD (14 * X4_K)
#define X2_B_BRO (18 * X4_K)
#define X2_B_BDO (20 * X4_K)
#define X2_PAIR_BYTES (X2_B + 22 * X4_K)      // 56 KiB per pair

// SKIP: tile_live is given (one byte per 32-sample tile, 0 = every output gradient of the tile is exactly zero: cnerf_field_backward_ex) — the
// flags are requested one phase ahead, like every other input of the pipeline, so the decision costs no memory round trip
template <int NGEO, bool SKIP>
__global__ void __launch_bounds__(FLD_THREADS) k_field_bwd_x2(const void *__restrict__ enc, const float *__restrict__ xyz, const float *__restrict__ dirs,
                                                              uint32_t dir_group, uint32_t P_, FieldDims dm, const float *__restrict__ pnet,
                                                              const float *__restrict__ pden, const float *__restrict__ prgb,
                                                              const float *__restrict__ g_sigma, const float *__restrict__ g_rgbc,
                                                              void *__restrict__ grad_enc, float *__restrict__ partials, uint32_t ablate_arg,
                                                              const uint8_t *__restrict__ tile_live, const void *__restrict__ wimg) {
#ifdef CNERF_TUNING
    const uint32_t ablate = ablate_arg;                       // measurement aid, tuning builds only
#else
    constexpr uint32_t ablate = 0;                            // release build: the ablation / role-timing branches fold away
    (void)ablate_arg;
#endif
    constexpr bool H = true;
    constexpr int SENC = 2;                                   // enc_pad == 32
    using PR = Prec<H>;
    using frag_t = typename PR::frag_t;
    using elem_t = typename PR::elem_t;
    __shared__ __attribute__((aligned(16))) elem_t x2_w[FLD_HID * (32 + 3 * FLD_HID + (FLD_HID + FLD_DIR)) + 2 * 32 * FLD_HID];
    extern __shared__ __attribute__((aligned(16))) unsigned char fld_lds[];
    elem_t *wl = x2_w;
    const FieldLds lo = fld_lds_layout<H>(dm);
    const MmOff po = x4_offsets(dm);

    constexpr uint32_t S64 = FLD_HID / PR::KS, SDIR = FLD_DIR / PR::KS, SR0 = S64 + SDIR;
    const uint32_t in_r0 = FLD_HID + FLD_DIR;
    const float *n0 = pnet, *n1 = pnet + FLD_HID * dm.enc_pad;
    const float *n2 = n1 + (NGEO == 2 ? FLD_HID * FLD_HID : 0);
    const float *d0 = pden, *dO = pden + FLD_HID * FLD_HID;
    const float *r0 = prgb, *rO = prgb + FLD_HID * in_r0;

    if (wimg) {
        // the packed fp16 image of the forward (field.hip: k_field_pack), 16-byte chunks into the bank-swizzled slots of fb_stage_layer<.., SWZ>:
        // half index i -> i ^ (((i >> 8) & 3) << 5), i.e. chunk c -> c ^ (((c >> 5) & 3) << 2) (every layer starts at a multiple of 2048 halves)
        const uint4 *src = reinterpret_cast<const uint4 *>(wimg);
        uint4 *dst = reinterpret_cast<uint4 *>(wl);
        for (uint32_t c = threadIdx.x; c < lo.off[7] / 8; c += FLD_THREADS) dst[c ^ (((c >> 5) & 3u) << 2)] = src[c];
    } else
    for (int rep = 0; rep < ((ablate & 64) ? 2 : 1); rep++) {                    // (bit 6: staged TWICE — the delta is the staging time)
    fb_stage_layer<H, 0, true>(wl + lo.off[0], n0, FLD_HID, dm.enc_pad, 2, SENC, dm.enc_pad);
    if (NGEO == 2) fb_stage_layer<H, 1, true>(wl + lo.off[1], n1, FLD_HID, FLD_HID, 2, S64, FLD_HID);
    fb_stage_layer<H, 1, true>(wl + lo.off[2], n2, FLD_HID, FLD_HID, 2, S64, FLD_HID);
    fb_stage_layer<H, 1, true>(wl + lo.off[3], d0, FLD_HID, FLD_HID, 2, S64, FLD_HID);
    fb_stage_layer<H, 1, true>(wl + lo.off[4], dO, 16, FLD_HID, 1, S64, FLD_HID);
    fb_stage_layer<H, 2, true>(wl + lo.off[5], r0, FLD_HID, in_r0, 2, SR0, in_r0);
    fb_stage_layer<H, 1, true>(wl + lo.off[6], rO, 16, FLD_HID, 1, S64, FLD_HID);
    if (ablate & 64) __syncthreads();
    }
    // the second K-step of the output-gradient images (features 16..31 of their padded 32-row tile) is never written: zero it once
    for (uint32_t i = threadIdx.x; i < 2 * 2 * (X4_K / 4); i += FLD_THREADS) {
        const uint32_t pr_ = i / (2 * (X4_K / 4)), w = i % (2 * (X4_K / 4));
        const uint32_t which = w / (X4_K / 4), k = w % (X4_K / 4);
        reinterpret_cast<uint32_t *>(fld_lds + pr_ * X2_PAIR_BYTES + X2_B + (which ? X2_B_BDO : X2_B_BRO) + X4_K)[k] = 0u;
    }
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63, li = lane & 31, hi = lane >> 5;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t pair = wave >> 1, role = wave & 1;
    unsigned char *xch = fld_lds + pair * X2_PAIR_BYTES;
    const unsigned char *wb = reinterpret_cast<const unsigned char *>(x2_w);
    [[maybe_unused]] unsigned long long tw_ = 0, tb_ = 0;
    const uint32_t n_tiles = (P_ + FLD_TILE - 1) / FLD_TILE;
    const uint32_t G = gridDim.x * 2, gp = blockIdx.x * 2 + pair;                   // tile pipelines of the whole launch / this one
    const uint32_t n_iter = (n_tiles + G - 1) / G;                                  // workgroup-uniform: same barrier count for every wave
    // Phase p of the launch works on tiles [p G, (p + 1) G): G consecutive tiles, two per workgroup.  WHICH two rotates with the phase: workgroup b
    // takes pair (b + 97 p) mod gridDim.x of the window.  With the plain assignment (pair b in every phase) a workgroup walks tiles 2b, 2b + 1
    // (+ k G): on the run() path (G = 512 tiles = two image rows of coarse samples) always the same image columns, and with early termination
    // (dead tiles cost a barrier and nothing else) the workgroups of the columns that see the object do the live tiles while the others idle —
    // the launch takes as long as its fullest column.  The rotation walks every workgroup across all columns and keeps what the plain
    // assignment has: the tiles in flight at one time are one contiguous window, and the two pipelines of a workgroup — they share the phase
    // barrier, so a phase is free only when BOTH tiles are dead — get neighbouring tiles, which die together.  (Measured alternatives on the
    // fitted field, 352 us plain: a multiplicative permutation of single tiles 418 us, of tile pairs 407 us — scattered windows.)  The
    // assignment does not depend on which tiles are live: the weight-gradient sums of a pipeline add the same tiles in the same order with
    // and without the flags.
    auto tile_at = [&](uint32_t p) { return 2 * (p * gridDim.x + (blockIdx.x + p * 97u) % gridDim.x) + pair; };
    // Early termination (SKIP): a window whose two tiles in this workgroup are both dead (cnerf_composite_run_backward_indexed_flush: every row's
    // output gradient exactly zero) is not a step of the pipelines at all.  The liveness of the workgroup's tile pairs is held 64 windows at a
    // time in a scalar register pair (lane l fetches the two flags of window base + l, one ballot), and every wave walks the set bits: step k of
    // the software pipeline works on the k-th LIVE window p_k.  All four waves build the same masks, so they agree on the sequence and on the
    // number of barriers.  What a dead pair still owes — zero rows of d(loss)/d(grid features) — wave A stores when it loads the mask.
    // History, fitted field (55 % of the tiles dead), 437 us without flags: a flag fetched one phase ahead and dead tiles skipped IN PLACE
    // 401 us (one exposed memory latency per dead phase), flags as masks 352 us, + rotation 328 us — in place, a dead tile between two live
    // ones empties one wave's slot of the phase but not the phase (profiles/r05_field_bwd_dead_phase.json: alternating live / dead windows cost
    // 0.8 of all-live); compacted, the pipeline only ever sees live windows.  A dead tile whose neighbour lives is processed like a live one
    // (all its products are zeros: same sums, same rows).  Without flags every window is live: the same code, the same order of additions.
    constexpr uint32_t END = 0xFFFFFFFFu;
    struct LiveIt { uint32_t base; uint64_t m; };
    auto pair_mask = [&](uint32_t base, bool fill) -> uint64_t {
        const uint32_t ph = base + lane;
        uint32_t f = 0;
        if (ph < n_iter) {
            if (SKIP) {
                const uint32_t t0 = tile_at(ph) - pair;                             // the pair's first tile
                if (t0 < n_tiles) f = tile_live[t0];
                if (t0 + 1 < n_tiles) f |= tile_live[t0 + 1];
            } else {
                f = 1;
            }
        }
        const uint64_t m = __ballot(f != 0);
        if (SKIP && fill) {
            const uint32_t nv = min(64u, n_iter - base);
            uint64_t dead = ~m & (nv == 64 ? ~0ull : (1ull << nv) - 1);
            while (dead) {
                const uint32_t tile = tile_at(base + (uint32_t)__builtin_ctzll(dead));
                dead &= dead - 1;
                const uint32_t row = tile * FLD_TILE + li;
                if (row < P_) {
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const uint32_t level = (uint32_t)fld_rho(r, (int)hi) >> 1;
                        if (level < dm.L) reinterpret_cast<uint32_t *>(grad_enc)[(size_t)level * P_ + row] = 0u;
                    }
                }
            }
        }
        return m;
    };
    auto it_next = [&](LiveIt &it, bool fill) -> uint32_t {                         // the next live window, END when there is none (and from then on)
        while (it.m == 0) {
            if (it.base + 64 >= n_iter) return END;
            it.base += 64;
            it.m = pair_mask(it.base, fill);
        }
        const uint32_t b = (uint32_t)__builtin_ctzll(it.m);
        it.m &= it.m - 1;
        return it.base + b;
    };
    float *part = partials + (size_t)gp * po.total;
    const uint32_t off_n0 = lo.off[0] * 2, off_n1 = lo.off[1] * 2, off_n2 = lo.off[2] * 2, off_d0 = lo.off[3] * 2, off_dO = lo.off[4] * 2,
                   off_r0 = lo.off[5] * 2, off_rO = lo.off[6] * 2;
    const X4Img lw = x4_lane_off_w(lane, false), lwn = x4_lane_off_w(lane, true);
    const X4Img lc = x4_lane_off_img(lane, false), ln = x4_lane_off_img(lane, true);

    if (role == 0) {
        // ======================================================================== wave A: geometry network + dW_n2 / n1 / n0
        cn_f16v wn2[2][2], wn1[2][2], wn0[2];
#pragma unroll
        for (int a = 0; a < 2; a++) {
            x4_zero(wn0[a]);
#pragma unroll
            for (int b = 0; b < 2; b++) { x4_zero(wn2[a][b]); x4_zero(wn1[a][b]); }
        }
        struct ASet { frag_t x0[SENC], h1[4], h2[4]; uint32_t p; bool v, live; };
        ASet S0, S1;                                          // activations of the tiles with even / odd phase index (forward at p, backward at p+2)
#pragma unroll
        for (int s = 0; s < SENC; s++) S0.x0[s] = S1.x0[s] = PR::zero();
#pragma unroll
        for (int s = 0; s < 4; s++) S0.h1[s] = S1.h1[s] = S0.h2[s] = S1.h2[s] = PR::zero();
        S0.p = S1.p = 0; S0.v = S1.v = false; S0.live = S1.live = false;
        frag_t N0[SENC], N1[SENC];                            // grid features requested one phase ahead (even / odd tiles)
        LiveIt it;
        it.base = 0; it.m = pair_mask(0, true);
        uint32_t pk = it_next(it, true);                      // window of step k (forward), END beyond the last
        uint32_t kend = pk == END ? 0u : END;                 // number of live windows, once known
        x4_enc_request<SENC>(enc, P_, dm.L, (pk == END ? 0u : tile_at(pk)) * FLD_TILE + li, hi, N0);
#pragma unroll
        for (int s = 0; s < SENC; s++) N1[s] = PR::zero();
        unsigned char *my = xch + X2_A;
        auto phase = [&](uint32_t k, ASet &S, frag_t (&xcur)[SENC], frag_t (&xnext)[SENC]) __attribute__((always_inline)) {
            X4_T0();
            asm volatile("" ::: "memory");
            const uint32_t pn = pk == END ? END : it_next(it, true);                // window of step k + 1
            if (pn == END && kend == END) kend = k + 1;
            x4_enc_request<SENC>(enc, P_, dm.L, (pn == END ? 0u : tile_at(pn)) * FLD_TILE + li, hi, xnext);   // (beyond the last: some tile, never used)
            // ---- backward of the tile of step k-2 (dz_3 was published by wave B in step k-1); S still holds that tile
            if (!(ablate & 1) && S.live) {
                const unsigned char *z3i = xch + X2_Z3 + (k & 1) * 4 * X4_K;
                frag_t z3[4];
                x4_fetch<4>(z3i, lane, z3);
                x4_publish<SENC>(my + X2_A_X0, lane, S.x0);
                x4_publish<4>(my + X2_A_H1, lane, S.h1);
                if (NGEO == 2) x4_publish<4>(my + X2_A_H2, lane, S.h2);
                cn_f16v acc[2];
                fb_zero(acc);
                x4_gemm_T<2, S64, S64>(wb + off_n2, lw, z3, acc);
                {   // dW_n2 = dz3 . hlast^T
                    const unsigned char *hl = my + ((NGEO == 2) ? X2_A_H2 : X2_A_H1);
                    cn_h8 z[2][2], a[2][2];
                    x4_load_block(z3i, lc, 0, z[0]); x4_load_block(z3i, lc, 1, z[1]);
                    x4_load_block(hl, lc, 0, a[0]); x4_load_block(hl, lc, 1, a[1]);
                    x4_dw(wn2[0][0], z[0], a[0]); x4_dw(wn2[0][1], z[0], a[1]); x4_dw(wn2[1][0], z[1], a[0]); x4_dw(wn2[1][1], z[1], a[1]);
                }
                frag_t z1[4];
                if (NGEO == 2) {
                    frag_t z2[4];
                    x4_c_to_b_masked(acc, S.h2, z2);
                    x4_publish<4>(my + X2_A_Z2, lane, z2);
                    fb_zero(acc);
                    x4_gemm_T<2, S64, S64>(wb + off_n1, lw, z2, acc);
                    cn_h8 z[2][2], a[2][2];
                    x4_load_block(my + X2_A_Z2, lc, 0, z[0]); x4_load_block(my + X2_A_Z2, lc, 1, z[1]);
                    x4_load_block(my + X2_A_H1, lc, 0, a[0]); x4_load_block(my + X2_A_H1, lc, 1, a[1]);
                    x4_dw(wn1[0][0], z[0], a[0]); x4_dw(wn1[0][1], z[0], a[1]); x4_dw(wn1[1][0], z[1], a[0]); x4_dw(wn1[1][1], z[1], a[1]);
                }
                x4_c_to_b_masked(acc, S.h1, z1);
                x4_publish<4>(my + X2_A_Z1, lane, z1);
                cn_f16v denc[1];
                fb_zero(denc);
                x4_gemm_T<1, S64, SENC>(wb + off_n0, lwn, z1, denc);
                {
                    cn_h8 z[2][2], a[2];
                    x4_load_block(my + X2_A_Z1, lc, 0, z[0]); x4_load_block(my + X2_A_Z1, lc, 1, z[1]);
                    x4_load_block(my + X2_A_X0, ln, 0, a);
                    x4_dw(wn0[0], z[0], a); x4_dw(wn0[1], z[1], a);
                }
                if (S.v) {
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const uint32_t level = (uint32_t)fld_rho(r, (int)hi) >> 1;
                        if (level < dm.L) {
                            union { cn_h2 h; uint32_t u; } v;
                            v.h = cn_h2{(_Float16)denc[0][r], (_Float16)denc[0][r + 1]};
                            reinterpret_cast<uint32_t *>(grad_enc)[(size_t)level * P_ + S.p] = v.u;
                        }
                    }
                }
            }
            // ---- forward of the tile of step k (overwrites S)
            if (pk != END && !(ablate & 2)) {
                const uint32_t tile = tile_at(pk);
                S.p = tile * FLD_TILE + li;
                S.v = S.p < P_;
                S.live = tile < n_tiles;                                                   // (wave-uniform)
            } else {
                S.live = false; S.v = false;
            }
            if (S.live) {
                x4_enc_mask<SENC>(xcur, dm.L, S.v, hi, S.x0);
                cn_f16v acc[2];
                fb_zero(acc);
                x4_gemm<2, SENC>(wl + lo.off[0], SENC, 0, S.x0, lane, acc);
                x4_c_to_b<true>(acc, S.h1);
                if (NGEO == 2) {
                    fb_zero(acc);
                    x4_gemm<2, S64>(wl + lo.off[1], S64, 0, S.h1, lane, acc);
                    x4_c_to_b<true>(acc, S.h2);
                }
                frag_t fea[4];
                fb_zero(acc);
                x4_gemm<2, S64>(wl + lo.off[2], S64, 0, (NGEO == 2) ? S.h2 : S.h1, lane, acc);
                x4_c_to_b<false>(acc, fea);
                x4_publish<4>(xch + X2_FEA + (k & 1) * 4 * X4_K, lane, fea);
            }
            pk = pn;
            X4_T1();
        };
        for (uint32_t k = 0; kend == END || k < kend + 2; k += 2) {                 // kend live windows + 2 steps to drain, in pairs
            phase(k, S0, N0, N1);
            phase(k + 1, S1, N1, N0);
        }
#pragma unroll
        for (int a = 0; a < 2; a++) {
            x4_store(part, po.n0, dm.enc_pad, 0, 64, dm.enc_pad, a, 0, li, hi, wn0[a]);
#pragma unroll
            for (int b = 0; b < 2; b++) {
                x4_store(part, po.n2, 64, 0, 64, 64, a, b, li, hi, wn2[a][b]);
                if (NGEO == 2) x4_store(part, po.n1, 64, 0, 64, 64, a, b, li, hi, wn1[a][b]);
            }
        }
    } else {
        // ======================================================================== wave B: density and colour heads (tile p-1 in phase p) + their dW
        cn_f16v wro[2], wrd[2], wrf[2][2], wdo[2], wd0[2][2];
#pragma unroll
        for (int a = 0; a < 2; a++) {
            x4_zero(wro[a]); x4_zero(wrd[a]); x4_zero(wdo[a]);
#pragma unroll
            for (int b = 0; b < 2; b++) { x4_zero(wrf[a][b]); x4_zero(wd0[a][b]); }
        }
        // per-sample inputs, requested unconditionally (clamped index) and masked at use: see x4_enc_request
        struct BIn { float x, y, z, gs, dx, dy, dz; float4 gc; uint32_t tile; };
        auto load_in = [&](uint32_t window) __attribute__((always_inline)) {
            BIn r;
            r.tile = window == END ? END : tile_at(window);
            const uint32_t p = min((window == END ? 0u : r.tile) * FLD_TILE + li, P_ - 1);
            r.x = xyz[(size_t)p * 3]; r.y = xyz[(size_t)p * 3 + 1]; r.z = xyz[(size_t)p * 3 + 2];
            r.gs = g_sigma[p];
            r.gc = *reinterpret_cast<const float4 *>(g_rgbc + (size_t)p * 4);
            const float *dp = dirs + (size_t)(p / dir_group) * 3;
            r.dx = dp[0]; r.dy = dp[1]; r.dz = dp[2];
            return r;
        };
        LiveIt it;
        it.base = 0; it.m = pair_mask(0, false);
        uint32_t pk = it_next(it, false);                     // window of step k (this wave works on step k-1's tile)
        uint32_t kend = pk == END ? 0u : END;
        BIn I0 = load_in(pk), I1 = I0;
        unsigned char *my = xch + X2_B;
        const bool dir_uniform = (dir_group % FLD_TILE) == 0;
        auto phase = [&](uint32_t k, const BIn &cur, BIn &nxt) __attribute__((always_inline)) {
            X4_T0();
            asm volatile("" ::: "memory");
            if (k >= 1) {
                if (pk != END) pk = it_next(it, false);
                if (pk == END && kend == END) kend = k;
                nxt = load_in(pk);
            }
            if (k >= 1 && !(ablate & 4) && cur.tile < n_tiles) {
                const uint32_t i = k - 1;
                const uint32_t tile = cur.tile;
                const bool valid = tile * FLD_TILE + li < P_;
                const unsigned char *fe = xch + X2_FEA + (i & 1) * 4 * X4_K;
                frag_t fea[4], dfr[SDIR];
                x4_fetch<4>(fe, lane, fea);
                // one direction per tile (run() path): 27 lanes evaluate one feature each; the 64-byte scratch is the head of the dz_d image,
                // which this wave rewrites later in the phase (its reads of the previous tile's image are already issued: in-order DS)
                if (dir_uniform) fld_dir_frags_uniform(cur.dx, cur.dy, cur.dz, lane, hi, my + X2_B_ZD, dfr);
                else fb_dir_frags_from<H>(cur.dx, cur.dy, cur.dz, valid, hi, dfr);
                x4_publish<SDIR>(my + X2_B_DIR, lane, dfr);
                // ---- forward of both heads
                cn_f16v acc[2], out[1];
                frag_t hd[4], hr[4];
                fb_zero(acc);
                x4_gemm<2, S64>(wl + lo.off[3], S64, 0, fea, lane, acc);
                x4_c_to_b<true>(acc, hd);
                x4_publish<4>(my + X2_B_HD, lane, hd);
                fb_zero(out);
                x4_gemm<1, S64>(wl + lo.off[4], S64, 0, hd, lane, out);
                const float raw = (float)(_Float16)out[0][0];
                fb_zero(acc);
                x4_gemm<2, S64>(wl + lo.off[5], SR0, 0, fea, lane, acc);
                x4_gemm<2, SDIR>(wl + lo.off[5], SR0, S64, dfr, lane, acc);
                x4_c_to_b<true>(acc, hr);
                x4_publish<4>(my + X2_B_HR, lane, hr);
                fb_zero(out);
                x4_gemm<1, S64>(wl + lo.off[6], S64, 0, hr, lane, out);
                // ---- output-layer gradients (sigmoid', clamped exp': provider_utils.py:26-29)
                frag_t bro[1], bdo[1];
                {
                    cn_h8 f = PR::zero(), g = PR::zero();
                    if (valid && hi == 0) {
                        const float x = cur.x, y = cur.y, z = cur.z;
                        const float gg = 5.0f * __expf(-(x * x + y * y + z * z) * (1.0f / 0.08f));
                        g[0] = (_Float16)(cur.gs * __expf(fminf(fmaxf(raw + gg, -15.0f), 15.0f)));
                        const float gcv[4] = {cur.gc.x, cur.gc.y, cur.gc.z, cur.gc.w};
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const float sg = (float)(_Float16)__builtin_amdgcn_rcpf(1.0f + __expf(-out[0][k]));
                            f[k] = (_Float16)((k < (int)dm.n_rgb_out) ? gcv[k] * sg * (1.0f - sg) : 0.0f);
                        }
                    }
                    bro[0] = f; bdo[0] = g;
                }
                x4_publish<1>(my + X2_B_BRO, lane, bro);
                x4_publish<1>(my + X2_B_BDO, lane, bdo);
                cn_f16v dfea[2];
                fb_zero(dfea);
                // ---- colour head
                {
                    frag_t zr[4];
                    fb_zero(acc);
                    x4_gemm_T<2, 1, S64>(wb + off_rO, lw, bro, acc);
                    x4_c_to_b_masked(acc, hr, zr);
                    x4_publish<4>(my + X2_B_ZR, lane, zr);
                    x4_gemm_T<2, S64, SR0>(wb + off_r0, lw, zr, dfea);
                }
                cn_h8 fa[2][2];                                                    // fea^T: second operand of dW_r0 (fea part) and dW_d0
                x4_load_block(fe, lc, 0, fa[0]); x4_load_block(fe, lc, 1, fa[1]);
                {   // dW_rO = d(out) . hr^T, dW_r0 = dz_r . [dir | fea]^T
                    cn_h8 zo[2], z[2][2], a[2][2], d[2];
                    x4_load_block(my + X2_B_BRO, lc, 0, zo);
                    x4_load_block(my + X2_B_HR, lc, 0, a[0]); x4_load_block(my + X2_B_HR, lc, 1, a[1]);
                    x4_dw(wro[0], zo, a[0]); x4_dw(wro[1], zo, a[1]);
                    x4_load_block(my + X2_B_ZR, lc, 0, z[0]); x4_load_block(my + X2_B_ZR, lc, 1, z[1]);
                    x4_load_block(my + X2_B_DIR, ln, 0, d);
                    x4_dw(wrd[0], z[0], d); x4_dw(wrd[1], z[1], d);
                    x4_dw(wrf[0][0], z[0], fa[0]); x4_dw(wrf[0][1], z[0], fa[1]); x4_dw(wrf[1][0], z[1], fa[0]); x4_dw(wrf[1][1], z[1], fa[1]);
                }
                // ---- density head
                {
                    frag_t zd[4];
                    fb_zero(acc);
                    x4_gemm_T<2, 1, S64>(wb + off_dO, lw, bdo, acc);
                    x4_c_to_b_masked(acc, hd, zd);
                    x4_publish<4>(my + X2_B_ZD, lane, zd);
                    x4_gemm_T<2, S64, S64>(wb + off_d0, lw, zd, dfea);
                }
                {   // dW_dO = d(raw) . hd^T, dW_d0 = dz_d . fea^T
                    cn_h8 zo[2], z[2][2], a[2][2];
                    x4_load_block(my + X2_B_BDO, lc, 0, zo);
                    x4_load_block(my + X2_B_HD, lc, 0, a[0]); x4_load_block(my + X2_B_HD, lc, 1, a[1]);
                    x4_dw(wdo[0], zo, a[0]); x4_dw(wdo[1], zo, a[1]);
                    x4_load_block(my + X2_B_ZD, lc, 0, z[0]); x4_load_block(my + X2_B_ZD, lc, 1, z[1]);
                    x4_dw(wd0[0][0], z[0], fa[0]); x4_dw(wd0[0][1], z[0], fa[1]); x4_dw(wd0[1][0], z[1], fa[0]); x4_dw(wd0[1][1], z[1], fa[1]);
                }
                frag_t z3[4];
                x4_c_to_b<false>(dfea, z3);
                x4_publish<4>(xch + X2_Z3 + (i & 1) * 4 * X4_K, lane, z3);
            }
            X4_T1();
        };
        for (uint32_t k = 0; kend == END || k < kend + 2; k += 2) {
            phase(k, I1, I0);
            phase(k + 1, I0, I1);
        }
#pragma unroll
        for (int b = 0; b < 2; b++) {
            x4_store(part, po.rO, 64, 0, dm.n_rgb_out, 64, 0, b, li, hi, wro[b]);
            x4_store(part, po.dO, 64, 0, 1, 64, 0, b, li, hi, wdo[b]);
        }
        // the padding of this pipeline's partial row (output-layer rows beyond the real outputs, the colour layer's columns 91..95): the row is
        // summed entry by entry into the parameter gradients, so it is written in full here instead of being zero-filled before the launch
        // (50 MB of memset per backward)
        for (uint32_t i = dm.n_rgb_out * 64 + lane; i < 16 * 64; i += 64) part[po.rO + i] = 0.0f;
        for (uint32_t i = 64 + lane; i < 16 * 64; i += 64) part[po.dO + i] = 0.0f;
        for (uint32_t i = lane; i < 64 * (96 - FLD_NDIR - 64); i += 64) {
            const uint32_t row = i / (96 - FLD_NDIR - 64), col = FLD_NDIR + 64 + i % (96 - FLD_NDIR - 64);
            part[po.r0 + row * 96 + col] = 0.0f;
        }
#pragma unroll
        for (int a = 0; a < 2; a++) {
            x4_store(part, po.r0, 96, 0, 64, FLD_NDIR, a, 0, li, hi, wrd[a]);
#pragma unroll
            for (int b = 0; b < 2; b++) {
                x4_store(part, po.r0, 96, FLD_NDIR, 64, 64, a, b, li, hi, wrf[a][b]);
                x4_store(part, po.d0, 64, 0, 64, 64, a, b, li, hi, wd0[a][b]);
            }
        }
    }
#ifdef CNERF_TUNING
    if ((ablate & 32) && lane == 0) {                         // role-timing dump: the last two rows of the FF_MAX_BLOCKS-row workspace (blocks <= 255 then)
        unsigned long long *tt = reinterpret_cast<unsigned long long *>(partials + (size_t)510 * po.total) + ((size_t)blockIdx.x * 4 + wave) * 2;
        tt[0] = tw_; tt[1] = tb_;
    }
#endif
}

// ------------------------------------------------------------------------------------------------ host entry (called from field_bwd_fused.hip)
void ff_reduce_partials(const float *partials, uint32_t n_partials, uint32_t total, uint32_t n_net, uint32_t n_den, float *g_net, float *g_den, float *g_rgb,
                        hipStream_t st);

bool x2_eligible(const FieldDims &dm) {
    static const int on = cn_tune_env("CNERF_FIELD_X2_BWD", 1);   // 0 (tuning builds): the four-wave kernel of field_bwd_fused.hip
    return on && dm.enc_pad == 32 && (dm.n_hidden_geo == 1 || dm.n_hidden_geo == 2);
}

int x2_launch(const void *enc, const float *xyz, const float *dirs, uint32_t dir_group, uint32_t P_, const FieldDims &dm, const float *pnet, const float *pden,
              const float *prgb, const float *g_sigma, const float *g_rgbc, void *grad_enc, float *g_net, float *g_den, float *g_rgb, void *workspace,
              uint32_t max_partials, const uint8_t *tile_live, hipStream_t st, const void *wimg) {
    const FieldLds lo = fld_lds_layout<true>(dm);
    if (lo.off[7] > FLD_HID * (32 + 3 * FLD_HID + (FLD_HID + FLD_DIR)) + 2 * 32 * FLD_HID) return CNERF_EINVAL;
    const uint32_t lds_bytes = 2 * X2_PAIR_BYTES;                               // dynamic part; the weight fragments are a 48 KiB static array
    const uint32_t n_tiles = cn_div_up(P_, FLD_TILE);
    uint32_t blocks = cn_div_up(n_tiles, 2);
    if (blocks > max_partials / 2) blocks = max_partials / 2;                   // two partial-gradient rows per workgroup
    if (blocks > 256) blocks = 256;
    const MmOff po = x4_offsets(dm);
    float *partials = reinterpret_cast<float *>(workspace);
    // (every pipeline writes its whole partial row, padding included: no zero fill)
    // measurement aid of tuning builds only (bit 0 A-backward, 1 A-forward, 2 B switched off — results wrong —, 5 role timing); the release
    // kernel is compiled with the mask fixed at 0 (X2_ABLATE below)
    static const int ablate = cn_tune_env("CNERF_X2_ABLATE", 0);
    if ((ablate & 32) && blocks > 255) blocks = 255;                          // rows 510 / 511 of the workspace hold the timing dump
#define X2_LAUNCH(KERN)                                                                                                                    \
    {                                                                                                                                      \
        auto kern = KERN;                                                                                                                  \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);            \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(FLD_THREADS), lds_bytes, st, enc, xyz, dirs, dir_group, P_, dm, pnet, pden, prgb, g_sigma, g_rgbc, \
                           grad_enc, partials, (uint32_t)ablate, tile_live, wimg);                                                             \
    }
    cn_stage(4, st);
    if (tile_live) {
        if (dm.n_hidden_geo == 2) X2_LAUNCH((k_field_bwd_x2<2, true>)) else X2_LAUNCH((k_field_bwd_x2<1, true>))
    } else {
        if (dm.n_hidden_geo == 2) X2_LAUNCH((k_field_bwd_x2<2, false>)) else X2_LAUNCH((k_field_bwd_x2<1, false>))
    }
    cn_stage(5, st);
    int rc = cn_launch_status();
    if (rc) return rc;
    ff_reduce_partials(partials, blocks * 2, po.total, po.d0, po.r0 - po.d0, g_net, g_den, g_rgb, st);
    cn_stage(6, st);
    return cn_launch_status();
}
