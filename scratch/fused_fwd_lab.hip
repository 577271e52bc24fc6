// Fused-forward laboratory (round 5, VERDICT r4 item 1 "measure before building"): can the hash-grid gather keep its speed when the work list is
// SAMPLE-major (a workgroup evaluates all 16 levels of its sample tile — what a gather -> field-MLP fusion needs) instead of today's level-major,
// XCD-sliced list (each XCD walks ~2 levels, whose tables stay hot in its 4 MiB L2)?  Every variant is checked bit for bit against the
// production kernel on the benchmark's ray-structured samples.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math scratch/fused_fwd_lab.hip -o scratch/fused_fwd_lab
#include "../customnerf_amd/csrc/gridencoder.hip"
#include <vector>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <random>
#include <functional>

bool bn_eligible(uint32_t, uint32_t, uint32_t, uint32_t, const GridLevels &) { return false; }
uint64_t bn_workspace_bytes(uint32_t, uint32_t, const GridLevels &, int) { return 0; }
int bn_backward(const void *, const float *, const GridLevels &, float *, uint32_t, uint32_t, uint32_t, int, uint32_t, int, void *, hipStream_t, bool) { return -1; }
int bn_prepare(const float *, const GridLevels &, uint32_t, uint32_t, uint32_t, int, uint32_t, int, void *, hipStream_t) { return -1; }
int bn_prepare_rows(const float *, const GridLevels &, uint32_t, uint32_t, uint32_t, int, uint32_t, int, void *, hipStream_t, uint32_t, uint32_t) { return -1; }
int bn_prepare_finish(const GridLevels &, uint32_t, uint32_t, int, void *, hipStream_t) { return -1; }
uint32_t bn_hist_block_points(int) { return 0; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// ---- variant A: one workgroup = one tile of 256 * SPT samples, all levels, non-persistent, dispatch order (block b -> XCD b % 8).
//      LOOP 0: level loop outside, samples inside (a level's table is used for the whole tile before the next level starts)
//      LOOP 1: sample loop outside, the 16 levels of a sample unrolled (more loads in flight per thread)
template <int SPT, int LOOP, int WPS>
__global__ void __launch_bounds__(256, WPS) k_sm_tile(const float *__restrict__ inputs, const __half *__restrict__ grid, const GridLevels lv,
                                                     __half *__restrict__ outputs, uint32_t B, uint32_t nl, uint32_t ostride) {
    const uint32_t t0 = blockIdx.x * (256 * SPT);
    float in[SPT][3];
    bool ok[SPT];
#pragma unroll
    for (int s = 0; s < SPT; s++) {
        const uint32_t b = min(t0 + s * 256 + threadIdx.x, B - 1);
        ge_load_coords<3>(inputs, b, in[s]);
        ok[s] = !(in[s][0] < 0 || in[s][0] > 1 || in[s][1] < 0 || in[s][1] > 1 || in[s][2] < 0 || in[s][2] > 1);
    }
    uint32_t *out = reinterpret_cast<uint32_t *>(outputs);
    if (LOOP == 0) {
        for (uint32_t i = 0; i < nl; i++) {
            const uint32_t level = lv.order[i];
            const unsigned char *table = reinterpret_cast<const unsigned char *>(grid) + (size_t)lv.offset[level] * 4;
#pragma unroll
            for (int s = 0; s < SPT; s++) {
                const uint32_t b = t0 + s * 256 + threadIdx.x;
                if (b < B) out[(size_t)level * ostride + b] = ok[s] ? gf_eval_level(in[s], table, lv.size[level], lv.resolution[level], lv.scale[level], 0u) : 0u;
            }
        }
    } else {
#pragma unroll
        for (int s = 0; s < SPT; s++) {
            const uint32_t b = t0 + s * 256 + threadIdx.x;
#pragma unroll 4
            for (uint32_t i = 0; i < nl; i++) {
                const uint32_t level = lv.order[i];
                const unsigned char *table = reinterpret_cast<const unsigned char *>(grid) + (size_t)lv.offset[level] * 4;
                if (b < B) out[(size_t)level * ostride + b] = ok[s] ? gf_eval_level(in[s], table, lv.size[level], lv.resolution[level], lv.scale[level], 0u) : 0u;
            }
        }
    }
}

// ---- variant B: persistent, XCD-lockstep.  gridDim = 8 * WG_PER_XCD; XCD x (= blockIdx % 8) owns the contiguous sample range x / 8 of the list;
//      its workgroups walk that range in rounds of WG_PER_XCD tiles, every workgroup sweeping the levels in the same order — so at any moment an
//      XCD's CUs gather from the same one or two tables (no barrier: they start together and do the same work per level).
//      The results of a tile go to LDS ([level][sample] dwords, as the fused kernel would keep them) and are streamed out at the end of the tile
//      (stand-in for the MLP's consumption; the production [L, B] buffer is written so that the check against the production kernel holds).
//      OFFSET: XCD x starts its level sweep at position 2 x of the order (each XCD works on different tables: the L2s do not duplicate).
template <int SPT, int WPS, bool OFFSET>
__global__ void __launch_bounds__(256, WPS) k_sm_persist(const float *__restrict__ inputs, const __half *__restrict__ grid, const GridLevels lv,
                                                        __half *__restrict__ outputs, uint32_t B, uint32_t nl, uint32_t ostride) {
    extern __shared__ uint32_t sm_lds[];                     // [nl][256 * SPT]
    constexpr uint32_t TS = 256 * SPT;
    const uint32_t xcd = blockIdx.x % CN_NXCD, k = blockIdx.x / CN_NXCD, wgx = gridDim.x / CN_NXCD;
    const uint32_t per = (B + CN_NXCD - 1) / CN_NXCD;
    const uint32_t r0 = xcd * per, r1 = min(B, r0 + per);
    uint32_t *out = reinterpret_cast<uint32_t *>(outputs);
    for (uint32_t t0 = r0 + k * TS; t0 < r1; t0 += wgx * TS) {
        float in[SPT][3];
        bool ok[SPT];
#pragma unroll
        for (int s = 0; s < SPT; s++) {
            const uint32_t b = min(t0 + s * 256 + threadIdx.x, B - 1);
            ge_load_coords<3>(inputs, b, in[s]);
            ok[s] = !(in[s][0] < 0 || in[s][0] > 1 || in[s][1] < 0 || in[s][1] > 1 || in[s][2] < 0 || in[s][2] > 1);
        }
        for (uint32_t i = 0; i < nl; i++) {
            const uint32_t pos = OFFSET ? (i + 2 * xcd) % nl : i;
            const uint32_t level = lv.order[pos];
            const unsigned char *table = reinterpret_cast<const unsigned char *>(grid) + (size_t)lv.offset[level] * 4;
#pragma unroll
            for (int s = 0; s < SPT; s++)
                sm_lds[level * TS + s * 256 + threadIdx.x] = ok[s] ? gf_eval_level(in[s], table, lv.size[level], lv.resolution[level], lv.scale[level], 0u) : 0u;
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < nl * TS; i += 256) {
            const uint32_t level = i / TS, j = i % TS, b = t0 + j;
            if (b < r1) out[(size_t)level * ostride + b] = sm_lds[i];
        }
        __syncthreads();
    }
}


// ---- variant C: the enc tile produced DIRECTLY in MFMA A-operand layout (no LDS staging at all): a wave owns 32 samples; lane (li, hi) gathers
//      the levels 8 s + 4 hi + {0..3}, s = 0, 1 of sample li — exactly the eight half2 values of its two 32x32x16 A fragments (k = 16 s + 8 hi + e) —
//      so the wave that gathers can feed the MLP from registers.  Free-running waves: every table is touched all the time (no level lockstep).
//      SPW: 32-sample groups per wave in flight together (independent loads)
template <int SPW, int WPS>
__global__ void __launch_bounds__(256, WPS) k_sm_frag(const float *__restrict__ inputs, const __half *__restrict__ grid, const GridLevels lv,
                                                     __half *__restrict__ outputs, uint32_t B, uint32_t nl, uint32_t ostride) {
    const uint32_t lane = threadIdx.x & 63, li = lane & 31, hi = lane >> 5, wave = threadIdx.x >> 6;
    const uint32_t g0 = (blockIdx.x * 4 + wave) * SPW;                              // first 32-sample group of this wave
    uint32_t *out = reinterpret_cast<uint32_t *>(outputs);
    float in[SPW][3];
    bool ok[SPW];
#pragma unroll
    for (int q = 0; q < SPW; q++) {
        const uint32_t b = min((g0 + q) * 32 + li, B - 1);
        ge_load_coords<3>(inputs, b, in[q]);
        ok[q] = !(in[q][0] < 0 || in[q][0] > 1 || in[q][1] < 0 || in[q][1] > 1 || in[q][2] < 0 || in[q][2] > 1);
    }
#pragma unroll
    for (int s = 0; s < 2; s++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t level = 8 * s + 4 * hi + j;
            if (level >= nl) continue;
            const unsigned char *table = reinterpret_cast<const unsigned char *>(grid) + (size_t)lv.offset[level] * 4;
#pragma unroll
            for (int q = 0; q < SPW; q++) {
                const uint32_t b = (g0 + q) * 32 + li;
                if (b < B) out[(size_t)level * ostride + b] = ok[q] ? gf_eval_level(in[q], table, lv.size[level], lv.resolution[level], lv.scale[level], 0u) : 0u;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ host
struct Scene { std::vector<float> unit; uint32_t B; };
static Scene make_samples(uint32_t HW, uint32_t S, bool fine_like) {
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
        const uint32_t py = ri / HW, px = ri % HW;
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

int main() {
    const uint32_t L = 16, H = 16, T19 = 1u << 19;
    const float S = log2f(expf(logf(2048.0f / 16.0f) / 15.0f));
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
    std::vector<uint32_t> tab(off);
    std::mt19937 rng(1);
    for (auto &v : tab) { const __half a = __float2half(((rng() & 0xFFFF) / 65536.0f - 0.5f)), b = __float2half(((rng() & 0xFFFF) / 65536.0f - 0.5f)); v = (uint32_t)__half_as_ushort(a) | ((uint32_t)__half_as_ushort(b) << 16); }
    uint32_t *d_tab; CK(hipMalloc(&d_tab, (size_t)off * 4)); CK(hipMemcpy(d_tab, tab.data(), (size_t)off * 4, hipMemcpyHostToDevice));
    GridLevels lv;
    if (ge_levels(offs.data(), L, L, S, H, lv)) { printf("ge_levels failed\n"); return 1; }

    for (int fine = 0; fine < 2; fine++) {
        Scene sc = make_samples(128, 64, fine != 0);
        const uint32_t B = sc.B;
        float *d_in; CK(hipMalloc(&d_in, (size_t)B * 12)); CK(hipMemcpy(d_in, sc.unit.data(), (size_t)B * 12, hipMemcpyHostToDevice));
        __half *d_ref, *d_out; CK(hipMalloc(&d_ref, (size_t)L * B * 4)); CK(hipMalloc(&d_out, (size_t)L * B * 4));
        printf("---- %s samples, B = %u\n", fine ? "importance-like (draw order)" : "stratified coarse", B);
        auto prod = [&] { cnerf_grid_encode_forward_strided(d_in, d_tab, offs.data(), d_ref, B, 3, 2, L, L, S, H, nullptr, 0, 0, 0, CNERF_F16, B, nullptr); };
        const float t_all = time_it(prod);
        printf("%-72s %7.1f us  (%.3f of 8 TB/s)\n", "production k_grid_fwd_fast (level-major, XCD-sliced)", t_all, (double)B * 588 / (t_all * 1e-6) / 8e12);
        prod(); CK(hipDeviceSynchronize());
        std::vector<uint32_t> ref((size_t)L * B), got((size_t)L * B);
        CK(hipMemcpy(ref.data(), d_ref, (size_t)L * B * 4, hipMemcpyDeviceToHost));
        auto check_time = [&](const char *name, const std::function<void()> &go) {
            CK(hipMemset(d_out, 0xFF, (size_t)L * B * 4));
            go(); CK(hipDeviceSynchronize()); CK(hipGetLastError());
            CK(hipMemcpy(got.data(), d_out, (size_t)L * B * 4, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (size_t i = 0; i < (size_t)L * B; i++) bad += got[i] != ref[i];
            const float t = time_it(go);
            printf("%-72s %7.1f us  (%.3f)  mismatches %zu\n", name, t, (double)B * 588 / (t * 1e-6) / 8e12, bad);
        };
        GridLevels nat = lv;                                   // natural level order 0..15 (lv.order = coarse / fine interleave)
        for (uint32_t i = 0; i < L; i++) nat.order[i] = (uint8_t)i;
#define TILE(SPT, LOOP, WPS, LV, NAME) check_time(NAME, [&] { hipLaunchKernelGGL((k_sm_tile<SPT, LOOP, WPS>), dim3((B + 256 * SPT - 1) / (256 * SPT)), dim3(256), 0, 0, d_in, (const __half *)d_tab, LV, d_out, B, L, B); })
        TILE(1, 0, 1, lv, "A  sample-major tile 256, level loop, dispatch order, interleaved levels");
        TILE(1, 0, 1, nat, "A  sample-major tile 256, level loop, natural level order");
        TILE(1, 1, 1, lv, "A  sample-major tile 256, levels unrolled x4 per sample");
        TILE(4, 0, 1, lv, "A  sample-major tile 1024 (4 samples / thread), level loop");
        TILE(4, 0, 2, lv, "A  sample-major tile 1024, level loop, <= 128 VGPR");
        TILE(8, 0, 1, lv, "A  sample-major tile 2048 (8 samples / thread), level loop");
#define PERS(SPT, WPS, OFS, WGX, LV, NAME) do { \
            auto kern = k_sm_persist<SPT, WPS, OFS>; const uint32_t ldsb = L * 256 * SPT * 4; \
            CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb)); \
            check_time(NAME, [&] { hipLaunchKernelGGL(kern, dim3(8 * (WGX)), dim3(256), ldsb, 0, d_in, (const __half *)d_tab, LV, d_out, B, L, B); }); } while (0)
        PERS(1, 1, false, 32 * 8, lv, "B  persistent XCD-lockstep, tile 256, 8 WG/CU (16 KB LDS each)");
        PERS(2, 1, false, 32 * 4, lv, "B  persistent XCD-lockstep, tile 512, 4 WG/CU (32 KB)");
        PERS(4, 1, false, 32 * 2, lv, "B  persistent XCD-lockstep, tile 1024, 2 WG/CU (64 KB)");
        PERS(4, 1, false, 32 * 2, nat, "B  persistent XCD-lockstep, tile 1024, 2 WG/CU, natural level order");
        PERS(8, 1, false, 32 * 1, lv, "B  persistent XCD-lockstep, tile 2048, 1 WG/CU (128 KB)");
        PERS(2, 1, true, 32 * 4, lv, "B  persistent, tile 512, 4 WG/CU, XCD x starts at level position 2x");
        PERS(4, 1, true, 32 * 2, lv, "B  persistent, tile 1024, 2 WG/CU, XCD x starts at level position 2x");
        PERS(1, 1, true, 32 * 8, lv, "B  persistent, tile 256, 8 WG/CU, XCD x starts at level position 2x");
#define FRAG(SPW, WPS, NAME) check_time(NAME, [&] { hipLaunchKernelGGL((k_sm_frag<SPW, WPS>), dim3((B + 128 * SPW - 1) / (128 * SPW)), dim3(256), 0, 0, d_in, (const __half *)d_tab, lv, d_out, B, L, B); })
        FRAG(1, 1, "C  MFMA-fragment layout: wave = 32 samples, lane = 8 levels, free-running");
        FRAG(2, 1, "C  MFMA-fragment layout, 2 sample groups per wave");
        FRAG(4, 1, "C  MFMA-fragment layout, 4 sample groups per wave");
        FRAG(1, 2, "C  MFMA-fragment layout, <= 128 VGPR");
        FRAG(2, 2, "C  MFMA-fragment layout, 2 groups, <= 128 VGPR");
        CK(hipFree(d_in)); CK(hipFree(d_ref)); CK(hipFree(d_out));
    }
    return 0;
}
