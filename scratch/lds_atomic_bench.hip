#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ void __launch_bounds__(1024) k(const uint32_t* __restrict__ idx, float* out, uint32_t n_per_wg, uint32_t mask) {
    extern __shared__ float acc[];
    for (uint32_t i = threadIdx.x; i < 16384; i += 1024) acc[i] = 0;
    __syncthreads();
    const uint32_t* p = idx + (size_t)blockIdx.x * n_per_wg;
    for (uint32_t i = threadIdx.x; i < n_per_wg; i += 1024) {
        uint32_t e = p[i] & mask;
        if (MODE == 0) { unsafeAtomicAdd(&acc[e * 2], 1.0f); unsafeAtomicAdd(&acc[e * 2 + 1], 2.0f); }
        else if (MODE == 1) { atomicAdd((uint32_t*)&acc[e * 2], 1u); atomicAdd((uint32_t*)&acc[e * 2 + 1], 2u); }
        else if (MODE == 2) { unsafeAtomicAdd(&acc[e], 1.0f); unsafeAtomicAdd(&acc[e + 8192], 2.0f); }       // SoA channels
        else if (MODE == 3) { acc[e * 2] += 1.0f; acc[e * 2 + 1] += 2.0f; }                                    // racy plain
        else if (MODE == 4) { atomicAdd((unsigned long long*)&acc[e * 2], 0x0000000200000001ull); }           // one 64-bit int atomic
        else if (MODE == 5) { atomicAdd((uint32_t*)&acc[e], 1u); }
        else if (MODE == 6) { atomicAdd((unsigned long long*)&acc[(e & 4095) * 4], 3ull); atomicAdd((unsigned long long*)&acc[(e & 4095) * 4 + 2], 5ull); }
        else if (MODE == 7) { unsafeAtomicAdd(&acc[e], 1.0f); }
        else if (MODE == 8) { __builtin_amdgcn_ds_atomic_fadd_v2f16((__attribute__((address_space(3))) __attribute__((ext_vector_type(2))) _Float16*)(&acc[e]), (__attribute__((ext_vector_type(2))) _Float16){(_Float16)1.0f, (_Float16)2.0f}); }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 16384; i += 1024) if (acc[i] != 0) out[blockIdx.x * 16384 + i] = acc[i];
}
template <int MODE> void run(const uint32_t* d_idx, float* d_out, uint32_t mask, const char* name) {
    const uint32_t wgs = 1024, per = 262144;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipFuncSetAttribute((const void*)&k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 16);
    k<MODE><<<wgs, 1024, 65536>>>(d_idx, d_out, per, mask);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int r = 0; r < 3; r++) k<MODE><<<wgs, 1024, 65536>>>(d_idx, d_out, per, mask);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("%-34s mask=%5u : %7.3f ms per launch, %7.1f G records/s\n", name, mask + 1, ms / 3, 3.0 * wgs * per / (ms * 1e-3) / 1e9);
}
int main() {
    size_t n = 1024ull * 262144;
    uint32_t* h = (uint32_t*)malloc(n * 4);
    uint32_t s = 1; for (size_t i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; h[i] = s >> 8; }
    uint32_t* d; (void)hipMalloc(&d, n * 4); (void)hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    float* out; (void)hipMalloc(&out, 1024ull * 16384 * 4);
    for (uint32_t mask : {8191u, 255u, 7u}) {
        run<0>(d, out, mask, "ds_add_f32 x2 (AoS)");
        run<1>(d, out, mask, "ds_add_u32 x2 (AoS)");
        run<2>(d, out, mask, "ds_add_f32 x2 (SoA)");
        run<3>(d, out, mask, "plain rmw x2 (racy)");
        run<4>(d, out, mask, "ds_add_u64 x1");
        run<5>(d, out, mask, "ds_add_u32 x1");
        run<6>(d, out, mask, "ds_add_u64 x2 (AoS, 4096 entries)");
        run<7>(d, out, mask, "ds_add_f32 x1");
        run<8>(d, out, mask, "ds_pk_add_f16 x1");
    }
    return 0;
}
