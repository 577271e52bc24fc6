// LDS atomic peak without a global record stream: indices come from an in-register LCG.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ void __launch_bounds__(1024) k(float* out, uint32_t iters, uint32_t mask) {
    extern __shared__ unsigned long long acc[];          // 8192 x u64 = 64 KiB
    for (uint32_t i = threadIdx.x; i < 8192; i += 1024) acc[i] = 0;
    __syncthreads();
    uint32_t s = blockIdx.x * 1024 + threadIdx.x + 12345u;
    for (uint32_t i = 0; i < iters; i++) {
        s = s * 1664525u + 1013904223u;
        const uint32_t e = (s >> 10) & mask;             // entry (0..4095)
        if (MODE == 0) { atomicAdd(&acc[e * 2], 3ull); atomicAdd(&acc[e * 2 + 1], 5ull); }                       // 2 x u64, AoS
        else if (MODE == 1) { atomicAdd((uint32_t*)&acc[e * 2], 3u); atomicAdd((uint32_t*)&acc[e * 2 + 1], 5u); } // 2 x u32 at the same addresses
        else if (MODE == 2) { atomicAdd(&acc[e * 2], 3ull); }                                                       // 1 x u64
        else if (MODE == 3) { atomicAdd(&acc[e], 3ull); atomicAdd(&acc[e + 4096], 5ull); }                        // 2 x u64, SoA
        else if (MODE == 4) { atomicAdd(&acc[e * 2], 3ull); atomicAdd(&acc[e * 2 + 1], 5ull); const uint32_t p = e ^ 1; atomicAdd(&acc[p * 2], 3ull); atomicAdd(&acc[p * 2 + 1], 5ull); }
        else if (MODE == 5) { atomicAdd((uint32_t*)&acc[e], 3u); }                                                  // 1 x u32
        else if (MODE == 6) { acc[e * 2] += 3ull; acc[e*2+1] += 5ull; }                                           // racy plain RMW
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 8192; i += 1024) if (acc[i] == 77) out[blockIdx.x] = 1;
}
template <int MODE> void run(float* out, uint32_t mask, int threads, const char* name, int natom) {
    const uint32_t wgs = 2048, iters = 256;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipFuncSetAttribute((const void*)&k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    k<MODE><<<wgs, threads, 65536>>>(out, iters, mask);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int r = 0; r < 3; r++) k<MODE><<<wgs, threads, 65536>>>(out, iters, mask);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double atoms = 3.0 * wgs * threads * iters * natom;
    printf("%-28s threads=%4d mask=%4u : %7.3f ms, %7.1f G atomics/s, %5.2f per clk per CU (2.4 GHz)\n", name, threads, mask + 1, ms / 3, atoms / (ms * 1e-3) / 1e9,
           atoms / (ms * 1e-3) / 256 / 2.4e9);
}
int main() {
    float* out; (void)hipMalloc(&out, 1 << 20);
    for (int threads : {1024, 512}) for (uint32_t mask : {4095u, 63u}) {
        run<0>(out, mask, threads, "u64 x2 AoS", 2);
        run<1>(out, mask, threads, "u32 x2 (stride 8B)", 2);
        run<2>(out, mask, threads, "u64 x1", 1);
        run<3>(out, mask, threads, "u64 x2 SoA", 2);
        run<4>(out, mask, threads, "u64 x4 (pair)", 4);
        run<5>(out, mask, threads, "u32 x1 dense", 1);
        run<6>(out, mask, threads, "plain rmw x2", 2);
    }
    return 0;
}
