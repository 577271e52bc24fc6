// How fast can 1790 workgroups each stream their own contiguous ~600 KB (the accumulate kernel's record read pattern)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int THREADS>
__global__ void __launch_bounds__(THREADS) k_read(const uint4* __restrict__ src, uint32_t per_wg16, uint32_t* out) {
    extern __shared__ unsigned char lds[];
    const uint4* p = src + (size_t)blockIdx.x * per_wg16;
    uint32_t acc = 0;
    uint32_t i = threadIdx.x;
    for (; i + 3 * THREADS < per_wg16; i += 4 * THREADS) {
        uint4 r[4];
#pragma unroll
        for (int u = 0; u < 4; u++) r[u] = p[i + u * THREADS];
#pragma unroll
        for (int u = 0; u < 4; u++) acc += r[u].x ^ r[u].y ^ r[u].z ^ r[u].w;
    }
    for (; i < per_wg16; i += THREADS) { const uint4 r = p[i]; acc += r.x ^ r.y; }
    if (acc == 0x12345u) { lds[threadIdx.x] = 1; out[blockIdx.x] = lds[0]; }
}
__global__ void k_write(uint4* dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
template <int THREADS> void run(const uint4* buf, uint32_t* out, uint32_t wgs, uint32_t per_wg16, uint32_t lds, uint4* wbuf, size_t n16, bool fresh) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipFuncSetAttribute((const void*)&k_read<THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 16);
    float best = 1e9;
    for (int rep = 0; rep < 4; rep++) {
        if (fresh) k_write<<<4096, 256>>>(wbuf, n16);
        (void)hipEventRecord(a);
        k_read<THREADS><<<wgs, THREADS, lds>>>(buf, per_wg16, out);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    const double bytes = (double)wgs * per_wg16 * 16;
    printf("threads %4d wgs %5u lds %5u %s: %7.3f ms  %6.2f TB/s\n", THREADS, wgs, lds, fresh ? "fresh " : "reread", best, bytes / (best * 1e-3) / 1e12);
}
int main() {
    const size_t total = 1073741824ull;               // 1 GiB of records
    uint4* buf; (void)hipMalloc(&buf, total);
    uint32_t* out; (void)hipMalloc(&out, 1 << 20);
    k_write<<<4096, 256>>>(buf, total / 16);
    (void)hipDeviceSynchronize();
    for (int fresh = 0; fresh < 2; fresh++) {
        run<1024>(buf, out, 1790, (uint32_t)(total / 16 / 1790), 65536 + 16, buf, total / 16, fresh);
        run<1024>(buf, out, 1790, (uint32_t)(total / 16 / 1790), 0, buf, total / 16, fresh);
        run<256>(buf, out, 7160, (uint32_t)(total / 16 / 7160), 0, buf, total / 16, fresh);
        run<1024>(buf, out, 512, (uint32_t)(total / 16 / 512), 65536 + 16, buf, total / 16, fresh);
        run<256>(buf, out, 2048, (uint32_t)(total / 16 / 2048), 0, buf, total / 16, fresh);
    }
    return 0;
}
