// microbenchmark: float atomic add throughput by memory scope on gfx950 (random addresses in a table)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
template <int SCOPE>
__global__ void k_atomic(float* table, uint32_t mask, uint32_t per_thread, uint32_t xcd_pin, uint32_t slice) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = tid * 2654435761u + 12345u;
    // xcd_pin: each XCD (blockIdx%8) touches its own slice of the table
    uint32_t base = xcd_pin ? (blockIdx.x % 8) * slice : 0;
    for (uint32_t i = 0; i < per_thread; i++) {
        s = s * 1664525u + 1013904223u;
        uint32_t idx = base + ((s >> 8) & mask);
        if (SCOPE == 0) __hip_atomic_fetch_add(&table[idx], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (SCOPE == 1) __hip_atomic_fetch_add(&table[idx], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (SCOPE == 2) __hip_atomic_fetch_add(&table[idx], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        else table[idx] += 1.0f;  // plain RMW (racy) for reference
    }
}
template <int SCOPE>
double run(float* d, uint32_t mask, uint32_t pin, uint32_t slice, const char* name) {
    const uint32_t blocks = 4096, threads = 256, per = 64;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k_atomic<SCOPE><<<blocks, threads>>>(d, mask, per, pin, slice);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; r++) k_atomic<SCOPE><<<blocks, threads>>>(d, mask, per, pin, slice);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double n = 5.0 * blocks * threads * per;
    printf("%-28s mask=%8u pin=%u : %8.2f G atomics/s (%.3f ms per launch)\n", name, mask + 1, pin, n / (ms * 1e-3) / 1e9, ms / 5);
    return n / (ms * 1e-3);
}
int main() {
    float* d; size_t n = 64u << 20; hipMalloc(&d, n * 4); hipMemset(d, 0, n * 4);
    for (uint32_t lg : {20u, 22u}) {           // 1M floats = 4 MB ; 4M floats = 16 MB
        uint32_t mask = (1u << lg) - 1;
        for (uint32_t pin : {0u, 1u}) {
            run<0>(d, mask, pin, mask + 1, "agent scope");
            run<1>(d, mask, pin, mask + 1, "workgroup scope");
            run<2>(d, mask, pin, mask + 1, "wavefront scope");
            run<3>(d, mask, pin, mask + 1, "plain rmw (racy)");
        }
    }
    // verify sum for workgroup scope with pinning (each XCD its own slice) -> must be exact
    hipMemset(d, 0, n * 4);
    k_atomic<1><<<4096, 256>>>(d, (1u << 20) - 1, 64, 1, 1u << 20);
    hipDeviceSynchronize();
    std::vector<float> h(8u << 20); hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    double sum = 0; for (float v : h) sum += v;
    printf("workgroup-scope pinned sum = %.0f (expected %.0f)\n", sum, 4096.0 * 256 * 64);
    hipMemset(d, 0, n * 4);
    k_atomic<1><<<4096, 256>>>(d, (1u << 20) - 1, 64, 0, 1u << 20);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    sum = 0; for (size_t i = 0; i < (1u << 20); i++) sum += h[i];
    printf("workgroup-scope UNPINNED sum = %.0f (expected %.0f; smaller => lost updates across XCD L2s)\n", sum, 4096.0 * 256 * 64);
    return 0;
}
