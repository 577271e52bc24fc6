// Does a record slab written by one kernel and read by the next stay on-die (Infinity Cache) when the slab is re-used?
// W: scattered (emit-like: 64 lanes -> 64 bins) or coalesced 8/16-byte stores; R: coalesced read + reduce.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int BYTES, int SCATTER>
__global__ void __launch_bounds__(256) kw(uint8_t* __restrict__ slab, uint64_t nrec, uint32_t nbins_log2) {
    const uint64_t bin_stride = (nrec >> nbins_log2) * BYTES;
    for (uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x; r < nrec; r += (uint64_t)gridDim.x * 256) {
        uint64_t addr;
        if (SCATTER) {
            const uint64_t s = r >> nbins_log2, j = r & ((1u << nbins_log2) - 1);
            const uint64_t bin = (j ^ (s * 2654435761u >> 7)) & ((1u << nbins_log2) - 1);
            addr = bin * bin_stride + s * BYTES;
        } else addr = r * BYTES;
        if (BYTES == 8) *reinterpret_cast<uint2*>(slab + addr) = make_uint2((uint32_t)r, 1u);
        else *reinterpret_cast<uint4*>(slab + addr) = make_uint4((uint32_t)r, 1u, 2u, 3u);
    }
}
__global__ void __launch_bounds__(256) kr(const uint4* __restrict__ slab, uint64_t n16, uint32_t* out) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) {
        const uint4 v = slab[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}
template <int BYTES, int SCATTER>
void run(uint8_t* buf, uint32_t* out, uint64_t total_bytes, uint64_t slab_bytes, int pingpong) {
    const uint64_t npass = total_bytes / slab_bytes, nrec = slab_bytes / BYTES;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(a);
        for (uint64_t p = 0; p < npass; p++) {
            uint8_t* s = buf + (pingpong == 2 ? p * slab_bytes : (pingpong == 1 ? (p & 1) * slab_bytes : 0));
            kw<BYTES, SCATTER><<<4096, 256>>>(s, nrec, 7);
            kr<<<4096, 256>>>((const uint4*)s, slab_bytes / 16, out);
        }
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    printf("rec %2dB %-9s slab %5llu MB x %3llu passes (%s): %7.3f ms  -> %6.2f TB/s (W+R bytes)\n", BYTES, SCATTER ? "scattered" : "coalesced",
           (unsigned long long)(slab_bytes >> 20), (unsigned long long)npass, pingpong == 2 ? "fresh buffer each" : pingpong == 1 ? "ping-pong" : "same slab", best,
           2.0 * total_bytes / (best * 1e-3) / 1e12);
}
int main() {
    const uint64_t total = 2ull << 30;
    uint8_t* buf; (void)hipMalloc(&buf, total);
    uint32_t* out; (void)hipMalloc(&out, 1 << 20);
    (void)hipMemset(buf, 0, total);
    for (uint64_t mb : {2048ull, 256ull, 128ull, 64ull, 32ull}) {
        run<8, 1>(buf, out, total, mb << 20, 0);
        run<8, 0>(buf, out, total, mb << 20, 0);
        run<16, 1>(buf, out, total, mb << 20, 0);
    }
    run<8, 1>(buf, out, total, 64ull << 20, 1);
    run<8, 1>(buf, out, total, 128ull << 20, 1);
    run<8, 1>(buf, out, total, 128ull << 20, 2);
    run<8, 1>(buf, out, total, 64ull << 20, 2);
    // write-only / read-only split at 2 GB
    {
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        float ms;
        (void)hipEventRecord(a); kw<8, 1><<<4096, 256>>>(buf, total / 8, 7); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        (void)hipEventElapsedTime(&ms, a, b); printf("W scattered 8B 2GB: %.3f ms\n", ms);
        (void)hipEventRecord(a); kw<8, 0><<<4096, 256>>>(buf, total / 8, 7); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        (void)hipEventElapsedTime(&ms, a, b); printf("W coalesced 8B 2GB: %.3f ms\n", ms);
        (void)hipEventRecord(a); kw<16, 1><<<4096, 256>>>(buf, total / 16, 7); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        (void)hipEventElapsedTime(&ms, a, b); printf("W scattered 16B 2GB: %.3f ms\n", ms);
        (void)hipEventRecord(a); kr<<<4096, 256>>>((const uint4*)buf, total / 16, out); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        (void)hipEventElapsedTime(&ms, a, b); printf("R 2GB: %.3f ms\n", ms);
    }
    return 0;
}
