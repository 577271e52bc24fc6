// LDS cycles of ds_read_b64_tr_b16 for candidate V-tile layouts of the attention kernel (gfx950): one wave per SIMD, 8 reads per iteration.
// hipcc -O3 --offload-arch=gfx950 tr_bank_bench.hip -o tr_bank_bench && ./tr_bank_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short s4 __attribute__((__vector_size__(4 * sizeof(short))));
__global__ void __launch_bounds__(256) k(const unsigned *addr, long long *cycles, short *sink, int iters, int plain) {
    extern __shared__ __attribute__((aligned(16))) short lds[];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = (short)i;
    __syncthreads();
    const unsigned a = addr[threadIdx.x & 63];
    s4 acc = {0, 0, 0, 0};
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const short *p = lds + (a >> 1) + u * 2048;
            s4 v;
            if (plain) v = *(const s4 *)p;
            else v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3))) *)p);
            acc ^= v;
        }
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (acc[0] == 12345 && acc[1] == 3) sink[threadIdx.x] = acc[2];
}
int main() {
    struct P { const char *name; unsigned (*f)(unsigned); };
    P pats[] = {
        {"linear l*8", [](unsigned l) { return l * 8; }},
        {"row-major pitch 192 (now, DT=2/3)", [](unsigned l) { unsigned i = l & 15, G = (l >> 4) & 1, hi = l >> 5; return (4 * hi + (i >> 2)) * 192 + (16 * G + 4 * (i & 3)) * 2; }},
        {"row-major pitch 128", [](unsigned l) { unsigned i = l & 15, G = (l >> 4) & 1, hi = l >> 5; return (4 * hi + (i >> 2)) * 128 + (16 * G + 4 * (i & 3)) * 2; }},
        {"row-major pitch 144", [](unsigned l) { unsigned i = l & 15, G = (l >> 4) & 1, hi = l >> 5; return (4 * hi + (i >> 2)) * 144 + (16 * G + 4 * (i & 3)) * 2; }},
        {"row-major pitch 160", [](unsigned l) { unsigned i = l & 15, G = (l >> 4) & 1, hi = l >> 5; return (4 * hi + (i >> 2)) * 160 + (16 * G + 4 * (i & 3)) * 2; }},
        {"row-major pitch 136", [](unsigned l) { unsigned i = l & 15, G = (l >> 4) & 1, hi = l >> 5; return (4 * hi + (i >> 2)) * 136 + (16 * G + 4 * (i & 3)) * 2; }},
        {"row-major pitch 96 (3 x 32 B)", [](unsigned l) { unsigned i = l & 15, G = (l >> 4) & 1, hi = l >> 5; return (4 * hi + (i >> 2)) * 96 + (16 * G + 4 * (i & 3)) * 2; }},
        {"row-major pitch 64", [](unsigned l) { unsigned i = l & 15, G = (l >> 4) & 1, hi = l >> 5; return (4 * hi + (i >> 2)) * 64 + (16 * G + 4 * (i & 3)) * 2; }},
        {"subtile [cb][32 keys][16 ch]: G*1024 + hi*128 + i*8", [](unsigned l) { unsigned i = l & 15, G = (l >> 4) & 1, hi = l >> 5; return G * 1024 + hi * 128 + i * 8; }},
        {"subtile, groups 512 apart (guide)", [](unsigned l) { return (l & 15) * 8 + (l >> 4) * 512; }},
        {"subtile, G*1024+64 pad: G*1088 + hi*128 + i*8", [](unsigned l) { unsigned i = l & 15, G = (l >> 4) & 1, hi = l >> 5; return G * 1088 + hi * 128 + i * 8; }},
        {"subtile, G*1152 + hi*128 + i*8", [](unsigned l) { unsigned i = l & 15, G = (l >> 4) & 1, hi = l >> 5; return G * 1152 + hi * 128 + i * 8; }},
        {"groups contiguous: (l>>4)*128 + i*8", [](unsigned l) { return (l >> 4) * 128 + (l & 15) * 8; }},
    };
    unsigned *d_addr; long long *d_cyc; short *d_sink;
    hipMalloc(&d_addr, 64 * 4); hipMalloc(&d_cyc, 8); hipMalloc(&d_sink, 512);
    const int iters = 2000;
    for (int plain = 0; plain < 2; plain++)
        for (auto &p : pats) {
            std::vector<unsigned> h(64);
            for (unsigned l = 0; l < 64; l++) h[l] = p.f(l);
            hipMemcpy(d_addr, h.data(), 256, hipMemcpyHostToDevice);
            long long c = 0;
            for (int rep = 0; rep < 2; rep++) {
                hipLaunchKernelGGL(k, dim3(1), dim3(256), 40960, 0, d_addr, d_cyc, d_sink, iters, plain);
                hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost);
            }
            printf("%s %-55s %6.2f clk / wave-instruction (4 waves on the CU)\n", plain ? "b64   " : "tr_b16", p.name, (double)c / (iters * 8.0));
        }
    return 0;
}
