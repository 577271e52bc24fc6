// does an out-of-range lane of buffer_load_dwordx4 ... lds write zeros to its LDS slot, or nothing?  (implicit-GEMM zero padding)
// hipcc -O3 --offload-arch=gfx950 buffer_lds_oob_test.hip -o buffer_lds_oob_test
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const uint4 *in, uint4 *out, unsigned nbytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const unsigned wave = threadIdx.x >> 6;
    reinterpret_cast<uint4 *>(lds)[threadIdx.x] = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, nbytes, 0x00020000);
    unsigned voff = threadIdx.x * 16;
    if (threadIdx.x & 1) voff = 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(lds + wave * 1024), 16, voff, 0, 0, 0);
    __syncthreads();
    out[threadIdx.x] = *reinterpret_cast<uint4 *>(lds + threadIdx.x * 16);
}
int main() {
    uint4 *din, *dout, h[128], o[128];
    for (int i = 0; i < 128; i++) h[i] = make_uint4(i + 1, i + 1, i + 1, i + 1);
    (void)hipMalloc(&din, sizeof(h)); (void)hipMalloc(&dout, sizeof(o));
    (void)hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(128), 2048, 0, din, dout, (unsigned)sizeof(h));
    (void)hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; i++) printf("lane %d: %08x %08x\n", i, o[i].x, o[i].w);
    int zeros = 0, kept = 0, good = 0;
    for (int i = 0; i < 128; i++) { if (i & 1) { zeros += o[i].x == 0; kept += o[i].x == 0xffffffffu; } else good += o[i].x == (unsigned)(i + 1); }
    printf("in-range lanes correct: %d/64, out-of-range lanes zero-filled: %d/64, left untouched: %d/64\n", good, zeros, kept);
    return 0;
}
