// semantics probe of ds_read_b64_tr_b16 (gfx950): every lane of a 16-lane group supplies the address of 4 consecutive 16-bit elements;
// prints which (supplying lane, element) each result element came from.  hipcc -O3 --offload-arch=gfx950 tr_read_test.hip -o tr_read_test
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((__vector_size__(4 * sizeof(short))));
__global__ void k(short *out) {
    __shared__ __attribute__((aligned(16))) short lds[64 * 4];
    const int l = threadIdx.x;
    for (int j = 0; j < 4; j++) lds[l * 4 + j] = (short)(l * 4 + j);          // lane l's own 8-byte chunk holds (l, j) encoded as 4 l + j
    __syncthreads();
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3))) *)(lds + l * 4));
    for (int j = 0; j < 4; j++) out[l * 4 + j] = v[j];
}
int main() {
    short *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l++) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; j++) printf("  (src lane %2d, elem %d)", h[l * 4 + j] / 4, h[l * 4 + j] % 4);
        printf("\n");
    }
    return 0;
}
