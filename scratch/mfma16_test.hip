// layout check of v_mfma_f32_16x16x32_f16 on gfx950: A lane (i = l&15, g = l>>4) holds A[i][8g+j]; B lane (n, g) holds B[8g+j][n];
// C lane (n, g) holds C[4g+r][n].  Asymmetric random operands, compared with a host product.   hipcc --offload-arch=gfx950 -O2 mfma16_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const _Float16 *A, const _Float16 *B, float *C) {
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    h8 a, b;
    for (int j = 0; j < 8; j++) { a[j] = A[i * 32 + 8 * g + j]; b[j] = B[(8 * g + j) * 16 + i]; }
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) C[(4 * g + r) * 16 + i] = c[r];
}
int main() {
    _Float16 hA[16 * 32], hB[32 * 16];
    float ref[256] = {0}, hC[256];
    srand(1);
    for (int x = 0; x < 512; x++) { hA[x] = (_Float16)((rand() % 17 - 8) / 8.0f); hB[x] = (_Float16)((rand() % 13 - 6) / 4.0f); }
    for (int i = 0; i < 16; i++) for (int n = 0; n < 16; n++) for (int k2 = 0; k2 < 32; k2++) ref[i * 16 + n] += (float)hA[i * 32 + k2] * (float)hB[k2 * 16 + n];
    _Float16 *dA, *dB; float *dC;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
    double e = 0;
    for (int x = 0; x < 256; x++) e = fmax(e, fabs(hC[x] - ref[x]));
    printf("mfma_f32_16x16x32_f16 layout check: max|diff| = %g (%s)\n", e, e < 1e-3 ? "OK" : "MISMATCH");
    return e < 1e-3 ? 0 : 1;
}
