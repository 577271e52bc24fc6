#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float rn_dpp(float old, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL, ROW_MASK, 0xF, false));
}
__device__ float incl_sum(float x) {
    x += rn_dpp<0x111>(0.0f, x); x += rn_dpp<0x112>(0.0f, x); x += rn_dpp<0x114>(0.0f, x); x += rn_dpp<0x118>(0.0f, x);
    x += rn_dpp<0x142, 0xA>(0.0f, x); x += rn_dpp<0x143, 0xC>(0.0f, x); return x;
}
__global__ void k(const float* in, float* out_incl, float* out_excl) {
    const float v = in[threadIdx.x];
    const float i = incl_sum(v);
    out_incl[threadIdx.x] = i;
    out_excl[threadIdx.x] = rn_dpp<0x138>(-1.0f, i);
}
int main() {
    float h[64], *d, *o1, *o2, r1[64], r2[64];
    for (int i = 0; i < 64; i++) h[i] = (float)(i + 1);
    hipMalloc(&d, 256); hipMalloc(&o1, 256); hipMalloc(&o2, 256);
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o1, o2);
    hipMemcpy(r1, o1, 256, hipMemcpyDeviceToHost); hipMemcpy(r2, o2, 256, hipMemcpyDeviceToHost);
    int bad = 0; float s = 0;
    for (int i = 0; i < 64; i++) { float prev = s; s += h[i]; if (r1[i] != s) { bad++; printf("incl[%d]=%g want %g\n", i, r1[i], s); } if (r2[i] != (i ? prev : -1.0f)) { bad++; printf("excl[%d]=%g want %g\n", i, r2[i], i ? prev : -1.0f); } }
    printf("bad=%d\n", bad);
    return 0;
}
