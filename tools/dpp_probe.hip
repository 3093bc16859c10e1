#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out) {
    const int lane = threadIdx.x;
    const float x = (float)lane;
    const int xv = __builtin_bit_cast(int, x);
    out[lane] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, xv, 0x111, 0xf, 0xf, true));
    out[64 + lane] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, xv, 0x101, 0xf, 0xf, true));
}
int main() {
    float* d; hipMalloc(&d, 128 * 4);
    k<<<1, 64>>>(d);
    float h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("row_shr:1 (0x111):"); for (int i = 0; i < 20; ++i) printf(" %g", h[i]); printf("\n");
    printf("row_shl:1 (0x101):"); for (int i = 0; i < 20; ++i) printf(" %g", h[64 + i]); printf("\n");
    return 0;
}
