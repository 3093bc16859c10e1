// probe of ds_read_b64_tr_b16 lane semantics (gfx950): prints, for every lane, the 4 values it receives when
// lane 4q+p of each 16-lane group supplies the address of row q, columns 4p..4p+3 of a [row][16] ushort image.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short img[64 * 16];
    for (int i = threadIdx.x; i < 64 * 16; i += 64) img[i] = (unsigned short)i;   // value = row*16 + col
    __syncthreads();
    const int l = threadIdx.x, G = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
    const int row = 8 * G + q;   // group G reads rows 8G..8G+3
    s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(img + row * 16 + 4 * p));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)t[j];
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l) {
        int G = l >> 4, i = l & 15;
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) {
            int r = h[l * 4 + j] / 16, c = h[l * 4 + j] % 16;
            printf(" (r%d,c%d)", r, c);
            if (r != 8 * G + j || c != i) ok = 0;   // expectation: lane i of group gets column i of rows 8G+0..3
        }
        printf("\n");
    }
    printf("EXPECTATION %s\n", ok ? "HOLDS" : "FAILS");
    return 0;
}
