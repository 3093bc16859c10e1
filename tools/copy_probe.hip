// which streaming-copy shape reaches the HBM copy rate on this box: blocks x unroll.  hipcc --offload-arch=gfx950 -O3 tools/copy_probe.hip -o tools/copy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
template <int U>
__global__ __launch_bounds__(256) void copy_k(const u4* __restrict__ s, u4* __restrict__ d, long n) {
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        u4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(s + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_nontemporal_store(v[u], d + i + u * stride);
    }
    for (; i < n; i += stride) d[i] = s[i];
}
template <int U> float run(const u4* s, u4* d, long n, int blocks) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) copy_k<U><<<blocks, 256>>>(s, d, n);
    hipEventRecord(a);
    for (int w = 0; w < 5; ++w) copy_k<U><<<blocks, 256>>>(s, d, n);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return 5.0f * 2.0f * n * 16 / (ms * 1e-3f) / 1e9f;
}
int main() {
    const long bytes = 1L << 30, n = bytes / 16;
    u4 *s, *d; hipMalloc((void**)&s, bytes); hipMalloc((void**)&d, bytes); hipMemset(s, 1, bytes);
    for (int blocks : {2048, 4096, 8192, 16384, 65536}) {
        printf("blocks %6d: U1 %.0f  U2 %.0f  U4 %.0f  U8 %.0f GB/s\n", blocks, run<1>(s, d, n, blocks), run<2>(s, d, n, blocks), run<4>(s, d, n, blocks), run<8>(s, d, n, blocks));
    }
    return 0;
}
