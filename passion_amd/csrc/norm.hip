// InstanceNorm statistics: per (batch, channel) mean and 1/sqrt(biased var + eps) over D*H*W.
// Replaces the reduction inside nn.InstanceNorm3d(affine=False) (reference code/models/blocks.py:18,
// applied at blocks.py:313).  The normalisation itself is fused into the consuming conv's loader.
// Channels-last input: a thread keeps one 4-channel vector lane for the whole sweep (coalesced 16 B
// loads), block partials are combined with wavefront shuffles + LDS and leave the block as f64
// atomics so that E[x^2] - E[x]^2 is evaluated without cancellation trouble.
#include "common.h"
#include "passion_hip.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void stats_partial_kernel(const T* __restrict__ x, long V, int C, long bstride,
                                                            double* __restrict__ scratch) {
    __shared__ float red[256 * 8];
    const int b = blockIdx.y;
    const int cvn = C / 4;              // channel vectors per voxel; divides 256
    const int tid = threadIdx.x;
    const long nvec = V * cvn;
    const T* xb = x + (long)b * bstride;
    float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
    // stride is a multiple of cvn, so a thread always sees the same 4 channels
    for (long i = (long)blockIdx.x * 256 + tid; i < nvec; i += (long)gridDim.x * 256) {
        f32x4 v = load4(xb + i * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { s[j] += v[j]; q[j] += v[j] * v[j]; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[tid * 8 + j] = s[j]; red[tid * 8 + 4 + j] = q[j]; }
    __syncthreads();
    if (tid < cvn) {
        double ds[4] = {0, 0, 0, 0}, dq[4] = {0, 0, 0, 0};
        for (int r = tid; r < 256; r += cvn) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { ds[j] += (double)red[r * 8 + j]; dq[j] += (double)red[r * 8 + 4 + j]; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            atomicAdd(&scratch[((long)b * C + tid * 4 + j) * 2 + 0], ds[j]);
            atomicAdd(&scratch[((long)b * C + tid * 4 + j) * 2 + 1], dq[j]);
        }
    }
}

__global__ void stats_finalize_kernel(const double* __restrict__ scratch, long V, int n, float eps,
                                      float* __restrict__ stats) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double mean = scratch[i * 2] / (double)V;
    double var = scratch[i * 2 + 1] / (double)V - mean * mean;
    if (var < 0) var = 0;
    stats[i * 2 + 0] = (float)mean;
    stats[i * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

}  // namespace

extern "C" int passion_instnorm_stats(const void* x, int dtype, int B, int64_t V, int C, int64_t bstride, float eps,
                                      double* scratch, float* stats, void* stream) {
    if (C % 4 || (256 % (C / 4)) != 0) return -1010;
    hipStream_t st = (hipStream_t)stream;
    HIP_RET(hipMemsetAsync(scratch, 0, sizeof(double) * 2 * (size_t)B * C, st));
    const long nvec = V * (C / 4);
    int gx = (int)hmin((nvec + 256 * 8 - 1) / (256 * 8), 1024);
    if (gx < 1) gx = 1;
    dim3 grid(gx, B);
    if (dtype == PASSION_F32)
        hipLaunchKernelGGL(stats_partial_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (long)V, C, (long)bstride, scratch);
    else if (dtype == PASSION_BF16)
        hipLaunchKernelGGL(stats_partial_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, (long)V, C, (long)bstride, scratch);
    else
        return -1007;
    LAUNCH_RET();
    const int n = B * C;
    hipLaunchKernelGGL(stats_finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, st, scratch, (long)V, n, eps, stats);
    LAUNCH_RET();
    return 0;
}
