// InstanceNorm statistics: per (batch, channel) mean and 1/sqrt(biased var + eps) over D*H*W.
// Replaces the reduction inside nn.InstanceNorm3d(affine=False) (reference code/models/blocks.py:18,
// applied at blocks.py:313).  The normalisation itself is fused into the consuming conv's loader.
// Channels-last input: a thread keeps one 4-channel vector lane for the whole sweep (coalesced 16 B
// loads, several in flight), lanes that share a channel vector are combined with wavefront shuffles,
// the four waves through LDS, and a block leaves with ONE f64 atomic per (channel, moment) so that
// E[x^2] - E[x]^2 is evaluated without cancellation trouble.
#include "common.h"
#include "passion_hip.h"

namespace {

template <typename T, int VEC>
__global__ __launch_bounds__(256) void stats_partial_kernel(const T* __restrict__ x, long V, int C, long bstride,
                                                            double* __restrict__ scratch) {
    __shared__ float red[4 * 64 * 2 * VEC];
    const int b = blockIdx.y;
    const int cvn = C / VEC;            // channel vectors per voxel; a power of two <= 256
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long nvec = V * cvn;
    const T* xb = x + (long)b * bstride;
    float s[VEC], q[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { s[j] = 0.f; q[j] = 0.f; }
    // stride is a multiple of cvn, so a thread always sees the same VEC channels
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + tid;
    for (; i + 3 * stride < nvec; i += 4 * stride) {   // four independent vector loads in flight
        float v0[VEC], v1[VEC], v2[VEC], v3[VEC];
        VecIO<T, VEC>::load(xb + i * VEC, v0); VecIO<T, VEC>::load(xb + (i + stride) * VEC, v1);
        VecIO<T, VEC>::load(xb + (i + 2 * stride) * VEC, v2); VecIO<T, VEC>::load(xb + (i + 3 * stride) * VEC, v3);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            s[j] += (v0[j] + v1[j]) + (v2[j] + v3[j]);
            q[j] += (v0[j] * v0[j] + v1[j] * v1[j]) + (v2[j] * v2[j] + v3[j] * v3[j]);
        }
    }
    for (; i < nvec; i += stride) {
        float v[VEC];
        VecIO<T, VEC>::load(xb + i * VEC, v);
#pragma unroll
        for (int j = 0; j < VEC; ++j) { s[j] += v[j]; q[j] += v[j] * v[j]; }
    }
    // lanes l and l ^ o share a channel vector whenever o is a multiple of cvn
    const int w = cvn < 64 ? cvn : 64;   // distinct channel vectors inside one wave
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        if (o >= w) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) { s[j] += __shfl_xor(s[j], o, 64); q[j] += __shfl_xor(q[j], o, 64); }
        }
    }
    if (lane < w) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) { red[(wv * 64 + lane) * 2 * VEC + j] = s[j]; red[(wv * 64 + lane) * 2 * VEC + VEC + j] = q[j]; }
    }
    __syncthreads();
    // channel vector cv lives in (wave wv, lane l) with (wv*64 + l) % cvn == cv
    if (tid < cvn) {
        double ds[VEC], dq[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) { ds[j] = 0; dq[j] = 0; }
        for (int t = tid; t < 256; t += cvn) {
            const int l = t & 63;
            if (l >= w) continue;
#pragma unroll
            for (int j = 0; j < VEC; ++j) { ds[j] += (double)red[t * 2 * VEC + j]; dq[j] += (double)red[t * 2 * VEC + VEC + j]; }
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {   // this block's slot of the partial slab: no atomics, reproducible
            scratch[(((long)blockIdx.x * gridDim.y + b) * C + tid * VEC + j) * 2 + 0] = ds[j];
            scratch[(((long)blockIdx.x * gridDim.y + b) * C + tid * VEC + j) * 2 + 1] = dq[j];
        }
    }
}

// 256 threads = 16 (b, c) outputs x 16 slices of the R partial slots; fixed summation order
__global__ __launch_bounds__(256) void stats_finalize_kernel(const double* __restrict__ part, int R, long V, int n, float eps,
                                                             float* __restrict__ stats) {
    __shared__ double rs[16][17], rq[16][17];
    const int li = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + li;
    double s = 0, q = 0;
    if (i < n)
        for (int r = sl; r < R; r += 16) { s += part[((long)r * n + i) * 2]; q += part[((long)r * n + i) * 2 + 1]; }
    rs[sl][li] = s; rq[sl][li] = q;
    __syncthreads();
    if (sl == 0 && i < n) {
        s = 0; q = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { s += rs[k][li]; q += rq[k][li]; }
        const double mean = s / (double)V;
        double var = q / (double)V - mean * mean;
        if (var < 0) var = 0;
        stats[i * 2 + 0] = (float)mean;
        stats[i * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// Small tensors (the deep levels: 8^3 ... 32^3): every conv there is a short dependent chain conv -> partial -> finalize
// -> conv of launches that are pure latency, so the statistics take ONE launch: a block owns `cvb` channel vectors of one
// sample over all voxels -- nothing to combine across blocks, the block finalises its own channels.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void stats_small_kernel(const T* __restrict__ x, long V, int C, long bstride, int cvb, float eps,
                                                          float* __restrict__ stats) {
    __shared__ float red[256 * 2 * VEC];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int cvn = C / VEC;                 // channel vectors per voxel
    const int j = tid % cvb, lane = tid / cvb, nlane = 256 / cvb;      // cvb is a power of two <= 64
    const int cv = blockIdx.x * cvb + j;
    const T* xb = x + (long)b * bstride + (long)cv * VEC;
    float s[VEC], q[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { s[i] = 0.f; q[i] = 0.f; }
    long v = lane;
    for (; v + 3 * nlane < V; v += 4 * nlane) {          // four independent vector loads in flight
        float v0[VEC], v1[VEC], v2[VEC], v3[VEC];
        VecIO<T, VEC>::load(xb + v * C, v0); VecIO<T, VEC>::load(xb + (v + nlane) * C, v1);
        VecIO<T, VEC>::load(xb + (v + 2 * nlane) * C, v2); VecIO<T, VEC>::load(xb + (v + 3 * nlane) * C, v3);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            s[i] += (v0[i] + v1[i]) + (v2[i] + v3[i]);
            q[i] += (v0[i] * v0[i] + v1[i] * v1[i]) + (v2[i] * v2[i] + v3[i] * v3[i]);
        }
    }
    for (; v < V; v += nlane) {
        float t[VEC];
        VecIO<T, VEC>::load(xb + v * C, t);
#pragma unroll
        for (int i = 0; i < VEC; ++i) { s[i] += t[i]; q[i] += t[i] * t[i]; }
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) { red[tid * 2 * VEC + i] = s[i]; red[tid * 2 * VEC + VEC + i] = q[i]; }
    __syncthreads();
    // thread (j, i): channel i of vector j, summed over the nlane voxel lanes in a fixed order, in double
    for (int o = tid; o < cvb * VEC; o += 256) {
        const int jj = o / VEC, i = o % VEC;
        double ds = 0, dq = 0;
        for (int l = 0; l < nlane; ++l) {
            ds += (double)red[(l * cvb + jj) * 2 * VEC + i];
            dq += (double)red[(l * cvb + jj) * 2 * VEC + VEC + i];
        }
        const double mean = ds / (double)V;
        double var = dq / (double)V - mean * mean;
        if (var < 0) var = 0;
        const int c = (blockIdx.x * cvb + jj) * VEC + i;
        stats[((long)b * C + c) * 2 + 0] = (float)mean;
        stats[((long)b * C + c) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// blocks per sample for the one-launch path (0: tensor too large, use partial + finalize)
inline int small_blocks(long V, int cvn) {
    const long vecs = V * cvn;
    if (vecs > 4 * 32768) return 0;
    int nb = 1;                                   // as many blocks as the channel vectors allow until a block has <= 4096 vectors:
    while (nb < cvn && (vecs / nb > 4096 || cvn / nb > 64)) nb *= 2;      // a single block is a 128-deep serial chain at 16^3 x 64
    return (vecs / nb <= 32768 && cvn % nb == 0 && cvn / nb <= 64) ? nb : 0;
}

inline int stats_blocks(long V, int C) {
    const long nvec = V * (C / 4);
    long gx = (nvec + 256 * 16 - 1) / (256 * 16);
    return (int)(gx < 1 ? 1 : (gx > 256 ? 256 : gx));
}

}  // namespace

extern "C" int64_t passion_instnorm_scratch_doubles(int B, int64_t V, int C) {
    return (int64_t)stats_blocks((long)V, C) * B * C * 2;
}

extern "C" int passion_instnorm_stats(const void* x, int dtype, int B, int64_t V, int C, int64_t bstride, float eps,
                                      double* scratch, float* stats, void* stream) {
    if (C % 4 || (256 % (C / 4)) != 0) return -1010;
    hipStream_t st = (hipStream_t)stream;
    {
        const bool v8 = dtype == PASSION_BF16 && C % 8 == 0 && bstride % 8 == 0;
        const int cvn = C / (v8 ? 8 : 4);
        const int nb = (dtype == PASSION_F32 || dtype == PASSION_BF16) ? small_blocks((long)V, cvn) : 0;
        if (nb > 0) {
            dim3 grid(nb, B);
            if (dtype == PASSION_F32)
                hipLaunchKernelGGL((stats_small_kernel<float, 4>), grid, dim3(256), 0, st, (const float*)x, (long)V, C, (long)bstride, cvn / nb, eps, stats);
            else if (v8)
                hipLaunchKernelGGL((stats_small_kernel<bf16_t, 8>), grid, dim3(256), 0, st, (const bf16_t*)x, (long)V, C, (long)bstride, cvn / nb, eps, stats);
            else
                hipLaunchKernelGGL((stats_small_kernel<bf16_t, 4>), grid, dim3(256), 0, st, (const bf16_t*)x, (long)V, C, (long)bstride, cvn / nb, eps, stats);
            LAUNCH_RET();
            return 0;
        }
    }
    const int gx = stats_blocks((long)V, C);
    dim3 grid(gx, B);
    if (dtype == PASSION_F32)
        hipLaunchKernelGGL((stats_partial_kernel<float, 4>), grid, dim3(256), 0, st, (const float*)x, (long)V, C, (long)bstride, scratch);
    else if (dtype == PASSION_BF16 && C % 8 == 0 && bstride % 8 == 0)
        hipLaunchKernelGGL((stats_partial_kernel<bf16_t, 8>), grid, dim3(256), 0, st, (const bf16_t*)x, (long)V, C, (long)bstride, scratch);
    else if (dtype == PASSION_BF16)
        hipLaunchKernelGGL((stats_partial_kernel<bf16_t, 4>), grid, dim3(256), 0, st, (const bf16_t*)x, (long)V, C, (long)bstride, scratch);
    else
        return -1007;
    LAUNCH_RET();
    const int n = B * C;
    hipLaunchKernelGGL(stats_finalize_kernel, dim3((n + 15) / 16), dim3(256), 0, st, scratch, gx, (long)V, n, eps, stats);
    LAUNCH_RET();
    return 0;
}
