// Streaming ops of the m2ftrans backbone that the mmformer / RFNet kernel set does not cover:
//   * grouped 3x3x3 conv with 1 or 4 input channels per group (reference code/models/blocks.py:45-47 DepthWiseConvBlock.conv2,
//     groups = channels; :77-79 GroupConvBlock.dwconv, groups = embed_dims over 4 * embed_dims channels), zero padding, stride 1,
//     channels-last.  The tensors are the 8^3 / 16^3 token volumes of the cross-token blocks (<= 512 channels): a few hundred
//     thousand outputs of 27 * cpg multiply-adds each -- one thread per output, no MFMA (K = 27 or 108 per output channel and no
//     channel mixing across groups: there is no GEMM to speak of);
//   * y = x * w_up: a modality's feature map scaled by its nearest-upsampled (p, p, p) attention weight map (reference
//     code/models/m2ftrans.py:443-458 Weight_Attention), all masked passes (teacher + students) of a PASSION step in one launch;
// (The block-masked attention softmax of code/models/mask.py:5-22 lives with the other attention kernels in attention.hip.)
// HBM-bound elementwise / small-stencil work: coalesced channel-vector accesses, fixed-order reductions (no float atomics).
#include "common.h"
#include "passion_hip.h"

namespace {

template <typename T> __device__ __forceinline__ float ldf(const T* p) { return (float)*p; }
template <typename T> __device__ __forceinline__ void stf(T* p, float v) { *p = (T)v; }

// y[b][v][co] = bias[co] + sum_t sum_j w[co][j][t] x[b][v + t - 1][g cpg + j],  g = co / cpg
template <typename T>
__global__ __launch_bounds__(256) void gconv_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                        T* __restrict__ y, int B, int D, int H, int W, int C, int cpg) {
    const long total = (long)B * D * H * W * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int co = (int)(i % C);
        long v = i / C;
        const int ww = (int)(v % W); v /= W;
        const int hh = (int)(v % H); v /= H;
        const int dd = (int)(v % D);
        const int b = (int)(v / D);
        const int c0 = (co / cpg) * cpg;
        float acc = bias ? bias[co] : 0.f;
        for (int td = 0; td < 3; ++td) {
            const int zd = dd + td - 1;
            if (zd < 0 || zd >= D) continue;
            for (int th = 0; th < 3; ++th) {
                const int zh = hh + th - 1;
                if (zh < 0 || zh >= H) continue;
                for (int tw = 0; tw < 3; ++tw) {
                    const int zw = ww + tw - 1;
                    if (zw < 0 || zw >= W) continue;
                    const T* xp = x + ((((long)b * D + zd) * H + zh) * W + zw) * C + c0;
                    const float* wp = w + ((long)co * cpg) * 27 + (td * 9 + th * 3 + tw);
                    for (int j = 0; j < cpg; ++j) acc = __builtin_fmaf(wp[j * 27], ldf(xp + j), acc);
                }
            }
        }
        stf(y + i, acc);
    }
}

// dx[b][v][ci] = sum_t sum_{co in group(ci)} w[co][ci - g cpg][t] dy[b][v - t + 1][co]
template <typename T>
__global__ __launch_bounds__(256) void gconv_dgrad_kernel(const T* __restrict__ dy, const float* __restrict__ w, T* __restrict__ dx,
                                                          int B, int D, int H, int W, int C, int cpg) {
    const long total = (long)B * D * H * W * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ci = (int)(i % C);
        long v = i / C;
        const int ww = (int)(v % W); v /= W;
        const int hh = (int)(v % H); v /= H;
        const int dd = (int)(v % D);
        const int b = (int)(v / D);
        const int c0 = (ci / cpg) * cpg, jl = ci - c0;
        float acc = 0.f;
        for (int td = 0; td < 3; ++td) {
            const int zd = dd - td + 1;
            if (zd < 0 || zd >= D) continue;
            for (int th = 0; th < 3; ++th) {
                const int zh = hh - th + 1;
                if (zh < 0 || zh >= H) continue;
                for (int tw = 0; tw < 3; ++tw) {
                    const int zw = ww - tw + 1;
                    if (zw < 0 || zw >= W) continue;
                    const T* gp = dy + ((((long)b * D + zd) * H + zh) * W + zw) * C + c0;
                    const float* wp = w + ((long)c0 * cpg + jl) * 27 + (td * 9 + th * 3 + tw);
                    for (int o = 0; o < cpg; ++o) acc = __builtin_fmaf(wp[(long)o * cpg * 27], ldf(gp + o), acc);
                }
            }
        }
        stf(dx + i, acc);
    }
}

// dw[co][j][t] = sum_b sum_v dy[b][v][co] x[b][v + t - 1][g cpg + j]: one wavefront per output, fixed order (lane-strided partial
// sums, butterfly), so the result repeats bit for bit.  db[co] = sum dy (wave of t == 13, j == 0 also carries it).
template <typename T>
__global__ __launch_bounds__(256) void gconv_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ dw,
                                                          float* __restrict__ db, int B, int D, int H, int W, int C, int cpg) {
    const long nout = (long)C * cpg * 27;
    const long o = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= nout) return;
    const int lane = threadIdx.x & 63;
    const int t = (int)(o % 27), j = (int)((o / 27) % cpg), co = (int)(o / (27L * cpg));
    const int td = t / 9 - 1, th = (t / 3) % 3 - 1, tw = t % 3 - 1;
    const int cin = (co / cpg) * cpg + j;
    const long V = (long)D * H * W;
    float acc = 0.f, accb = 0.f;
    for (long q = lane; q < (long)B * V; q += 64) {
        long v = q % V;
        const int b = (int)(q / V);
        const int ww = (int)(v % W); v /= W;
        const int hh = (int)(v % H);
        const int dd = (int)(v / H);
        const float g = ldf(dy + q * C + co);
        accb += g;
        const int zd = dd + td, zh = hh + th, zw = ww + tw;
        if (zd < 0 || zd >= D || zh < 0 || zh >= H || zw < 0 || zw >= W) continue;
        acc = __builtin_fmaf(g, ldf(x + ((((long)b * D + zd) * H + zh) * W + zw) * C + cin), acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) dw[o] = acc;
    if (db && t == 13 && j == 0) {
        accb = wave_sum(accb);
        if (lane == 0) db[co] = accb;
    }
}

// y[b][v][c] = x[b % Bx][v][c] * w[b][v >> k per axis]
template <typename T, int VEC>
__global__ __launch_bounds__(256) void scale_nearest_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, T* __restrict__ y,
                                                                int Bv, int Bx, int D, int H, int W, int C8, int k) {
    const long total = (long)Bv * D * H * W * C8;
    const int d = D >> k, h = H >> k, ws = W >> k;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        long v = i / C8;
        const int c8 = (int)(i % C8);
        const int ww = (int)(v % W); v /= W;
        const int hh = (int)(v % H); v /= H;
        const int dd = (int)(v % D);
        const int b = (int)(v / D);
        const float s = w[(((long)b * d + (dd >> k)) * h + (hh >> k)) * ws + (ww >> k)];
        float xv[VEC];
        VecIO<T, VEC>::load(x + (((((long)(b % Bx) * D + dd) * H + hh) * W + ww) * C8 + c8) * VEC, xv);
#pragma unroll
        for (int j = 0; j < VEC; ++j) xv[j] *= s;
        VecIO<T, VEC>::store(y + i * VEC, xv);
    }
}

// dx[bx][v][c] = sum_p dy[p Bx + bx][v][c] * w[p Bx + bx][...]   (w carries no gradient: the attention map is detached)
template <typename T, int VEC>
__global__ __launch_bounds__(256) void scale_nearest_bwd_kernel(const T* __restrict__ dy, const float* __restrict__ w, T* __restrict__ dx,
                                                                int Bv, int Bx, int D, int H, int W, int C8, int k) {
    const long total = (long)Bx * D * H * W * C8;
    const int d = D >> k, h = H >> k, ws = W >> k;
    const int P = Bv / Bx;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        long v = i / C8;
        const int c8 = (int)(i % C8);
        const int ww = (int)(v % W); v /= W;
        const int hh = (int)(v % H); v /= H;
        const int dd = (int)(v % D);
        const int bx = (int)(v / D);
        float acc[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
        for (int p = 0; p < P; ++p) {
            const int b = p * Bx + bx;
            const float s = w[(((long)b * d + (dd >> k)) * h + (hh >> k)) * ws + (ww >> k)];
            float g[VEC];
            VecIO<T, VEC>::load(dy + (((((long)b * D + dd) * H + hh) * W + ww) * C8 + c8) * VEC, g);
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[j] = __builtin_fmaf(g[j], s, acc[j]);
        }
        VecIO<T, VEC>::store(dx + i * VEC, acc);
    }
}

unsigned grid_for(long total) { return (unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256); }

}  // namespace

extern "C" int passion_gconv3d_fwd(const void* x, const float* weight, const float* bias, void* y, int dtype, int B, int D, int H, int W,
                                   int C, int cpg, void* stream) {
    if (B <= 0 || C <= 0 || cpg <= 0 || C % cpg) return -1003;
    const long total = (long)B * D * H * W * C;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PASSION_F32)
        hipLaunchKernelGGL(gconv_fwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st, (const float*)x, weight, bias, (float*)y, B, D, H, W, C, cpg);
    else if (dtype == PASSION_BF16)
        hipLaunchKernelGGL(gconv_fwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, st, (const bf16_t*)x, weight, bias, (bf16_t*)y, B, D, H, W, C, cpg);
    else return -1007;
    LAUNCH_RET();
    return 0;
}

extern "C" int passion_gconv3d_bwd(const void* x, const float* weight, const void* dy, void* dx, float* dweight, float* dbias, int dtype,
                                   int B, int D, int H, int W, int C, int cpg, void* stream) {
    if (B <= 0 || C <= 0 || cpg <= 0 || C % cpg) return -1003;
    const long total = (long)B * D * H * W * C;
    const long nout = (long)C * cpg * 27;
    hipStream_t st = (hipStream_t)stream;
    if (dtype != PASSION_F32 && dtype != PASSION_BF16) return -1007;
    if (dx) {
        if (dtype == PASSION_F32)
            hipLaunchKernelGGL(gconv_dgrad_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st, (const float*)dy, weight, (float*)dx, B, D, H, W, C, cpg);
        else
            hipLaunchKernelGGL(gconv_dgrad_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, st, (const bf16_t*)dy, weight, (bf16_t*)dx, B, D, H, W, C, cpg);
        LAUNCH_RET();
    }
    if (dweight) {
        if (dtype == PASSION_F32)
            hipLaunchKernelGGL(gconv_wgrad_kernel<float>, dim3((unsigned)((nout + 3) / 4)), dim3(256), 0, st, (const float*)x, (const float*)dy, dweight, dbias, B, D, H, W, C, cpg);
        else
            hipLaunchKernelGGL(gconv_wgrad_kernel<bf16_t>, dim3((unsigned)((nout + 3) / 4)), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, dweight, dbias, B, D, H, W, C, cpg);
        LAUNCH_RET();
    }
    return 0;
}

#define SCALE_LAUNCH(KERN, TT, VV, SRC, DST, BT)                                                                                          \
    hipLaunchKernelGGL((KERN<TT, VV>), dim3(grid_for((long)(BT) * D * H * W * (C / VV))), dim3(256), 0, st, (const TT*)(SRC), w, (TT*)(DST), Bv, Bx, \
                       D, H, W, C / VV, k)

extern "C" int passion_scale_nearest_fwd(const void* x, const float* w, void* y, int dtype, int Bv, int Bx, int D, int H, int W, int C, int k,
                                         void* stream) {
    if (Bv <= 0 || Bx <= 0 || Bv % Bx || C % 4 || k < 0 || ((D | H | W) & ((1 << k) - 1))) return -1003;
    if (dtype != PASSION_F32 && dtype != PASSION_BF16) return -1007;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PASSION_F32) SCALE_LAUNCH(scale_nearest_fwd_kernel, float, 4, x, y, Bv);
    else { if (C % 8 == 0) SCALE_LAUNCH(scale_nearest_fwd_kernel, bf16_t, 8, x, y, Bv); else SCALE_LAUNCH(scale_nearest_fwd_kernel, bf16_t, 4, x, y, Bv); }
    LAUNCH_RET();
    return 0;
}

extern "C" int passion_scale_nearest_bwd(const void* dy, const float* w, void* dx, int dtype, int Bv, int Bx, int D, int H, int W, int C, int k,
                                         void* stream) {
    if (Bv <= 0 || Bx <= 0 || Bv % Bx || C % 4 || k < 0 || ((D | H | W) & ((1 << k) - 1))) return -1003;
    if (dtype != PASSION_F32 && dtype != PASSION_BF16) return -1007;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PASSION_F32) SCALE_LAUNCH(scale_nearest_bwd_kernel, float, 4, dy, dx, Bx);
    else { if (C % 8 == 0) SCALE_LAUNCH(scale_nearest_bwd_kernel, bf16_t, 8, dy, dx, Bx); else SCALE_LAUNCH(scale_nearest_bwd_kernel, bf16_t, 4, dy, dx, Bx); }
    LAUNCH_RET();
    return 0;
}
