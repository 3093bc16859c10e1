// RFNet-specific streaming kernels (reference code/models/rfnet.py, code/models/blocks.py:354-370, 495-531, 582-626).
//
//  norm_act      : materialise y = sum_k LeakyReLU(InstanceNorm(x_k)) for 1 or 2 inputs.  RFNet is post-norm
//                  (general_conv3d = Conv -> IN -> LReLU, blocks.py:366-370): a conv still consumes the raw output of
//                  its producer and normalises it in its loader, but the encoder residual x + c3(c2(x))
//                  (rfnet.py:37), the upsampling inputs and the prototype features need the activated values.
//  region_gap    : S[b][i][m][c] = sum_v y_m[c](v) * prob_i(v),  P[b][i] = sum_v prob_i(v)  -- the global average
//                  pools of modal_fusion (blocks.py:507-508) for all 4 regions x 4 modalities in one pass.
//  region_mix    : R[b][v][i*C + c] = prob_i(v) * sum_m w[b][i][m] * y_m[c](v)  -- modality features split into
//                  regions (blocks.py:602-610), weighted modality sum (blocks.py:516) and the reshape feeding
//                  region_fusion (blocks.py:530), without the (B,4,4,C,D,H,W) intermediate of the reference.
// Absent modalities (mask) read as zeros.  All are HBM-bound: 16-byte loads, one pass, reductions through
// wavefront shuffles + per-block partial slabs folded in a fixed order (no atomics).
#include "common.h"
#include "passion_hip.h"

namespace {

struct Mod4 {
    const void* y[4];
    void* dy[4];
    const uint8_t* mask;   // device [B][mask_ld] modality-present flags (NULL: all present)
    int mask_ld;
    int bmod;              // > 0: y holds bmod samples, virtual sample b reads y[b % bmod] (several masked passes in one launch)
    __device__ __forceinline__ int ysrc(int b) const { return bmod > 0 ? b % bmod : b; }
    __device__ __forceinline__ bool present(int m, int b) const { return mask ? mask[(long)b * mask_ld + m] != 0 : true; }
};

// ------------------------------------------------------------------------------------------------ norm_act
template <typename T>
__global__ void norm_act_fwd_kernel(const T* __restrict__ x0, const float* __restrict__ st0, const T* __restrict__ x1,
                                    const float* __restrict__ st1, T* __restrict__ out, long V, int C, int out_ld,
                                    int out_coff, float slope) {
    const int b = blockIdx.y;
    const int cvn = C / 4;
    const long nvec = V * cvn;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cvn) * 4;
        const long v = i / cvn;
        f32x4 a = load4(x0 + ((long)b * V + v) * C + c), r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float t = (a[j] - st0[((long)b * C + c + j) * 2]) * st0[((long)b * C + c + j) * 2 + 1];
            r[j] = t > 0.f ? t : t * slope;
        }
        if (x1) {
            a = load4(x1 + ((long)b * V + v) * C + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = (a[j] - st1[((long)b * C + c + j) * 2]) * st1[((long)b * C + c + j) * 2 + 1];
                r[j] += t > 0.f ? t : t * slope;
            }
        }
        store4(out + ((long)b * V + v) * out_ld + out_coff + c, r);
    }
}

// backward, pass 1: per (b, c) sums of g = dOut * lrelu'(xhat) and g * xhat  -> part[blk][b][c][2]
template <typename T>
__global__ __launch_bounds__(256) void norm_act_bwd_sums_kernel(const T* __restrict__ x, const float* __restrict__ st,
                                                                const T* __restrict__ dout, long V, int C, int out_ld,
                                                                int out_coff, float slope, float* __restrict__ part) {
    __shared__ float red[256 * 8];
    const int b = blockIdx.y, cvn = C / 4, tid = threadIdx.x;
    const long nvec = V * cvn;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    for (long i = (long)blockIdx.x * 256 + tid; i < nvec; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cvn) * 4;
        const long v = i / cvn;
        const f32x4 a = load4(x + ((long)b * V + v) * C + c), g = load4(dout + ((long)b * V + v) * out_ld + out_coff + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xh = (a[j] - st[((long)b * C + c + j) * 2]) * st[((long)b * C + c + j) * 2 + 1];
            const float gg = g[j] * (xh > 0.f ? 1.f : slope);
            s1[j] += gg; s2[j] += gg * xh;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[tid * 8 + j] = s1[j]; red[tid * 8 + 4 + j] = s2[j]; }
    __syncthreads();
    if (tid < cvn) {   // threads tid, tid + cvn, ... share the channel vector (cvn divides 256)
        float a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
        for (int t = tid; t < 256; t += cvn)
#pragma unroll
            for (int j = 0; j < 4; ++j) { a1[j] += red[t * 8 + j]; a2[j] += red[t * 8 + 4 + j]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            part[(((long)blockIdx.x * gridDim.y + b) * C + tid * 4 + j) * 2 + 0] = a1[j];
            part[(((long)blockIdx.x * gridDim.y + b) * C + tid * 4 + j) * 2 + 1] = a2[j];
        }
    }
}

// out[i] = sum_r part[r*n + i]  (16 outputs x 16 slices per block, fixed order)
__global__ __launch_bounds__(256) void fold_kernel(const float* __restrict__ part, int R, long n, float* __restrict__ out) {
    __shared__ float red[16][17];
    const int li = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const long i = (long)blockIdx.x * 16 + li;
    float a = 0.f;
    if (i < n)
        for (int r = sl; r < R; r += 16) a += part[(long)r * n + i];
    red[sl][li] = a;
    __syncthreads();
    if (sl == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][li];
        out[i] = t;
    }
}

// backward, pass 2: dx = rstd * (g - mean(g) - xhat * mean(g xhat))
template <typename T>
__global__ void norm_act_bwd_apply_kernel(const T* __restrict__ x, const float* __restrict__ st, const T* __restrict__ dout,
                                          const float* __restrict__ sums, T* __restrict__ dx, long V, int C, int out_ld,
                                          int out_coff, float slope) {
    const int b = blockIdx.y, cvn = C / 4;
    const long nvec = V * cvn;
    const float invV = 1.f / (float)V;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cvn) * 4;
        const long v = i / cvn;
        const f32x4 a = load4(x + ((long)b * V + v) * C + c), g = load4(dout + ((long)b * V + v) * out_ld + out_coff + c);
        f32x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float rstd = st[((long)b * C + c + j) * 2 + 1];
            const float xh = (a[j] - st[((long)b * C + c + j) * 2]) * rstd;
            const float gg = g[j] * (xh > 0.f ? 1.f : slope);
            r[j] = rstd * (gg - sums[((long)b * C + c + j) * 2] * invV - xh * sums[((long)b * C + c + j) * 2 + 1] * invV);
        }
        store4(dx + ((long)b * V + v) * C + c, r);
    }
}

// ------------------------------------------------------------------------------------------------ region_gap
// part[blk][b][(i*4+m)*C + c] and part_p[blk][b][i]
template <typename T>
__global__ __launch_bounds__(256) void region_gap_kernel(Mod4 md, const float* __restrict__ prob, long V, int C,
                                                         float* __restrict__ part) {
    __shared__ float red[256 * 4];
    const int b = blockIdx.y, cvn = C / 4, tid = threadIdx.x;
    const long nvec = V * cvn;
    const int nout = 16 * C + 4;
    float acc[4][4][4];   // [region i][modality m][channel lane]
    float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][m][j] = 0.f;
    for (long it = (long)blockIdx.x * 256 + tid; it < nvec; it += (long)gridDim.x * 256) {
        const int c = (int)(it % cvn) * 4;
        const long v = it / cvn;
        const f32x4 pr = load4(prob + ((long)b * V + v) * 4);
        if (c == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) ps[i] += pr[i];
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (!md.present(m, b)) continue;
            const f32x4 y = load4(reinterpret_cast<const T*>(md.y[m]) + ((long)md.ysrc(b) * V + v) * C + c);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][m][j] += y[j] * pr[i];
        }
    }
    float* out = part + ((long)blockIdx.x * gridDim.y + b) * nout;
    // reduce each of the 64 accumulators over the threads that share this channel vector (tid % cvn)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j) red[tid * 4 + j] = acc[i][m][j];
            __syncthreads();
            if (tid < cvn) {
                float a[4] = {0, 0, 0, 0};
                for (int t = tid; t < 256; t += cvn)
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[j] += red[t * 4 + j];
#pragma unroll
                for (int j = 0; j < 4; ++j) out[(i * 4 + m) * C + tid * 4 + j] = a[j];
            }
        }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) red[tid * 4 + i] = ps[i];
    __syncthreads();
    if (tid < 4) {
        float a = 0.f;
        for (int t = 0; t < 256; ++t) a += red[t * 4 + tid];
        out[16 * C + tid] = a;
    }
}

// dy_m[c](v) = sum_i gS[b][i][m][c] * prob_i(v)   (prob is detached in the reference, blocks.py / rfnet.py:128)
template <typename T>
__global__ void region_gap_bwd_kernel(Mod4 md, const float* __restrict__ prob, const float* __restrict__ gS, long V, int C, int P, int Bphys) {
    // P masked passes over the same Bphys samples (virtual sample b = p * Bphys + bp): their gradients wrt y add up, summed here in
    // f32 and written once (the per-pass tensors + a torch sum were 2 x P tensor passes per modality and level)
    const int bp = blockIdx.y, cvn = C / 4;
    const long nvec = V * cvn;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < nvec; it += (long)gridDim.x * blockDim.x) {
        const int c = (int)(it % cvn) * 4;
        const long v = it / cvn;
        f32x4 r[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) r[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int p = 0; p < P; ++p) {
            const int b = p * Bphys + bp;
            const f32x4 pr = load4(prob + ((long)b * V + v) * 4);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (!md.dy[m] || !md.present(m, b)) continue;
#pragma unroll
                for (int i = 0; i < 4; ++i) r[m] += pr[i] * load4(gS + (((long)b * 4 + i) * 4 + m) * C + c);
            }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
            if (md.dy[m]) store4(reinterpret_cast<T*>(md.dy[m]) + ((long)bp * V + v) * C + c, r[m]);
    }
}

// ------------------------------------------------------------------------------------------------ region_mix
template <typename T>
__global__ void region_mix_fwd_kernel(Mod4 md, const float* __restrict__ prob, const float* __restrict__ w, T* __restrict__ out,
                                      long V, int C) {
    const int b = blockIdx.y, cvn = C / 4;
    const long nvec = V * cvn;
    float wl[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int m = 0; m < 4; ++m) wl[i][m] = md.present(m, b) ? w[((long)b * 4 + i) * 4 + m] : 0.f;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < nvec; it += (long)gridDim.x * blockDim.x) {
        const int c = (int)(it % cvn) * 4;
        const long v = it / cvn;
        const f32x4 pr = load4(prob + ((long)b * V + v) * 4);
        f32x4 y[4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
            y[m] = md.present(m, b) ? load4(reinterpret_cast<const T*>(md.y[m]) + ((long)md.ysrc(b) * V + v) * C + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 r = pr[i] * (wl[i][0] * y[0] + wl[i][1] * y[1] + wl[i][2] * y[2] + wl[i][3] * y[3]);
            store4(out + ((long)b * V + v) * 4 * C + i * C + c, r);
        }
    }
}

// dy_m[c] = sum_i w[i][m] prob_i dR[i*C+c];  dw[b][i][m] = sum_{v,c} dR[i*C+c] prob_i y_m[c]  -> part[blk][b][16]
// PP >= P masked passes over the same Bphys samples in one launch (virtual sample b = p * Bphys + bp, all passes read y[bp]): a block
// row owns a PHYSICAL sample, y is loaded once per voxel, dy is the f32 sum over the passes written once, dw stays per pass.
template <typename T, int PP>
__global__ __launch_bounds__(256) void region_mix_bwd_kernel(Mod4 md, const float* __restrict__ prob, const float* __restrict__ w,
                           const T* __restrict__ dR, long V, int C, float* __restrict__ part, int P, int Bphys) {
    __shared__ float red[4][16];
    __shared__ __attribute__((aligned(16))) float wls[PP][16];          // [pass][i * 4 + m], zero for absent modalities
    const int bp = blockIdx.y, cvn = C / 4, tid = threadIdx.x;
    const int Bv = P * Bphys;
    const long nvec = V * cvn;
    if (tid < PP * 16) {
        const int p = tid / 16, im = tid % 16, b = p * Bphys + bp;
        wls[p][im] = (p < P && md.present(im % 4, b)) ? w[(long)b * 16 + im] : 0.f;
    }
    unsigned pres = 0u;                        // bit p * 4 + m: modality m present in pass p
#pragma unroll
    for (int p = 0; p < PP; ++p)
#pragma unroll
        for (int m = 0; m < 4; ++m)
            if (p < P && md.present(m, p * Bphys + bp)) pres |= 1u << (p * 4 + m);
    float dw[PP][16];
#pragma unroll
    for (int p = 0; p < PP; ++p)
#pragma unroll
        for (int k = 0; k < 16; ++k) dw[p][k] = 0.f;
    __syncthreads();
    for (long it = (long)blockIdx.x * 256 + tid; it < nvec; it += (long)gridDim.x * 256) {
        const int c = (int)(it % cvn) * 4;
        const long v = it / cvn;
        f32x4 y[4], r[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            y[m] = load4(reinterpret_cast<const T*>(md.y[m]) + ((long)bp * V + v) * C + c);
            r[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int p = 0; p < PP; ++p) {
            if (p >= P) continue;
            // one pass of loads at a time: hoisting all PP passes' loads cost 256 registers + 37 spilled dwords (423 us per launch
            // against 267 for the per-pass kernel this replaces)
            if (PP > 1) asm volatile("" ::: "memory");
            const long b = (long)p * Bphys + bp;
            const f32x4 pr = load4(prob + (b * V + v) * 4);
            f32x4 g[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) g[i] = pr[i] * load4(dR + (b * V + v) * 4 * C + i * C + c);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (!((pres >> (p * 4 + m)) & 1u)) continue;               // block-uniform
#pragma unroll
                for (int i = 0; i < 4; ++i) dw[p][i * 4 + m] += g[i][0] * y[m][0] + g[i][1] * y[m][1] + g[i][2] * y[m][2] + g[i][3] * y[m][3];
                r[m] += wls[p][m] * g[0] + wls[p][4 + m] * g[1] + wls[p][8 + m] * g[2] + wls[p][12 + m] * g[3];
            }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
            if (md.dy[m]) store4(reinterpret_cast<T*>(md.dy[m]) + ((long)bp * V + v) * C + c, r[m]);
    }
    const int wv = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int p = 0; p < PP; ++p) {
        if (p >= P) continue;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float s_ = wave_sum(dw[p][k]);
            if (lane == 0) red[wv][k] = s_;
        }
        __syncthreads();
        if (tid < 16) part[((long)blockIdx.x * Bv + (p * Bphys + bp)) * 16 + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
        __syncthreads();
    }
}

inline int egrid(long n) { return (int)hmin((n + 255) / 256, 4096); }
inline int rgrid(long nvec) { long g = (nvec + 256 * 16 - 1) / (256 * 16); return (int)(g < 1 ? 1 : (g > 256 ? 256 : g)); }

int fill_mod(Mod4& md, const void* const* y, void* const* dy, const uint8_t* mask, int mask_ld, int bmod = 0) {
    for (int m = 0; m < 4; ++m) {
        md.y[m] = y ? y[m] : nullptr;
        md.dy[m] = dy ? dy[m] : nullptr;
    }
    md.mask = mask; md.mask_ld = mask_ld; md.bmod = bmod;
    return 0;
}

}  // namespace

#define DISPATCH_T(dtype, CALLF, CALLB)                 \
    if ((dtype) == PASSION_F32) { CALLF; }              \
    else if ((dtype) == PASSION_BF16) { CALLB; }        \
    else return -1007;

extern "C" int passion_norm_act_fwd(const void* x0, const float* st0, const void* x1, const float* st1, void* out, int dtype,
                                    int B, int64_t V, int C, int out_ld, int out_coff, float slope, void* stream) {
    if (C % 4 || out_ld % 4 || out_coff % 4) return -1021;
    dim3 grid(egrid(V * (C / 4)), B);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(norm_act_fwd_kernel<float>, grid, dim3(256), 0, st, (const float*)x0, st0, (const float*)x1, st1, (float*)out, (long)V, C, out_ld, out_coff, slope),
               hipLaunchKernelGGL(norm_act_fwd_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x0, st0, (const bf16_t*)x1, st1, (bf16_t*)out, (long)V, C, out_ld, out_coff, slope));
    LAUNCH_RET();
    return 0;
}

extern "C" int64_t passion_norm_act_scratch_floats(int B, int64_t V, int C) {
    return (int64_t)(rgrid((long)V * (C / 4)) + 1) * B * C * 2;
}

// gradient of passion_norm_act_fwd wrt ONE of its inputs (call once per input)
extern "C" int passion_norm_act_bwd(const void* x, const float* stats, const void* dout, void* dx, int dtype, int B, int64_t V,
                                    int C, int out_ld, int out_coff, float slope, float* scratch, void* stream) {
    if (C % 4 || (256 % (C / 4)) != 0) return -1021;
    hipStream_t st = (hipStream_t)stream;
    const int gx = rgrid((long)V * (C / 4));
    const long n = (long)B * C * 2;
    float* sums = scratch + (long)gx * n;
    dim3 rg(gx, B), eg(egrid(V * (C / 4)), B);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(norm_act_bwd_sums_kernel<float>, rg, dim3(256), 0, st, (const float*)x, stats, (const float*)dout, (long)V, C, out_ld, out_coff, slope, scratch),
               hipLaunchKernelGGL(norm_act_bwd_sums_kernel<bf16_t>, rg, dim3(256), 0, st, (const bf16_t*)x, stats, (const bf16_t*)dout, (long)V, C, out_ld, out_coff, slope, scratch));
    LAUNCH_RET();
    hipLaunchKernelGGL(fold_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st, scratch, gx, n, sums);
    LAUNCH_RET();
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(norm_act_bwd_apply_kernel<float>, eg, dim3(256), 0, st, (const float*)x, stats, (const float*)dout, sums, (float*)dx, (long)V, C, out_ld, out_coff, slope),
               hipLaunchKernelGGL(norm_act_bwd_apply_kernel<bf16_t>, eg, dim3(256), 0, st, (const bf16_t*)x, stats, (const bf16_t*)dout, sums, (bf16_t*)dx, (long)V, C, out_ld, out_coff, slope));
    LAUNCH_RET();
    return 0;
}

extern "C" int64_t passion_region_scratch_floats(int B, int64_t V, int C) {
    return (int64_t)rgrid((long)V * (C / 4)) * B * (16 * C + 4);
}

// sums: [B][16*C + 4] = S[b][i][m][c] then P[b][i]
extern "C" int passion_region_gap_fwd(const void* const* y, const uint8_t* mask, int mask_ld, const float* prob, int dtype, int B,
                                      int64_t V, int C, int bmod, float* scratch, float* sums, void* stream) {
    if (C % 4 || (256 % (C / 4)) != 0) return -1021;
    Mod4 md;
    int rc = fill_mod(md, y, nullptr, mask, mask_ld, bmod);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int gx = rgrid((long)V * (C / 4));
    dim3 rg(gx, B);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(region_gap_kernel<float>, rg, dim3(256), 0, st, md, prob, (long)V, C, scratch),
               hipLaunchKernelGGL(region_gap_kernel<bf16_t>, rg, dim3(256), 0, st, md, prob, (long)V, C, scratch));
    LAUNCH_RET();
    const long n = (long)B * (16 * C + 4);
    hipLaunchKernelGGL(fold_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st, scratch, gx, n, sums);
    LAUNCH_RET();
    return 0;
}

// gS: [B][4][4][C] gradient wrt S; writes dy[m] (skipped when NULL)
// bmod > 0: B = P * bmod virtual samples (P masked passes over bmod physical ones, virtual b = p * bmod + bp); dy[m] then holds bmod
// samples: the SUM of the passes' gradients
extern "C" int passion_region_gap_bwd(void* const* dy, const uint8_t* mask, int mask_ld, const float* prob, const float* gS, int dtype,
                                      int B, int64_t V, int C, int bmod, void* stream) {
    if (C % 4) return -1021;
    if (bmod < 0 || (bmod > 0 && B % bmod)) return -1022;
    Mod4 md;
    int rc = fill_mod(md, nullptr, dy, mask, mask_ld);
    if (rc) return rc;
    const int Bphys = bmod > 0 ? bmod : B, P = B / Bphys;
    dim3 grid(egrid(V * (C / 4)), Bphys);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(region_gap_bwd_kernel<float>, grid, dim3(256), 0, st, md, prob, gS, (long)V, C, P, Bphys),
               hipLaunchKernelGGL(region_gap_bwd_kernel<bf16_t>, grid, dim3(256), 0, st, md, prob, gS, (long)V, C, P, Bphys));
    LAUNCH_RET();
    return 0;
}

// w: [B][4 regions][4 modalities]; out: [B][V][4*C]
extern "C" int passion_region_mix_fwd(const void* const* y, const uint8_t* mask, int mask_ld, const float* prob, const float* w, void* out,
                                      int dtype, int B, int64_t V, int C, int bmod, void* stream) {
    if (C % 4) return -1021;
    Mod4 md;
    int rc = fill_mod(md, y, nullptr, mask, mask_ld, bmod);
    if (rc) return rc;
    dim3 grid(egrid(V * (C / 4)), B);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(region_mix_fwd_kernel<float>, grid, dim3(256), 0, st, md, prob, w, (float*)out, (long)V, C),
               hipLaunchKernelGGL(region_mix_fwd_kernel<bf16_t>, grid, dim3(256), 0, st, md, prob, w, (bf16_t*)out, (long)V, C));
    LAUNCH_RET();
    return 0;
}

// dR: [B][V][4*C]; writes dy[m] (skipped when NULL) and dw [B][4][4]; scratch: passion_region_scratch_floats.
// bmod > 0 (B = P * bmod virtual samples, P <= 8 passes over bmod physical ones): dy[m] holds bmod samples, the SUM over the passes
extern "C" int passion_region_mix_bwd(const void* const* y, void* const* dy, const uint8_t* mask, int mask_ld, const float* prob,
                                      const float* w, const void* dR, int dtype, int B, int64_t V, int C, int bmod, float* scratch,
                                      float* dw, void* stream) {
    if (C % 4) return -1021;
    Mod4 md;
    int rc = fill_mod(md, y, dy, mask, mask_ld, bmod);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (bmod < 0 || (bmod > 0 && B % bmod)) return -1022;
    const int Bphys = bmod > 0 ? bmod : B, P = B / Bphys;
    if (P > 8) return -1022;
    const int gx = rgrid((long)V * (C / 4));
    dim3 rg(gx, Bphys);
#define MIX_BWD(PPV)                                                                                                                         \
    DISPATCH_T(dtype,                                                                                                                        \
               hipLaunchKernelGGL((region_mix_bwd_kernel<float, PPV>), rg, dim3(256), 0, st, md, prob, w, (const float*)dR, (long)V, C, scratch, P, Bphys),  \
               hipLaunchKernelGGL((region_mix_bwd_kernel<bf16_t, PPV>), rg, dim3(256), 0, st, md, prob, w, (const bf16_t*)dR, (long)V, C, scratch, P, Bphys))
    if (P == 1) { MIX_BWD(1); }
    else if (P <= 3) { MIX_BWD(3); }
    else if (P <= 5) { MIX_BWD(5); }
    else { MIX_BWD(8); }
#undef MIX_BWD
    LAUNCH_RET();
    const long n = (long)B * 16;
    hipLaunchKernelGGL(fold_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st, scratch, gx, n, dw);
    LAUNCH_RET();
    return 0;
}
