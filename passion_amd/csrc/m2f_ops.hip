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

// ---- channel-vectorised forms (round 3): the scalar kernels above spend a 2-byte load per multiply-add and, in the weight
// gradient, one wavefront per weight striding through the volume (6.7 ms of a 50 ms m2ftrans step).  Here a thread owns the
// 8 consecutive channels of one voxel (one 16-byte access per tap), a block owns a range of CB <= 64 channels whose weights
// it stages in LDS as tab[tap][i][c], and `wsets` weight sets share a launch: sample b uses set b / (B / wsets) -- the four
// key_map_* and four value_map_* blocks of MultiMaskAttentionLayer (blocks.py:246-297) run as ONE batch of 8 B samples.
// Forward and input gradient are the same kernel over differently staged tables (gather form, taps mirrored).
template <typename T> __device__ __forceinline__ void gld8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void gld8<bf16_t>(const bf16_t* p, float (&v)[8]) { VecIO<bf16_t, 8>::load(p, v); }
template <> __device__ __forceinline__ void gld8<float>(const float* p, float (&v)[8]) {
    const f32x4 a = load4(p), b = load4(p + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
}
template <typename T> __device__ __forceinline__ void gst8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void gst8<bf16_t>(bf16_t* p, const float (&v)[8]) { VecIO<bf16_t, 8>::store(p, v); }
template <> __device__ __forceinline__ void gst8<float>(float* p, const float (&v)[8]) {
    store4(p, (f32x4){v[0], v[1], v[2], v[3]});
    store4(p + 4, (f32x4){v[4], v[5], v[6], v[7]});
}

// grid (voxel blocks, C / CB, B).  DGRAD = false: y = bias + sum_t sum_i w[co][i][t] x[v + t - 1][g cpg + i];
// DGRAD = true: dx = sum_t sum_o w[g cpg + o][ci - g cpg][26 - t] dy[v + t - 1][g cpg + o]
template <typename T, int CPG, bool DGRAD>
__global__ __launch_bounds__(256) void gconv_vec_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                        T* __restrict__ y, int B, int D, int H, int W, int C, int CB, int wsets) {
    extern __shared__ float tab[];                        // [27][CPG][CB]
    const int b = blockIdx.z, c0 = blockIdx.y * CB;
    const int set = b / (B / wsets);
    const float* ws = w + (long)set * C * CPG * 27;
    for (int i = threadIdx.x; i < 27 * CPG * CB; i += 256) {
        const int c = i % CB, k = (i / CB) % CPG, t = i / (CB * CPG);
        const int ch = c0 + c, gb = (ch / CPG) * CPG;
        tab[i] = DGRAD ? ws[((long)(gb + k) * CPG + (ch - gb)) * 27 + (26 - t)] : ws[((long)ch * CPG + k) * 27 + t];
    }
    __syncthreads();
    const int nsl = CB >> 3, vpb = 256 / nsl;
    const int sl = threadIdx.x % nsl, vi = threadIdx.x / nsl;
    const long V = (long)D * H * W;
    const long v = (long)blockIdx.x * vpb + vi;
    if (v >= V) return;
    const int ww = (int)(v % W), hh = (int)((v / W) % H), dd = (int)(v / ((long)W * H));
    const int cl = sl * 8, ch = c0 + cl;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = (!DGRAD && bias) ? bias[(long)set * C + ch + j] : 0.f;
    const T* xb = x + (long)b * V * C + ch;
    for (int td = 0; td < 3; ++td) {
        const int zd = dd + td - 1;
        if (zd < 0 || zd >= D) continue;
        for (int th = 0; th < 3; ++th) {
            const int zh = hh + th - 1;
            if (zh < 0 || zh >= H) continue;
#pragma unroll
            for (int tw = 0; tw < 3; ++tw) {
                const int zw = ww + tw - 1;
                if (zw < 0 || zw >= W) continue;
                float xv[8];
                gld8<T>(xb + (((long)zd * H + zh) * W + zw) * C, xv);
                const float* tp = tab + (td * 9 + th * 3 + tw) * CPG * CB + cl;
#pragma unroll
                for (int k = 0; k < CPG; ++k) {
                    const f32x4 wa = *reinterpret_cast<const f32x4*>(tp + k * CB), wb = *reinterpret_cast<const f32x4*>(tp + k * CB + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[j] = __builtin_fmaf(wa[j], CPG == 1 ? xv[j] : xv[(j / CPG) * CPG + k], acc[j]);
                        acc[4 + j] = __builtin_fmaf(wb[j], CPG == 1 ? xv[4 + j] : xv[((4 + j) / CPG) * CPG + k], acc[4 + j]);
                    }
                }
            }
        }
    }
    gst8<T>(y + ((long)b * V + v) * C + ch, acc);
}

// weight gradient, partial over a chunk of voxels: grid (chunks, C / CB, wsets); thread = (tap lane 0..31: 27 taps, lane 27 = bias,
// rest idle) x (8-channel slice): part[chunk][set][c][i][28] (column 27 of i = 0: the bias gradient)
template <typename T, int CPG>
__global__ __launch_bounds__(256) void gconv_wgrad_vec_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ part,
                                                              int B, int D, int H, int W, int C, int CB, int wsets, int chunk) {
    const int set = blockIdx.z, c0 = blockIdx.y * CB;
    const int nsl = CB >> 3, nvl = 256 / (32 * nsl);          // voxel lanes: threads beyond the 32 x nsl (tap, slice) pairs split the chunk
    const int tl = threadIdx.x & 31, sl = (threadIdx.x >> 5) % nsl, vl = (threadIdx.x >> 5) / nsl;
    const int bper = B / wsets;
    const long V = (long)D * H * W, Q = (long)bper * V;        // voxels of this weight set
    const long lo = (long)blockIdx.x * chunk, hi = lo + chunk < Q ? lo + chunk : Q;
    const int td = tl / 9 - 1, th = (tl / 3) % 3 - 1, tw = tl % 3 - 1;
    const int ch = c0 + sl * 8;
    float acc[8][CPG];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int k = 0; k < CPG; ++k) acc[j][k] = 0.f;
    if (tl < 28) {
        // four voxels per trip, every load issued before the first use (clamped addresses, zero weights for what does not exist):
        // a trip costs one memory latency, not eight
        for (long q0 = lo + vl; q0 < hi; q0 += 4L * nvl) {
            float g[4][8], xv[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long q = q0 + (long)u * nvl;
                const bool live = q < hi;
                const long qc = live ? q : lo;
                const long bq = (long)set * bper + qc / V, v = qc % V;
                gld8<T>(dy + (bq * V + v) * C + ch, g[u]);
                const int ww = (int)(v % W), hh = (int)((v / W) % H), dd = (int)(v / ((long)W * H));
                const int zd = dd + td, zh = hh + th, zw = ww + tw;
                const bool in = tl < 27 && zd >= 0 && zd < D && zh >= 0 && zh < H && zw >= 0 && zw < W;
                const long vx = in ? ((long)zd * H + zh) * W + zw : v;
                gld8<T>(x + (bq * V + vx) * C + ch, xv[u]);
                const float keep = live ? 1.f : 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) { g[u][j] *= keep; if (!in) xv[u][j] = 0.f; }
            }
            if (tl == 27) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j][0] += g[u][j];
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < 8; ++j)
#pragma unroll
                        for (int k = 0; k < CPG; ++k)
                            acc[j][k] = __builtin_fmaf(g[u][j], CPG == 1 ? xv[u][j] : xv[u][(j / CPG) * CPG + k], acc[j][k]);
            }
        }
    }
    // the nvl voxel lanes of a (tap, slice) pair, in lane order
    extern __shared__ float red[];                         // [nvl][nsl][32][8 * CPG]
    float* mine = red + (((long)vl * nsl + sl) * 32 + tl) * (8 * CPG);
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int k = 0; k < CPG; ++k) mine[j * CPG + k] = acc[j][k];
    __syncthreads();
    if (vl == 0 && tl < 28) {
        float* out = part + (((long)blockIdx.x * wsets + set) * C + ch) * CPG * 28;
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int k = 0; k < CPG; ++k) {
                float t = 0.f;
                for (int u = 0; u < nvl; ++u) t += red[(((long)u * nsl + sl) * 32 + tl) * (8 * CPG) + j * CPG + k];
                if (tl < 27 || k == 0) out[((long)j * CPG + k) * 28 + tl] = t;
            }
    }
}

// dw[set][c][i][t] = sum_chunks part[..][t], db[set][c] = sum_chunks part[..][i = 0][27]
__global__ __launch_bounds__(256) void gconv_wfold_kernel(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ db,
                                                          int nchunk, long n_ci, int cpg) {         // n_ci = wsets * C * cpg
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_ci * 28) return;
    const long ci = i / 28;
    const int t = (int)(i % 28);
    float s = 0.f;
    for (int k = 0; k < nchunk; ++k) s += part[((long)k * n_ci + ci) * 28 + t];
    if (t < 27) dw[ci * 27 + t] = s;
    else if (db && ci % cpg == 0) db[ci / cpg] = s;          // the bias column rides with i == 0
}

int gconv_cb(int C) {                                        // channel range of a block, 0 = the scalar kernels
    if (C % 64 == 0) return 64;
    if (C == 8 || C == 16 || C == 32) return C;
    return 0;
}
constexpr int GC_CHUNK = 32;                                // voxels per weight-gradient chunk

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

extern "C" int64_t passion_gconv3d_part_floats(int B, int D, int H, int W, int C, int cpg, int wsets) {
    if (B <= 0 || wsets <= 0 || B % wsets || !gconv_cb(C) || (cpg != 1 && cpg != 4)) return 0;
    const long Q = (long)(B / wsets) * D * H * W;
    return ((Q + GC_CHUNK - 1) / GC_CHUNK) * (long)wsets * C * cpg * 28;
}

#define GCONV_VEC(TT, CPGV, DG, SRC, DST, BIAS)                                                                                              \
    hipLaunchKernelGGL((gconv_vec_kernel<TT, CPGV, DG>), dim3((unsigned)((V + vpb - 1) / vpb), C / CB, B), dim3(256), 27 * CPGV * CB * 4, st, \
                       (const TT*)(SRC), weight, BIAS, (TT*)(DST), B, D, H, W, C, CB, wsets)

extern "C" int passion_gconv3d_fwd(const void* x, const float* weight, const float* bias, void* y, int dtype, int B, int D, int H, int W,
                                   int C, int cpg, int wsets, void* stream) {
    if (B <= 0 || C <= 0 || cpg <= 0 || C % cpg || wsets <= 0 || B % wsets) return -1003;
    if (dtype != PASSION_F32 && dtype != PASSION_BF16) return -1007;
    const long total = (long)B * D * H * W * C;
    hipStream_t st = (hipStream_t)stream;
    const int CB = gconv_cb(C);
    if (CB && (cpg == 1 || cpg == 4)) {
        const long V = (long)D * H * W;
        const int vpb = 256 / (CB >> 3);
        if (dtype == PASSION_F32) { if (cpg == 1) GCONV_VEC(float, 1, false, x, y, bias); else GCONV_VEC(float, 4, false, x, y, bias); }
        else { if (cpg == 1) GCONV_VEC(bf16_t, 1, false, x, y, bias); else GCONV_VEC(bf16_t, 4, false, x, y, bias); }
        LAUNCH_RET();
        return 0;
    }
    if (wsets != 1) return -1003;
    if (dtype == PASSION_F32)
        hipLaunchKernelGGL(gconv_fwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st, (const float*)x, weight, bias, (float*)y, B, D, H, W, C, cpg);
    else
        hipLaunchKernelGGL(gconv_fwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, st, (const bf16_t*)x, weight, bias, (bf16_t*)y, B, D, H, W, C, cpg);
    LAUNCH_RET();
    return 0;
}

#define GCONV_WG(TT, CPGV)                                                                                                                   \
    hipLaunchKernelGGL((gconv_wgrad_vec_kernel<TT, CPGV>), dim3((unsigned)nchunk, C / CB, wsets), dim3(256), 256 * 8 * CPGV * 4, st,          \
                       (const TT*)x, (const TT*)dy, part, B, D, H, W, C, CB, wsets, GC_CHUNK)

extern "C" int passion_gconv3d_bwd(const void* x, const float* weight, const void* dy, void* dx, float* dweight, float* dbias, float* part,
                                   int dtype, int B, int D, int H, int W, int C, int cpg, int wsets, void* stream) {
    if (B <= 0 || C <= 0 || cpg <= 0 || C % cpg || wsets <= 0 || B % wsets) return -1003;
    const long total = (long)B * D * H * W * C;
    const long nout = (long)C * cpg * 27;
    hipStream_t st = (hipStream_t)stream;
    if (dtype != PASSION_F32 && dtype != PASSION_BF16) return -1007;
    const int CB = gconv_cb(C);
    if (CB && (cpg == 1 || cpg == 4)) {
        const long V = (long)D * H * W;
        const int vpb = 256 / (CB >> 3);
        if (dx) {
            const float* nobias = nullptr;
            if (dtype == PASSION_F32) { if (cpg == 1) GCONV_VEC(float, 1, true, dy, dx, nobias); else GCONV_VEC(float, 4, true, dy, dx, nobias); }
            else { if (cpg == 1) GCONV_VEC(bf16_t, 1, true, dy, dx, nobias); else GCONV_VEC(bf16_t, 4, true, dy, dx, nobias); }
            LAUNCH_RET();
        }
        if (dweight) {
            if (!part) return -1003;
            const long Q = (long)(B / wsets) * V;
            const long nchunk = (Q + GC_CHUNK - 1) / GC_CHUNK;
            if (dtype == PASSION_F32) { if (cpg == 1) GCONV_WG(float, 1); else GCONV_WG(float, 4); }
            else { if (cpg == 1) GCONV_WG(bf16_t, 1); else GCONV_WG(bf16_t, 4); }
            LAUNCH_RET();
            const long n_ci = (long)wsets * C * cpg;
            hipLaunchKernelGGL(gconv_wfold_kernel, dim3((unsigned)((n_ci * 28 + 255) / 256)), dim3(256), 0, st, part, dweight, dbias, (int)nchunk, n_ci, cpg);
            LAUNCH_RET();
        }
        return 0;
    }
    if (wsets != 1) return -1003;
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
