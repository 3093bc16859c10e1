// Trilinear x k resampling with align_corners=True on channels-last volumes.
// Replaces nn.Upsample(scale_factor=k, mode='trilinear', align_corners=True)
// (reference code/models/mmformer.py:70,75,80,85,143,366-370).  Source index arithmetic follows
// ATen's area_pixel_compute_source_index: src = dst * (in-1)/(out-1) evaluated in f32.
// Forward is an 8-tap gather; backward is the exact adjoint written as a gather as well (every input
// voxel sums the output voxels that referenced it), so there are no atomics and the result is
// deterministic.  For large k a whole wavefront shares one input voxel and reduces with shuffles.
#include "common.h"
#include "conv_common.h"
#include "passion_hip.h"

namespace {

struct Src { int i0, i1; float l0, l1; };

__device__ __forceinline__ Src src_index(int o, float scale, int in) {
    Src s;
    const float r = scale * (float)o;
    s.i0 = (int)r;
    if (s.i0 > in - 1) s.i0 = in - 1;
    s.i1 = s.i0 + (s.i0 < in - 1 ? 1 : 0);
    s.l1 = r - (float)s.i0;
    s.l0 = 1.f - s.l1;
    return s;
}

__host__ __device__ inline float axis_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

// IT: index type of the flattened (b, d, h, w, channel vector) index.  `unsigned` whenever the tensor has < 2^32 vectors:
// the five div/mod pairs per element are then 32-bit (a 64-bit division by a run-time value is ~100 instructions each
// and made this gather VALU-bound).
template <typename T, int VEC, typename IT>
__global__ __launch_bounds__(256) void upsample_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int D,
                                                           int H, int W, int C, int k) {
    const int Do = D * k, Ho = H * k, Wo = W * k, cvn = C / VEC;
    const float sd = axis_scale(D, Do), sh = axis_scale(H, Ho), sw = axis_scale(W, Wo);
    const IT total = (IT)B * Do * Ho * Wo * cvn;
    for (IT i = (IT)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (IT)gridDim.x * blockDim.x) {
        const int cv = (int)(i % (IT)cvn);
        IT v = i / (IT)cvn;
        const int ow = (int)(v % (IT)Wo); v /= (IT)Wo;
        const int oh = (int)(v % (IT)Ho); v /= (IT)Ho;
        const int od = (int)(v % (IT)Do);
        const int b = (int)(v / (IT)Do);
        const Src a = src_index(od, sd, D), e = src_index(oh, sh, H), f = src_index(ow, sw, W);
        const T* xb = x + (long)b * D * H * W * C + cv * VEC;
        float r[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) r[j] = 0.f;
        // same association as ATen: ((w-blend) h-blend) d-blend, weights applied innermost first
        const int di[2] = {a.i0, a.i1}, hi[2] = {e.i0, e.i1}, wi[2] = {f.i0, f.i1};
        const float dl[2] = {a.l0, a.l1}, hl[2] = {e.l0, e.l1}, wl[2] = {f.l0, f.l1};
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float rh[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) rh[j] = 0.f;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float t0[VEC], t1[VEC];
                VecIO<T, VEC>::load(xb + ((((long)di[p]) * H + hi[q]) * W + wi[0]) * C, t0);
                VecIO<T, VEC>::load(xb + ((((long)di[p]) * H + hi[q]) * W + wi[1]) * C, t1);
#pragma unroll
                for (int j = 0; j < VEC; ++j) rh[j] += hl[q] * (wl[0] * t0[j] + wl[1] * t1[j]);
            }
#pragma unroll
            for (int j = 0; j < VEC; ++j) r[j] += dl[p] * rh[j];
        }
        VecIO<T, VEC>::store(y + (long)i * VEC, r);
    }
}


// The same gather with the InstanceNorm statistics of the OUTPUT as a by-product (the decoder's d*_c1 convs normalise the
// upsampled tensor, reference mmformer.py:98-101 / blocks.py:313): grid (blocks, B), a thread keeps its channel vector, the
// (sum, sum of squares) of the stored values leave the block through the slot / ticket fold of conv_common.h -- no second
// pass over y, no statistics launches.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void upsample_fwd_stats_kernel(const T* __restrict__ x, T* __restrict__ y, int D, int H, int W,
                                                                 int C, int k, pconv::FwdStats fs) {
    __shared__ float red[256 * 2 * VEC];
    __shared__ float wsum[512];
    const int Do = D * k, Ho = H * k, Wo = W * k, cvn = C / VEC;
    const int b = blockIdx.y, tid = threadIdx.x;
    const float sd = axis_scale(D, Do), sh = axis_scale(H, Ho), sw = axis_scale(W, Wo);
    const unsigned total = (unsigned)Do * Ho * Wo * cvn;                   // per sample, < 2^32 (checked by the launcher)
    const T* xs = x + (long)b * D * H * W * C;
    T* ys = y + (long)b * Do * Ho * Wo * C;
    float s[VEC], q[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { s[j] = 0.f; q[j] = 0.f; }
    for (unsigned i = blockIdx.x * 256u + tid; i < total; i += gridDim.x * 256u) {      // stride % cvn == 0: same channels every time
        const int cv = (int)(i % (unsigned)cvn);
        unsigned v = i / (unsigned)cvn;
        const int ow = (int)(v % (unsigned)Wo); v /= (unsigned)Wo;
        const int oh = (int)(v % (unsigned)Ho);
        const int od = (int)(v / (unsigned)Ho);
        const Src a = src_index(od, sd, D), e = src_index(oh, sh, H), f = src_index(ow, sw, W);
        const T* xb = xs + cv * VEC;
        float r[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) r[j] = 0.f;
        const int di[2] = {a.i0, a.i1}, hi[2] = {e.i0, e.i1}, wi[2] = {f.i0, f.i1};
        const float dl[2] = {a.l0, a.l1}, hl[2] = {e.l0, e.l1}, wl[2] = {f.l0, f.l1};
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float rh[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) rh[j] = 0.f;
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                float t0[VEC], t1[VEC];
                VecIO<T, VEC>::load(xb + ((((long)di[p]) * H + hi[qq]) * W + wi[0]) * C, t0);
                VecIO<T, VEC>::load(xb + ((((long)di[p]) * H + hi[qq]) * W + wi[1]) * C, t1);
#pragma unroll
                for (int j = 0; j < VEC; ++j) rh[j] += hl[qq] * (wl[0] * t0[j] + wl[1] * t1[j]);
            }
#pragma unroll
            for (int j = 0; j < VEC; ++j) r[j] += dl[p] * rh[j];
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {                                     // what gets stored
            r[j] = to_f32(from_f32<T>(r[j]));
            s[j] += r[j];
            q[j] = __builtin_fmaf(r[j], r[j], q[j]);
        }
        VecIO<T, VEC>::store(ys + (long)i * VEC, r);
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) { red[tid * 2 * VEC + j] = s[j]; red[tid * 2 * VEC + VEC + j] = q[j]; }
    __syncthreads();
    // thread tid holds channel vector tid % cvn: channel c = cv * VEC + j, summed over the 256 / cvn threads in a fixed order
    for (int o = tid; o < C * 2; o += 256) {
        const int c = o >> 1, k2 = o & 1, cv = c / VEC, j = c % VEC;
        float t = 0.f;
        for (int l = cv; l < 256; l += cvn) t += red[l * 2 * VEC + k2 * VEC + j];
        wsum[o] = t;
    }
    __syncthreads();
    pconv::fwd_stats_commit(fs, C, b, blockIdx.x, 0, C, wsum, tid);
}

// The same result, bit for bit in y, from a row-wise schedule: the flat-index version above spends ~160 VALU instructions per
// 16-byte output on three run-time divisions and three source-coordinate evaluations and ~250 on scalar f32 blending -- the
// gather was VALU-bound at 1.4 TB/s (302 MB in 221 us for the sep decoder's 4 x 64^3 x 16 -> 128^3).  Here a block walks output
// rows (od, oh): the D / H source planes and weights are workgroup-uniform, a thread keeps ITS W source columns and weights
// for the whole launch (<= ROWV vectors of the row), and the blend runs on channel PAIRS (v_pk_mul_f32 / v_pk_add_f32) in the
// same order of operations as before: ((w-blend) h-blend) d-blend, weights applied innermost first, no contraction.
typedef float f32x2_r __attribute__((ext_vector_type(2)));
template <typename T, int VEC, int ROWV>
__global__ __launch_bounds__(256) void upsample_rows_fwd_stats_kernel(const T* __restrict__ x, T* __restrict__ y, int D, int H, int W,
                                                                      int C, int k, pconv::FwdStats fs) {
    __shared__ float red[256 * 2 * VEC];
    __shared__ float wsum[512];
    constexpr int NP = VEC / 2;
    const int Do = D * k, Ho = H * k, Wo = W * k, cvn = C / VEC;
    const int b = blockIdx.y, tid = threadIdx.x;
    const float sd = axis_scale(D, Do), sh = axis_scale(H, Ho), sw = axis_scale(W, Wo);
    const int rowvec = Wo * cvn;                                           // <= 256 * ROWV (checked by the launcher)
    const T* xs = x + (long)b * D * H * W * C;
    T* ys = y + (long)b * Do * Ho * Wo * C;
    int off0[ROWV], off1[ROWV];
    f32x2_r wl0[ROWV], wl1[ROWV];
    bool valid[ROWV];
#pragma unroll
    for (int v = 0; v < ROWV; ++v) {
        const int t = tid + 256 * v;
        valid[v] = t < rowvec;
        const int tc = valid[v] ? t : 0, ow = tc / cvn, cv = tc % cvn;     // 256 % cvn == 0: the same channel vector for every v
        const Src f = src_index(ow, sw, W);
        off0[v] = f.i0 * C + cv * VEC; off1[v] = f.i1 * C + cv * VEC;
        wl0[v] = (f32x2_r){f.l0, f.l0}; wl1[v] = (f32x2_r){f.l1, f.l1};
    }
    f32x2_r s[NP], q[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) { s[j] = (f32x2_r){0.f, 0.f}; q[j] = (f32x2_r){0.f, 0.f}; }
    const int nrows = Do * Ho;
    for (int r = blockIdx.x; r < nrows; r += gridDim.x) {
        const int od = r / Ho, oh = r - od * Ho;                            // workgroup-uniform
        const Src a = src_index(od, sd, D), e = src_index(oh, sh, H);
        const T* rowp[2][2] = {{xs + ((long)a.i0 * H + e.i0) * W * C, xs + ((long)a.i0 * H + e.i1) * W * C},
                               {xs + ((long)a.i1 * H + e.i0) * W * C, xs + ((long)a.i1 * H + e.i1) * W * C}};
        const f32x2_r dl[2] = {{a.l0, a.l0}, {a.l1, a.l1}}, hl[2] = {{e.l0, e.l0}, {e.l1, e.l1}};
        T* yr = ys + (long)r * Wo * C;
#pragma unroll
        for (int v = 0; v < ROWV; ++v) {
            if (!valid[v]) continue;
            float t0[2][2][VEC], t1[2][2][VEC];
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {                            // all eight loads first
                    VecIO<T, VEC>::load(rowp[p][qq] + off0[v], t0[p][qq]);
                    VecIO<T, VEC>::load(rowp[p][qq] + off1[v], t1[p][qq]);
                }
            float out[VEC];
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                f32x2_r rr = {0.f, 0.f};
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    f32x2_r rh = {0.f, 0.f};
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq) {
                        const f32x2_r a0 = {t0[p][qq][2 * j], t0[p][qq][2 * j + 1]}, a1 = {t1[p][qq][2 * j], t1[p][qq][2 * j + 1]};
                        rh = rh + hl[qq] * (wl0[v] * a0 + wl1[v] * a1);
                    }
                    rr = rr + dl[p] * rh;
                }
                const f32x2_r st = {to_f32(from_f32<T>(rr.x)), to_f32(from_f32<T>(rr.y))};      // what gets stored
                s[j] = s[j] + st;
                q[j] = __builtin_elementwise_fma(st, st, q[j]);
                out[2 * j] = st.x; out[2 * j + 1] = st.y;
            }
            VecIO<T, VEC>::store(yr + (long)(tid + 256 * v) * VEC, out);
        }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) { red[tid * 2 * VEC + j] = s[j >> 1][j & 1]; red[tid * 2 * VEC + VEC + j] = q[j >> 1][j & 1]; }
    __syncthreads();
    // thread tid holds channel vector tid % cvn: channel c = cv * VEC + j, summed over the 256 / cvn threads in a fixed order
    for (int o = tid; o < C * 2; o += 256) {
        const int c = o >> 1, k2 = o & 1, cv = c / VEC, j = c % VEC;
        float t = 0.f;
        for (int l = cv; l < 256; l += cvn) t += red[l * 2 * VEC + k2 * VEC + j];
        wsum[o] = t;
    }
    __syncthreads();
    pconv::fwd_stats_commit(fs, C, b, blockIdx.x, 0, C, wsum, tid);
}

// Row-wise with the D/H blend SHARED along W (PASSION_UPSAMPLE_SHARED, default on): the blended source row
//   u[iw][c] = dl0 (hl0 x[d0][h0][iw][c] + hl1 x[d0][h1][iw][c]) + dl1 (hl0 x[d1][h0][iw][c] + hl1 x[d1][h1][iw][c])
// is formed once per output row (W vectors, f32, in LDS) and every output of the row blends two of its entries along W -- each
// source vector is unpacked and weighted once instead of once per output that touches it (2k outputs): ~90 VALU instructions
// per 16-byte output instead of ~220, the gather is no longer VALU-bound.  The order of the f32 operations differs from the
// flat kernels (they blend W first, as ATen does): results agree to f32 rounding, not bit for bit -- so only bf16 tensors take
// this path (their store rounds at 2^-8; the f32 parity mode keeps ATen's order: one 96^3 RFNet gradient moved from under to
// just over its 5e-3 bound against the reference fixture otherwise).  passion_upsample_fwd and passion_upsample_fwd_stats both
// take it for the same shapes, so they still agree with each other exactly.
// fs.stats == nullptr: no statistics.
template <typename T, int VEC, int ROWV>
__global__ __launch_bounds__(256) void upsample_shared_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int D, int H, int W,
                                                                  int C, int k, pconv::FwdStats fs) {
    __shared__ float red[256 * 2 * VEC];
    __shared__ float wsum[512];
    extern __shared__ __attribute__((aligned(16))) float urow[];          // [2][W * C]
    constexpr int NP = VEC / 2;
    const int Do = D * k, Ho = H * k, Wo = W * k, cvn = C / VEC;
    const int b = blockIdx.y, tid = threadIdx.x;
    const float sd = axis_scale(D, Do), sh = axis_scale(H, Ho), sw = axis_scale(W, Wo);
    const int rowvec = Wo * cvn, srcvec = W * cvn;                        // rowvec <= 256 * ROWV, srcvec <= 256 (launcher)
    const T* xs = x + (long)b * D * H * W * C;
    T* ys = y + (long)b * Do * Ho * Wo * C;
    int off0[ROWV], off1[ROWV];
    f32x2_r wl0[ROWV], wl1[ROWV];
    bool valid[ROWV];
#pragma unroll
    for (int v = 0; v < ROWV; ++v) {
        const int t = tid + 256 * v;
        valid[v] = t < rowvec;
        const int tc = valid[v] ? t : 0, ow = tc / cvn, cv = tc % cvn;     // 256 % cvn == 0: the same channel vector for every v
        const Src f = src_index(ow, sw, W);
        off0[v] = f.i0 * C + cv * VEC; off1[v] = f.i1 * C + cv * VEC;      // float offsets into the blended row
        wl0[v] = (f32x2_r){f.l0, f.l0}; wl1[v] = (f32x2_r){f.l1, f.l1};
    }
    f32x2_r s[NP], q[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) { s[j] = (f32x2_r){0.f, 0.f}; q[j] = (f32x2_r){0.f, 0.f}; }
    const int nrows = Do * Ho;
    int buf = 0;
    for (int r = blockIdx.x; r < nrows; r += gridDim.x, buf ^= 1) {
        const int od = r / Ho, oh = r - od * Ho;                            // workgroup-uniform
        const Src a = src_index(od, sd, D), e = src_index(oh, sh, H);
        float* u = urow + buf * (W * C);
        if (tid < srcvec) {
            const f32x2_r dl0 = {a.l0, a.l0}, dl1 = {a.l1, a.l1}, hl0 = {e.l0, e.l0}, hl1 = {e.l1, e.l1};
            float t00[VEC], t01[VEC], t10[VEC], t11[VEC];
            const long o = (long)tid * VEC;
            VecIO<T, VEC>::load(xs + ((long)a.i0 * H + e.i0) * W * C + o, t00);
            VecIO<T, VEC>::load(xs + ((long)a.i0 * H + e.i1) * W * C + o, t01);
            VecIO<T, VEC>::load(xs + ((long)a.i1 * H + e.i0) * W * C + o, t10);
            VecIO<T, VEC>::load(xs + ((long)a.i1 * H + e.i1) * W * C + o, t11);
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const f32x2_r x00 = {t00[2 * j], t00[2 * j + 1]}, x01 = {t01[2 * j], t01[2 * j + 1]};
                const f32x2_r x10 = {t10[2 * j], t10[2 * j + 1]}, x11 = {t11[2 * j], t11[2 * j + 1]};
                const f32x2_r uu = dl0 * (hl0 * x00 + hl1 * x01) + dl1 * (hl0 * x10 + hl1 * x11);
                *reinterpret_cast<f32x2_r*>(u + o + 2 * j) = uu;
            }
        }
        __syncthreads();                                                   // (double-buffered: one barrier per row)
        T* yr = ys + (long)r * Wo * C;
#pragma unroll
        for (int v = 0; v < ROWV; ++v) {
            if (!valid[v]) continue;
            float out[VEC];
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const f32x2_r u0 = *reinterpret_cast<const f32x2_r*>(u + off0[v] + 2 * j);
                const f32x2_r u1 = *reinterpret_cast<const f32x2_r*>(u + off1[v] + 2 * j);
                const f32x2_r rr = wl0[v] * u0 + wl1[v] * u1;
                const f32x2_r st = {to_f32(from_f32<T>(rr.x)), to_f32(from_f32<T>(rr.y))};      // what gets stored
                s[j] = s[j] + st;
                q[j] = __builtin_elementwise_fma(st, st, q[j]);
                out[2 * j] = st.x; out[2 * j + 1] = st.y;
            }
            VecIO<T, VEC>::store(yr + (long)(tid + 256 * v) * VEC, out);
        }
    }
    if (!fs.stats) return;
#pragma unroll
    for (int j = 0; j < VEC; ++j) { red[tid * 2 * VEC + j] = s[j >> 1][j & 1]; red[tid * 2 * VEC + VEC + j] = q[j >> 1][j & 1]; }
    __syncthreads();
    for (int o = tid; o < C * 2; o += 256) {
        const int c = o >> 1, k2 = o & 1, cv = c / VEC, j = c % VEC;
        float t = 0.f;
        for (int l = cv; l < 256; l += cvn) t += red[l * 2 * VEC + k2 * VEC + j];
        wsum[o] = t;
    }
    __syncthreads();
    pconv::fwd_stats_commit(fs, C, b, blockIdx.x, 0, C, wsum, tid);
}

// shapes the shared-row kernel takes (both entry points ask the same question, so their outputs agree bit for bit)
static int g_upsample_shared = -1;
static bool shared_rows_ok(int D, int H, int W, int C, int k, int cvn) {
    if (g_upsample_shared < 0) { const char* e = getenv("PASSION_UPSAMPLE_SHARED"); g_upsample_shared = !(e && e[0] == '0'); }
    return g_upsample_shared && (long)W * k * cvn <= 512 && (long)W * cvn <= 256 && 256 % cvn == 0 && C <= 256 &&
           (long)D * k * H * k * W * k * cvn < (1L << 31);
}
template <typename T, int VEC>
static void launch_shared(const void* x, void* y, int B, int D, int H, int W, int C, int k, int blocks, const pconv::FwdStats& fs, hipStream_t st) {
    const size_t dyn = (size_t)2 * W * C * sizeof(float);
    if ((long)W * k * (C / VEC) <= 256)
        hipLaunchKernelGGL((upsample_shared_fwd_kernel<T, VEC, 1>), dim3(blocks, B), dim3(256), dyn, st, (const T*)x, (T*)y, D, H, W, C, k, fs);
    else
        hipLaunchKernelGGL((upsample_shared_fwd_kernel<T, VEC, 2>), dim3(blocks, B), dim3(256), dyn, st, (const T*)x, (T*)y, D, H, W, C, k, fs);
}

// weight with which output index o reads input index i along one axis
__device__ __forceinline__ float axis_weight(int o, int i, float scale, int in) {
    const Src s = src_index(o, scale, in);
    float w = 0.f;
    if (s.i0 == i) w += s.l0;
    if (s.i1 == i && s.i1 != s.i0) w += s.l1;
    if (s.i1 == i && s.i1 == s.i0) w += s.l1;  // clamped edge: both taps hit the same voxel
    return w;
}

__device__ __forceinline__ void axis_range(int i, float scale, int out, int& lo, int& hi) {
    // outputs whose source coordinate lies in (i-1, i+1); widened by one on each side for rounding safety
    if (scale <= 0.f) { lo = 0; hi = out - 1; return; }
    lo = (int)floorf((float)(i - 1) / scale) - 1;
    hi = (int)ceilf((float)(i + 1) / scale) + 1;
    if (lo < 0) lo = 0;
    if (hi > out - 1) hi = out - 1;
}

// G lanes cooperate on one (input voxel, 4-channel vector).
template <typename T, int G>
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int B, int D,
                                                           int H, int W, int C, int k) {
    const int Do = D * k, Ho = H * k, Wo = W * k, cvn = C / 4;
    const float sd = axis_scale(D, Do), sh = axis_scale(H, Ho), sw = axis_scale(W, Wo);
    const long items = (long)B * D * H * W * cvn;
    const int sub = threadIdx.x % G;
    for (long it = ((long)blockIdx.x * blockDim.x + threadIdx.x) / G; it < items + (G > 1 ? 0 : 0);
         it += ((long)gridDim.x * blockDim.x) / G) {
        const int cv = (int)(it % cvn);
        long v = it / cvn;
        const int iw = (int)(v % W); v /= W;
        const int ih = (int)(v % H); v /= H;
        const int id = (int)(v % D);
        const int b = (int)(v / D);
        int dlo, dhi, hlo, hhi, wlo, whi;
        axis_range(id, sd, Do, dlo, dhi);
        axis_range(ih, sh, Ho, hlo, hhi);
        axis_range(iw, sw, Wo, wlo, whi);
        const int nd = dhi - dlo + 1, nh = hhi - hlo + 1, nw = whi - wlo + 1;
        const T* dyb = dy + (long)b * Do * Ho * Wo * C + cv * 4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int j = sub; j < nd * nh * nw; j += G) {
            const int ow = wlo + j % nw, oh = hlo + (j / nw) % nh, od = dlo + j / (nw * nh);
            const float wd = axis_weight(od, id, sd, D);
            if (wd == 0.f) continue;
            const float wh = axis_weight(oh, ih, sh, H);
            if (wh == 0.f) continue;
            const float ww = axis_weight(ow, iw, sw, W);
            if (ww == 0.f) continue;
            acc += (wd * wh * ww) * load4(dyb + ((((long)od) * Ho + oh) * Wo + ow) * C);
        }
        if (G > 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = wave_sum(acc[j]);
        }
        if (sub == 0) store4(dx + it * 4, acc);
    }
}

// k > 2: the adjoint is separable (A_d x A_h x A_w), so it is applied one axis at a time, W first (the contiguous and
// largest reduction): [outer][No][inner] -> [outer][Ni][inner], one thread per output (4-channel vector), which sums
// the ~2k inputs along the axis that reference it.  Three passes read dy once and 1/k, 1/k^2 of it again -- the
// one-pass gather over the (2k+3)^3 candidate window took 250-500 us for the k = 4/8/16 deep-supervision maps.
// Four candidates per trip: their weights first, then four UNCONDITIONAL loads (clamped index, weight 0 past the window), then the
// sums in the same ascending order as before -- a load behind `if (w != 0)` serialised one memory latency per candidate (12-36 of
// them per output), which is what these launches spent their time on.  IT: 32-bit index arithmetic whenever the tensor allows.
template <typename TI, typename TO, typename IT>
__global__ __launch_bounds__(256) void upsample_bwd_axis_kernel(const TI* __restrict__ in, TO* __restrict__ out, long outer,
                                                                int Ni, int No, long inner_) {
    const float sc = axis_scale(Ni, No);
    const IT inner = (IT)inner_;
    const IT total = (IT)outer * (IT)Ni * inner;
    for (IT idx = (IT)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (IT)gridDim.x * blockDim.x) {
        const IT iv = idx % inner;
        const int i = (int)((idx / inner) % (IT)Ni);
        const IT o_ = idx / (inner * (IT)Ni);
        int lo, hi;
        axis_range(i, sc, No, lo, hi);
        const TI* base = in + ((long)(o_ * (IT)No) * inner_ + (long)iv) * 4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int o = lo; o <= hi; o += 4) {
            float w[4];
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int oo = min(o + u, hi);
                w[u] = (o + u <= hi) ? axis_weight(oo, i, sc, Ni) : 0.f;
                v[u] = load4(base + (long)oo * inner_ * 4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (w[u] != 0.f) acc += w[u] * v[u];          // (a select, not a branch around a load: same sums as before, bit for bit)
        }
        store4(out + (long)idx * 4, acc);
    }
}

template <typename TI, typename TO>
int launch_axis(const TI* in, TO* out, long outer, int Ni, int No, long inner, hipStream_t st) {
    const long total = outer * Ni * inner;
    const int blocks = (int)hmin((total + 255) / 256, 16384);
    if (total + 16384L * 256 < (1L << 31) && outer * No * inner < (1L << 29))
        hipLaunchKernelGGL((upsample_bwd_axis_kernel<TI, TO, unsigned>), dim3(blocks), dim3(256), 0, st, in, out, outer, Ni, No, inner);
    else
        hipLaunchKernelGGL((upsample_bwd_axis_kernel<TI, TO, long>), dim3(blocks), dim3(256), 0, st, in, out, outer, Ni, No, inner);
    LAUNCH_RET();
    return 0;
}

template <typename T>
int upsample_bwd_separable(const T* dy, T* dx, int B, int D, int H, int W, int C, int k, float* scratch, hipStream_t st) {
    const int Do = D * k, Ho = H * k, Wo = W * k;
    const long cv = C / 4;
    float* t1 = scratch;                                   // [B][Do][Ho][W][C]
    float* t2 = scratch + (long)B * Do * Ho * W * C;       // [B][Do][H][W][C]
    int rc = launch_axis<T, float>(dy, t1, (long)B * Do * Ho, W, Wo, cv, st);
    if (rc) return rc;
    rc = launch_axis<float, float>(t1, t2, (long)B * Do, H, Ho, (long)W * cv, st);
    if (rc) return rc;
    return launch_axis<float, T>(t2, dx, (long)B, D, Do, (long)H * W * cv, st);
}

// k <= 2: one thread per (input voxel, 4-channel vector).  At most 5 output indices per axis can reference an
// input index; their weights are evaluated once per axis (15 evaluations) instead of once per candidate triple.
template <typename T, int VEC, typename IT>
__global__ __launch_bounds__(256) void upsample_bwd_k2_kernel(const T* __restrict__ dy, T* __restrict__ dx, int B, int D,
                                                              int H, int W, int C, int k) {
    constexpr int NC = 6;
    const int Do = D * k, Ho = H * k, Wo = W * k, cvn = C / VEC;
    const float sd = axis_scale(D, Do), sh = axis_scale(H, Ho), sw = axis_scale(W, Wo);
    const IT items = (IT)B * D * H * W * cvn;
    for (IT it = (IT)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (IT)gridDim.x * blockDim.x) {
        const int cv = (int)(it % (IT)cvn);
        IT v = it / (IT)cvn;
        const int iw = (int)(v % (IT)W); v /= (IT)W;
        const int ih = (int)(v % (IT)H); v /= (IT)H;
        const int id = (int)(v % (IT)D);
        const int b = (int)(v / (IT)D);
        // outputs that can reference input i have src in (i-1, i+1), i.e. o in ((i-1)/scale, (i+1)/scale): a window of
        // 2/scale ~ 4.03 indices for k = 2.  Slots o = lo + j, j < NC, with statically indexed weights (zero = skip).
        const int dlo = max(0, (int)floorf((float)(id - 1) / sd)), hlo = max(0, (int)floorf((float)(ih - 1) / sh)),
                  wlo = max(0, (int)floorf((float)(iw - 1) / sw));
        float wd[NC], wh[NC], ww[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            wd[j] = (dlo + j < Do) ? axis_weight(dlo + j, id, sd, D) : 0.f;
            wh[j] = (hlo + j < Ho) ? axis_weight(hlo + j, ih, sh, H) : 0.f;
            ww[j] = (wlo + j < Wo) ? axis_weight(wlo + j, iw, sw, W) : 0.f;
        }
        const T* dyb = dy + (long)b * Do * Ho * Wo * C + cv * VEC;
        float acc[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
#pragma unroll
        for (int a = 0; a < NC; ++a) {
            if (wd[a] == 0.f) continue;
#pragma unroll
            for (int e = 0; e < NC; ++e) {
                if (wh[e] == 0.f) continue;
                const float wde = wd[a] * wh[e];
                const T* row = dyb + (((long)(dlo + a)) * Ho + (hlo + e)) * Wo * C;
#pragma unroll
                for (int f = 0; f < NC; ++f) {
                    if (ww[f] == 0.f) continue;
                    float t[VEC];
                    VecIO<T, VEC>::load(row + (long)(wlo + f) * C, t);
                    const float wt = wde * ww[f];
#pragma unroll
                    for (int j = 0; j < VEC; ++j) acc[j] += wt * t[j];
                }
            }
        }
        VecIO<T, VEC>::store(dx + (long)it * VEC, acc);
    }
}


// Row-wise x2 backward for bf16 tensors (the adjoint of upsample_shared_fwd_kernel's schedule).  upsample_bwd_k2_kernel gathers
// up to 6 x 6 x 6 candidate outputs per input vector, evaluates 18 axis weights per thread and unpacks 64 vectors: VALU-bound
// at 1.2 TB/s (268 MB of dY in 240 us for 4 x 64^3 x 16).  Here a block walks INPUT rows (id, ih): the <= 6 x 6 dY rows that
// reach it and their weights wd * wh are workgroup-uniform (twelve threads evaluate them one row ahead, through LDS); every
// thread sums ITS vector of those rows -- v[ow][c] = sum wd wh dY[od][oh][ow][c], the D/H part of the adjoint, once per output
// column instead of once per input that touches it -- into an f32 row in LDS, and the threads that own an input vector gather
// their <= 6 W taps (constant per thread for the whole launch) from that row.  f32 tensors keep the flat kernel (same sums in
// another order).
template <typename T, int VEC, int ROWV>
__global__ __launch_bounds__(256) void upsample_shared_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int D, int H, int W,
                                                                  int C, int k) {
    constexpr int NC = 6, NP = VEC / 2;
    extern __shared__ __attribute__((aligned(16))) float vrow[];          // [2][Wo * C]
    __shared__ float wts[2][2][NC];                                        // [buffer][axis d / h][candidate]
    __shared__ int los[2][2];
    const int Do = D * k, Ho = H * k, Wo = W * k, cvn = C / VEC;
    const int b = blockIdx.y, tid = threadIdx.x;
    const float sd = axis_scale(D, Do), sh = axis_scale(H, Ho), sw = axis_scale(W, Wo);
    const int rowvec = Wo * cvn, srcvec = W * cvn;
    const T* dys = dy + (long)b * Do * Ho * Wo * C;
    T* dxs = dx + (long)b * D * H * W * C;
    // input role: the W taps of this thread's input vector
    const bool in_thr = tid < srcvec;
    const int iw = in_thr ? tid / cvn : 0, cvi = tid % cvn;
    const int wlo = max(0, (int)floorf((float)(iw - 1) / sw));
    float ww[NC];
#pragma unroll
    for (int f = 0; f < NC; ++f) ww[f] = (in_thr && wlo + f < Wo) ? axis_weight(wlo + f, iw, sw, W) : 0.f;
    const int voff = wlo * C + cvi * VEC;
    const int nrows = D * H;
    auto row_weights = [&](int r, int bufw) {                              // threads 0 .. 2 NC - 1: one candidate each
        if (tid < 2 * NC && r < nrows) {
            const int ax = tid / NC, a = tid % NC, id = r / H, ih = r - id * H;
            const int i = ax ? ih : id, in = ax ? H : D, out = ax ? Ho : Do;
            const float sc = ax ? sh : sd;
            const int lo = max(0, (int)floorf((float)(i - 1) / sc));
            wts[bufw][ax][a] = (lo + a < out) ? axis_weight(lo + a, i, sc, in) : 0.f;
            if (a == 0) los[bufw][ax] = lo;
        }
    };
    row_weights(blockIdx.x, 0);
    __syncthreads();
    int buf = 0;
    for (int r = blockIdx.x; r < nrows; r += gridDim.x, buf ^= 1) {
        row_weights(r + gridDim.x, buf ^ 1);                               // visible after this iteration's barrier
        float* v = vrow + buf * (Wo * C);
        const int dlo = los[buf][0], hlo = los[buf][1];
        float wd[NC], wh[NC];
#pragma unroll
        for (int a = 0; a < NC; ++a) { wd[a] = wts[buf][0][a]; wh[a] = wts[buf][1][a]; }
#pragma unroll
        for (int u = 0; u < ROWV; ++u) {
            const int t = tid + 256 * u;
            if (t >= rowvec) continue;
            f32x2_r acc[NP];
#pragma unroll
            for (int j = 0; j < NP; ++j) acc[j] = (f32x2_r){0.f, 0.f};
#pragma unroll
            for (int a = 0; a < NC; ++a) {
                if (wd[a] == 0.f) continue;                                // workgroup-uniform
#pragma unroll
                for (int e = 0; e < NC; ++e) {
                    if (wh[e] == 0.f) continue;
                    const float wde = wd[a] * wh[e];
                    const f32x2_r w2 = {wde, wde};
                    float tv[VEC];
                    VecIO<T, VEC>::load(dys + (((long)(dlo + a)) * Ho + (hlo + e)) * Wo * C + (long)t * VEC, tv);
#pragma unroll
                    for (int j = 0; j < NP; ++j) acc[j] = __builtin_elementwise_fma(w2, (f32x2_r){tv[2 * j], tv[2 * j + 1]}, acc[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < NP; ++j) *reinterpret_cast<f32x2_r*>(v + t * VEC + 2 * j) = acc[j];
        }
        __syncthreads();
        if (in_thr) {
            f32x2_r o[NP];
#pragma unroll
            for (int j = 0; j < NP; ++j) o[j] = (f32x2_r){0.f, 0.f};
#pragma unroll
            for (int f = 0; f < NC; ++f) {
                if (ww[f] == 0.f) continue;                                // (also keeps the reads inside the row)
                const f32x2_r w2 = {ww[f], ww[f]};
#pragma unroll
                for (int j = 0; j < NP; ++j)
                    o[j] = __builtin_elementwise_fma(w2, *reinterpret_cast<const f32x2_r*>(v + voff + f * C + 2 * j), o[j]);
            }
            float out[VEC];
#pragma unroll
            for (int j = 0; j < NP; ++j) { out[2 * j] = o[j].x; out[2 * j + 1] = o[j].y; }
            VecIO<T, VEC>::store(dxs + ((long)r * W + iw) * C + cvi * VEC, out);
        }
    }
}

// The same adjoint over 2 x 2 BLOCKS of input rows (id, ih): the dY rows that reach a block are the union of its four rows' candidates
// (<= 8 per axis instead of 4 x <= 6) and every one of them is loaded ONCE per block and added into up to four accumulators --
// upsample_shared_bwd_kernel loads ~20 dY rows per input row (1.3 GB out of L2 for 268 MB of dY at 4 x 64^3 x 16: L2-bound at
// 1.5 TB/s), this one ~10.
__device__ __forceinline__ float sgpr_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
template <typename T, int VEC, int ROWV>
__global__ __launch_bounds__(256) void upsample_shared_bwd2_kernel(const T* __restrict__ dy, T* __restrict__ dx, int D, int H, int W,
                                                                   int C, int k) {
    constexpr int NC = 6, NC2 = 8, NP = VEC / 2;
    extern __shared__ __attribute__((aligned(16))) float vrow[];          // [4][Wo * C]
    __shared__ float wts[2][2][2][NC2];                                    // [buffer][axis d / h][row of the pair][candidate]
    __shared__ int los[2][2];
    const int Do = D * k, Ho = H * k, Wo = W * k, cvn = C / VEC;
    const int b = blockIdx.y, tid = threadIdx.x;
    const float sd = axis_scale(D, Do), sh = axis_scale(H, Ho), sw = axis_scale(W, Wo);
    const int rowvec = Wo * cvn, srcvec = W * cvn;
    const T* dys = dy + (long)b * Do * Ho * Wo * C;
    T* dxs = dx + (long)b * D * H * W * C;
    const bool in_thr = tid < srcvec;
    const int iw = in_thr ? tid / cvn : 0, cvi = tid % cvn;
    const int wlo = max(0, (int)floorf((float)(iw - 1) / sw));
    float ww[NC];
#pragma unroll
    for (int f = 0; f < NC; ++f) ww[f] = (in_thr && wlo + f < Wo) ? axis_weight(wlo + f, iw, sw, W) : 0.f;
    const int voff = wlo * C + cvi * VEC;
    const int Dp = (D + 1) / 2, Hp = (H + 1) / 2, nunits = Dp * Hp;
    auto unit_weights = [&](int u, int bufw) {                             // threads 0 .. 31: (axis, row of the pair, candidate)
        if (tid < 32 && u < nunits) {
            const int ax = tid >> 4, j = (tid >> 3) & 1, a = tid & 7;
            const int ip = ax ? u % Hp : u / Hp, i = 2 * ip + j, in = D + ax * (H - D), out = in * k;   // (no select tables in scratch)
            const float sc = ax ? sh : sd;
            const int lo = max(0, (int)floorf((float)(2 * ip - 1) / sc));
            wts[bufw][ax][j][a] = (i < in && lo + a < out) ? axis_weight(lo + a, i, sc, in) : 0.f;
            if (j == 0 && a == 0) los[bufw][ax] = lo;
        }
    };
    unit_weights(blockIdx.x, 0);
    __syncthreads();
    int buf = 0;
    for (int u = blockIdx.x; u < nunits; u += gridDim.x, buf ^= 1) {
        unit_weights(u + gridDim.x, buf ^ 1);                              // visible after this iteration's barriers
        const int idp = u / Hp, ihp = u % Hp;
        const int dlo = __builtin_amdgcn_readfirstlane(los[buf][0]), hlo = __builtin_amdgcn_readfirstlane(los[buf][1]);
        float wd[2][NC2], wh[2][NC2];
#pragma unroll
        for (int a = 0; a < NC2; ++a) {
            wd[0][a] = sgpr_f(wts[buf][0][0][a]); wd[1][a] = sgpr_f(wts[buf][0][1][a]);   // workgroup-uniform: scalar registers
            wh[0][a] = sgpr_f(wts[buf][1][0][a]); wh[1][a] = sgpr_f(wts[buf][1][1][a]);
        }
#pragma unroll
        for (int uu = 0; uu < ROWV; ++uu) {
            const int t = tid + 256 * uu;
            if (t >= rowvec) continue;
            f32x2_r acc[2][2][NP];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int p = 0; p < NP; ++p) acc[i][j][p] = (f32x2_r){0.f, 0.f};
#pragma unroll
            for (int a = 0; a < NC2; ++a) {
                if (wd[0][a] == 0.f && wd[1][a] == 0.f) continue;          // workgroup-uniform
#pragma unroll
                for (int e = 0; e < NC2; ++e) {
                    if (wh[0][e] == 0.f && wh[1][e] == 0.f) continue;
                    float tv[VEC];                                         // (all 8 rows of a plane in flight first: 138 registers, slower)
                    VecIO<T, VEC>::load(dys + (((long)(dlo + a)) * Ho + (hlo + e)) * Wo * C + (long)t * VEC, tv);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const float wde = wd[i][a] * wh[j][e];
                            const f32x2_r w2 = {wde, wde};
#pragma unroll
                            for (int p = 0; p < NP; ++p) acc[i][j][p] = __builtin_elementwise_fma(w2, (f32x2_r){tv[2 * p], tv[2 * p + 1]}, acc[i][j][p]);
                        }
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int p = 0; p < NP; ++p) *reinterpret_cast<f32x2_r*>(vrow + (i * 2 + j) * (Wo * C) + t * VEC + 2 * p) = acc[i][j][p];
        }
        __syncthreads();
        if (in_thr) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int id = 2 * idp + i, ih = 2 * ihp + j;
                    const float* v = vrow + (i * 2 + j) * (Wo * C);
                    f32x2_r o[NP];
#pragma unroll
                    for (int p = 0; p < NP; ++p) o[p] = (f32x2_r){0.f, 0.f};
#pragma unroll
                    for (int f = 0; f < NC; ++f)
                        if (ww[f] != 0.f) {                                // (also keeps the reads inside the row)
                            const f32x2_r w2 = {ww[f], ww[f]};
#pragma unroll
                            for (int p = 0; p < NP; ++p)
                                o[p] = __builtin_elementwise_fma(w2, *reinterpret_cast<const f32x2_r*>(v + voff + f * C + 2 * p), o[p]);
                        }
                    float out[VEC];
#pragma unroll
                    for (int p = 0; p < NP; ++p) { out[2 * p] = o[p].x; out[2 * p + 1] = o[p].y; }
                    if (id < D && ih < H) VecIO<T, VEC>::store(dxs + (((long)id * H + ih) * W + iw) * C + cvi * VEC, out);
                }
        }
        __syncthreads();                                                   // the rows are rewritten by the next unit
    }
}

// ---- depth-sharded variant (BASELINE.json configs[4]: one volume split along D over the ranks).
// A rank holds input planes [id0, id0 + Dl_in) of a volume of Dg planes (its slab plus one halo plane from each neighbour)
// and produces output planes [od0, od0 + Dl_out) of the k Dg output planes: the D axis uses GLOBAL align_corners coordinates,
// H and W are whole.  Backward = exact adjoint restricted to the local outputs, written as a gather over the local inputs
// (the halo planes' gradients go back to the neighbours).
template <typename T>
__global__ __launch_bounds__(256) void upsample_window_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int Dl_in, int id0,
                                                                  int Dg, int Dl_out, int od0, int H, int W, int C, int k) {
    const int Ho = H * k, Wo = W * k, cvn = C / 4;
    const float sd = axis_scale(Dg, Dg * k), sh = axis_scale(H, Ho), sw = axis_scale(W, Wo);
    const long total = (long)B * Dl_out * Ho * Wo * cvn;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % cvn);
        long v = i / cvn;
        const int ow = (int)(v % Wo); v /= Wo;
        const int oh = (int)(v % Ho); v /= Ho;
        const int od = (int)(v % Dl_out);
        const int b = (int)(v / Dl_out);
        const Src a = src_index(od0 + od, sd, Dg), e = src_index(oh, sh, H), f = src_index(ow, sw, W);
        const T* xb = x + (long)b * Dl_in * H * W * C + cv * 4;
        const int di[2] = {min(max(a.i0 - id0, 0), Dl_in - 1), min(max(a.i1 - id0, 0), Dl_in - 1)}, hi[2] = {e.i0, e.i1}, wi[2] = {f.i0, f.i1};
        const float dl[2] = {a.l0, a.l1}, hl[2] = {e.l0, e.l1}, wl[2] = {f.l0, f.l1};
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            f32x4 rh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const f32x4 t0 = load4(xb + ((((long)di[p]) * H + hi[q]) * W + wi[0]) * C);
                const f32x4 t1 = load4(xb + ((((long)di[p]) * H + hi[q]) * W + wi[1]) * C);
                rh += hl[q] * (wl[0] * t0 + wl[1] * t1);
            }
            r += dl[p] * rh;
        }
        store4(y + i * 4, r);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void upsample_window_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int B, int Dl_in, int id0,
                                                                  int Dg, int Dl_out, int od0, int H, int W, int C, int k) {
    const int Ho = H * k, Wo = W * k, cvn = C / 4;
    const float sd = axis_scale(Dg, Dg * k), sh = axis_scale(H, Ho), sw = axis_scale(W, Wo);
    const long items = (long)B * Dl_in * H * W * cvn;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(it % cvn);
        long v = it / cvn;
        const int iw = (int)(v % W); v /= W;
        const int ih = (int)(v % H); v /= H;
        const int idl = (int)(v % Dl_in);
        const int b = (int)(v / Dl_in);
        const int gi = id0 + idl;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (gi >= 0 && gi < Dg) {
            int dlo, dhi, hlo, hhi, wlo, whi;
            axis_range(gi, sd, Dg * k, dlo, dhi);
            axis_range(ih, sh, Ho, hlo, hhi);
            axis_range(iw, sw, Wo, wlo, whi);
            dlo = max(dlo, od0); dhi = min(dhi, od0 + Dl_out - 1);
            const T* dyb = dy + (long)b * Dl_out * Ho * Wo * C + cv * 4;
            for (int go = dlo; go <= dhi; ++go) {
                const float wd = axis_weight(go, gi, sd, Dg);
                if (wd == 0.f) continue;
                for (int oh = hlo; oh <= hhi; ++oh) {
                    const float wh = axis_weight(oh, ih, sh, H);
                    if (wh == 0.f) continue;
                    for (int ow = wlo; ow <= whi; ++ow) {
                        const float ww = axis_weight(ow, iw, sw, W);
                        if (ww == 0.f) continue;
                        acc += (wd * wh * ww) * load4(dyb + ((((long)(go - od0)) * Ho + oh) * Wo + ow) * C);
                    }
                }
            }
        }
        store4(dx + it * 4, acc);
    }
}

// plane d of every sample of a (B, D, P) tensor <-> a contiguous (B, P) buffer: the halo planes a rank sends / receives
template <typename T>
__global__ __launch_bounds__(256) void halo_pack_kernel(const T* __restrict__ x, T* __restrict__ buf, int B, int D, long P8, int d) {
    const long total = (long)B * P8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / P8, e = i % P8;
        reinterpret_cast<uint4*>(buf)[i] = reinterpret_cast<const uint4*>(x)[(b * D + d) * P8 + e];
    }
}
template <typename T>
__global__ __launch_bounds__(256) void halo_unpack_kernel(const T* __restrict__ buf, T* __restrict__ y, int B, int D, long P, int d, int accumulate) {
    const long total = (long)B * P;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / P, e = i % P;
        const long o = (b * D + d) * P + e;
        y[o] = accumulate ? from_f32<T>(to_f32(y[o]) + to_f32(buf[i])) : buf[i];
    }
}

// D pass of the separable adjoint of the WINDOWED upsampling (the W and H passes are the whole-axis ones above): input planes
// [od0, od0 + Dl_out) of the k Dg output planes, local input plane idl = global plane id0 + idl (planes outside [0, Dg): zero)
template <typename TO>
__global__ __launch_bounds__(256) void upsample_bwd_dwindow_kernel(const float* __restrict__ in, TO* __restrict__ out, int B, int Dl_in, int id0, int Dg,
                                                                   int Dl_out, int od0, long inner, int k) {
    const float sd = axis_scale(Dg, Dg * k);
    const long total = (long)B * Dl_in * inner;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long iv = idx % inner;
        const int idl = (int)((idx / inner) % Dl_in);
        const long b = idx / (inner * Dl_in);
        const int gi = id0 + idl;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (gi >= 0 && gi < Dg) {
            int lo, hi;
            axis_range(gi, sd, Dg * k, lo, hi);
            lo = max(lo, od0); hi = min(hi, od0 + Dl_out - 1);
            const float* base = in + ((b * Dl_out) * inner + iv) * 4;
            for (int o = lo; o <= hi; o += 4) {
                float w[4];
                f32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int oo = min(o + u, hi);
                    w[u] = (o + u <= hi) ? axis_weight(oo, gi, sd, Dg) : 0.f;
                    v[u] = load4(base + (long)(oo - od0) * inner * 4);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (w[u] != 0.f) acc += w[u] * v[u];
            }
        }
        store4(out + idx * 4, acc);
    }
}

// ---- depth sharding, round 5: ONE launch per halo extension / its adjoint, two tiny ones per global InstanceNorm statistic.
// (The first form built an extended slab from torch.zeros + copy_ + two plane packs + two plane puts, its adjoint from seven
// launches, the global statistics from ~20 elementwise torch ops on (B, C) tensors: ~17 000 launches per 240 x 240 x 160 step, and
// the step was bound by them on the host AND on the GPU.)
// ext (B, n_lo + Dl + n_hi, P): planes below n_lo - 1 and above n_lo + Dl zero; plane n_lo - 1 / n_lo + Dl from the low / high source:
// mode 0 zeros, 1 a (B, P) buffer (the neighbour's plane), 2 plane `pl` of x itself (reflect at a global end)
__global__ __launch_bounds__(256) void halo_extend_kernel(const uint4* __restrict__ x, const uint4* __restrict__ lo, const uint4* __restrict__ hi,
                                                          uint4* __restrict__ ext, int B, int Dl, long P8, int n_lo, int n_hi, int lo_mode, int hi_mode,
                                                          int lo_pl, int hi_pl) {
    const int De = n_lo + Dl + n_hi;
    const long total = (long)B * De * P8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long e = i % P8, bz = i / P8;
        const int z = (int)(bz % De);
        const long b = bz / De;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (z >= n_lo && z < n_lo + Dl) v = x[(b * Dl + (z - n_lo)) * P8 + e];
        else if (z == n_lo - 1) { if (lo_mode == 1) v = lo[b * P8 + e]; else if (lo_mode == 2) v = x[(b * Dl + lo_pl) * P8 + e]; }
        else if (z == n_lo + Dl) { if (hi_mode == 1) v = hi[b * P8 + e]; else if (hi_mode == 2) v = x[(b * Dl + hi_pl) * P8 + e]; }
        ext[i] = v;
    }
}

// adjoint: gx (B, Dl, P) = the owned planes of g (+ the halo plane's gradient folded onto plane fold_lo / fold_hi: reflect at a global end);
// send (2, B, P) = gradients of the two halo planes (zeros where there is none): they go back to their owners
template <typename T>
__global__ __launch_bounds__(256) void halo_extend_bwd_kernel(const T* __restrict__ g, T* __restrict__ gx, T* __restrict__ send, int B, int Dl, long P,
                                                              int n_lo, int n_hi, int fold_lo, int fold_hi) {
    constexpr int VE = 16 / (int)sizeof(T);
    const int De = n_lo + Dl + n_hi;
    const long PV = P / VE, nown = (long)B * Dl * PV, total = nown + 2L * B * PV;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        if (i < nown) {
            const long e = i % PV, bd = i / PV;
            const int d = (int)(bd % Dl);
            const long b = bd / Dl;
            union { uint4 u; T t[VE]; } a, c;
            a.u = *reinterpret_cast<const uint4*>(g + ((b * De + n_lo + d) * P + e * VE));
            if ((d == fold_lo && n_lo > 0) || (d == fold_hi && n_hi > 0)) {
                float acc[VE];
#pragma unroll
                for (int j = 0; j < VE; ++j) acc[j] = to_f32(a.t[j]);
                if (d == fold_lo && n_lo > 0) {
                    c.u = *reinterpret_cast<const uint4*>(g + ((b * De + n_lo - 1) * P + e * VE));
#pragma unroll
                    for (int j = 0; j < VE; ++j) acc[j] += to_f32(c.t[j]);
                }
                if (d == fold_hi && n_hi > 0) {
                    c.u = *reinterpret_cast<const uint4*>(g + ((b * De + n_lo + Dl) * P + e * VE));
#pragma unroll
                    for (int j = 0; j < VE; ++j) acc[j] += to_f32(c.t[j]);
                }
#pragma unroll
                for (int j = 0; j < VE; ++j) a.t[j] = from_f32<T>(acc[j]);
            }
            *reinterpret_cast<uint4*>(gx + i * VE) = a.u;
        } else if (send) {
            const long k = i - nown, e = k % PV, sb = k / PV;
            const int which = (int)(sb / B);
            const long b = sb % B;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (which == 0 && n_lo > 0) v = *reinterpret_cast<const uint4*>(g + ((b * De + n_lo - 1) * P + e * VE));
            if (which == 1 && n_hi > 0) v = *reinterpret_cast<const uint4*>(g + ((b * De + n_lo + Dl) * P + e * VE));
            *reinterpret_cast<uint4*>(send + k * VE) = v;
        }
    }
}

// (mean, rstd) f32 of a rank's slab (n voxels) -> the three moments the ranks sum (double): n, n mean, n (var + mean^2)
__global__ __launch_bounds__(256) void stats_moments_kernel(const float* __restrict__ st, double* __restrict__ mom, long BC, double n, double eps) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BC) return;
    const double mean = st[2 * i], rstd = st[2 * i + 1];
    double var = 1.0 / (rstd * rstd) - eps;
    if (var < 0.0) var = 0.0;
    mom[3 * i] = n; mom[3 * i + 1] = mean * n; mom[3 * i + 2] = (var + mean * mean) * n;
}
__global__ __launch_bounds__(256) void moments_stats_kernel(const double* __restrict__ mom, float* __restrict__ st, long BC, double eps) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BC) return;
    const double ng = mom[3 * i], gm = mom[3 * i + 1] / ng;
    double gv = mom[3 * i + 2] / ng - gm * gm;
    if (gv < 0.0) gv = 0.0;
    st[2 * i] = (float)gm;
    st[2 * i + 1] = (float)(1.0 / sqrt(gv + eps));
}

}  // namespace

// 32-bit index arithmetic unless the tensor (or the grid-stride overshoot) would overflow it
#define FWD_LAUNCH(T, VEC, xin, yout)                                                                                              \
    do {                                                                                                                           \
        if (total < (1L << 31))                                                                                                    \
            hipLaunchKernelGGL((upsample_fwd_kernel<T, VEC, unsigned>), dim3(blocks), dim3(256), 0, st, xin, yout, B, D, H, W, C, k); \
        else                                                                                                                       \
            hipLaunchKernelGGL((upsample_fwd_kernel<T, VEC, long>), dim3(blocks), dim3(256), 0, st, xin, yout, B, D, H, W, C, k);   \
    } while (0)
#define K2_LAUNCH(T, VEC, dyin, dxout)                                                                                               \
    do {                                                                                                                             \
        if (items < (1L << 31))                                                                                                      \
            hipLaunchKernelGGL((upsample_bwd_k2_kernel<T, VEC, unsigned>), dim3(blocks), dim3(256), 0, st, dyin, dxout, B, D, H, W, C, k); \
        else                                                                                                                         \
            hipLaunchKernelGGL((upsample_bwd_k2_kernel<T, VEC, long>), dim3(blocks), dim3(256), 0, st, dyin, dxout, B, D, H, W, C, k);   \
    } while (0)

extern "C" int passion_upsample_fwd(const void* x, void* y, int dtype, int B, int D, int H, int W, int C, int k,
                                    void* stream) {
    if (C % 4 || k < 1) return -1011;
    const bool v8 = dtype == PASSION_BF16 && C % 8 == 0;
    const long total = (long)B * D * k * H * k * W * k * (C / (v8 ? 8 : 4));
    const int blocks = (int)hmin((total + 255) / 256, 16384);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PASSION_BF16 && shared_rows_ok(D, H, W, C, k, C / (v8 ? 8 : 4))) {       // f32 tensors keep ATen's order of operations
        pconv::FwdStats none{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0.f, 0.0};
        const int rb = (int)hmin((long)D * k * H * k, 4096);
        if (dtype == PASSION_F32) launch_shared<float, 4>(x, y, B, D, H, W, C, k, rb, none, st);
        else if (v8) launch_shared<bf16_t, 8>(x, y, B, D, H, W, C, k, rb, none, st);
        else launch_shared<bf16_t, 4>(x, y, B, D, H, W, C, k, rb, none, st);
        LAUNCH_RET();
        return 0;
    }
    if (dtype == PASSION_F32)
        FWD_LAUNCH(float, 4, (const float*)x, (float*)y);
    else if (v8)
        FWD_LAUNCH(bf16_t, 8, (const bf16_t*)x, (bf16_t*)y);
    else if (dtype == PASSION_BF16)
        FWD_LAUNCH(bf16_t, 4, (const bf16_t*)x, (bf16_t*)y);
    else
        return -1007;
    LAUNCH_RET();
    return 0;
}


extern "C" int passion_upsample_fwd_stats(const void* x, void* y, int dtype, int B, int D, int H, int W, int C, int k,
                                          float* stats, float eps, void* stream) {
    if (C % 4 || k < 1 || !stats) return -1011;
    hipStream_t st = (hipStream_t)stream;
    const bool v8 = dtype == PASSION_BF16 && C % 8 == 0;
    const int cvn = C / (v8 ? 8 : 4);
    const long total = (long)D * k * H * k * W * k * cvn;                  // per sample
    if (C > 256 || 256 % cvn || total >= (1L << 32) || (dtype != PASSION_F32 && dtype != PASSION_BF16)) {
        // shapes the fused kernel does not take: the plain gather, then the statistics kernels over y
        int rc = passion_upsample_fwd(x, y, dtype, B, D, H, W, C, k, stream);
        if (rc) return rc;
        const int64_t V = (int64_t)D * k * H * k * W * k;
        void* plain = nullptr;
        rc = pconv::stream_scratch(st, 0, (size_t)passion_instnorm_scratch_doubles(B, V, C) * sizeof(double), nullptr, &plain);
        if (rc) return rc;
        return passion_instnorm_stats(y, dtype, B, V, C, V * C, eps, static_cast<double*>(plain), stats, stream);
    }
    const long rowvec = (long)W * k * cvn, nrows = (long)D * k * H * k;
    const bool rows = rowvec <= 512;                                        // row-wise schedule: a thread owns <= 2 vectors of a row
    // Workgroups (= slots of the statistics fold) per sample.  Every workgroup ends with the reduction of its sums through LDS, a slot
    // write, a ticket and -- for the last of each group -- a fold: that tail is worth ~16 rows of streaming, so a launch wants ~1024
    // workgroups in all and >= 16 rows each (measured, x2 to 128^3 x 16 at batch 4: 1024 per sample 146 us, 512: 120, 256: 101, 128: 145;
    // to 64^3 x 32: 90 -> 33 us; at batch 1 the old 1024 stays best: 46 us against 71 at 256)
    static const long stats_blocks = []() { const char* e = getenv("PASSION_UPSAMPLE_STATS_BLOCKS"); return e && atol(e) > 0 ? atol(e) : 0L; }();
    // (f32 = the parity mode keeps its 1024: the order of its sums is part of trajectories that are compared gate by gate)
    const long row_blocks = stats_blocks ? stats_blocks : (dtype == PASSION_BF16 ? hmax(1, hmin(hmax(128, 1024 / B), nrows / 16)) : 1024L);
    const int blocks = (int)hmin(rows ? nrows : (total + 255) / 256, rows ? row_blocks : 1024L);
    pconv::FwdStats fs;
    const int rc = pconv::fwd_stats_setup_raw(fs, B, C, blocks, 1, 1.0 / ((double)D * k * H * k * W * k), eps, stats, st);
    if (rc) return rc;
#define ROWS_LAUNCH(T, VEC)                                                                                                             \
    do {                                                                                                                                \
        if (rowvec <= 256)                                                                                                              \
            hipLaunchKernelGGL((upsample_rows_fwd_stats_kernel<T, VEC, 1>), dim3(blocks, B), dim3(256), 0, st, (const T*)x, (T*)y, D, H, W, C, k, fs); \
        else                                                                                                                            \
            hipLaunchKernelGGL((upsample_rows_fwd_stats_kernel<T, VEC, 2>), dim3(blocks, B), dim3(256), 0, st, (const T*)x, (T*)y, D, H, W, C, k, fs); \
    } while (0)
    if (rows && dtype == PASSION_BF16 && shared_rows_ok(D, H, W, C, k, cvn)) {
        if (dtype == PASSION_F32) launch_shared<float, 4>(x, y, B, D, H, W, C, k, blocks, fs, st);
        else if (v8) launch_shared<bf16_t, 8>(x, y, B, D, H, W, C, k, blocks, fs, st);
        else launch_shared<bf16_t, 4>(x, y, B, D, H, W, C, k, blocks, fs, st);
    } else
    if (rows) {
        if (dtype == PASSION_F32) ROWS_LAUNCH(float, 4);
        else if (v8) ROWS_LAUNCH(bf16_t, 8);
        else ROWS_LAUNCH(bf16_t, 4);
    } else
    if (dtype == PASSION_F32)
        hipLaunchKernelGGL((upsample_fwd_stats_kernel<float, 4>), dim3(blocks, B), dim3(256), 0, st, (const float*)x, (float*)y, D, H, W, C, k, fs);
    else if (v8)
        hipLaunchKernelGGL((upsample_fwd_stats_kernel<bf16_t, 8>), dim3(blocks, B), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, D, H, W, C, k, fs);
    else
        hipLaunchKernelGGL((upsample_fwd_stats_kernel<bf16_t, 4>), dim3(blocks, B), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, D, H, W, C, k, fs);
#undef ROWS_LAUNCH
    LAUNCH_RET();
    return 0;
}

// one axis pass of the separable adjoint on f32 4-channel vectors, for the fused loss gradients of loss.hip
namespace pconv {
int upsample_axis_f32(const float* in, float* out, long outer, int Ni, int No, long inner, hipStream_t st) {
    return launch_axis<float, float>(in, out, outer, Ni, No, inner, st);
}
}  // namespace pconv

extern "C" int64_t passion_upsample_bwd_scratch_floats(int B, int D, int H, int W, int C, int k) {
    if (k <= 2) return 0;
    return (int64_t)B * D * k * H * k * W * C + (int64_t)B * D * k * H * W * C;
}

extern "C" int passion_upsample_bwd(const void* dy, void* dx, int dtype, int B, int D, int H, int W, int C, int k,
                                    float* scratch, void* stream) {
    if (C % 4 || k < 1) return -1011;
    if (k > 2 && scratch) {
        hipStream_t st2 = (hipStream_t)stream;
        if (dtype == PASSION_F32) return upsample_bwd_separable<float>((const float*)dy, (float*)dx, B, D, H, W, C, k, scratch, st2);
        if (dtype == PASSION_BF16) return upsample_bwd_separable<bf16_t>((const bf16_t*)dy, (bf16_t*)dx, B, D, H, W, C, k, scratch, st2);
        return -1007;
    }
    const long items = (long)B * D * H * W * (C / 4);
    hipStream_t st = (hipStream_t)stream;
    if (k == 2 && D > 1 && H > 1 && W > 1 && dtype == PASSION_BF16 && shared_rows_ok(D, H, W, C, k, C / (C % 8 == 0 ? 8 : 4))) {
        const int rb = (int)hmin((long)D * H, 2048);
        const size_t dyn = (size_t)2 * W * k * C * sizeof(float);
        const bool one = (long)W * k * (C / (C % 8 == 0 ? 8 : 4)) <= 256;
        const bf16_t* g = (const bf16_t*)dy;
        bf16_t* o = (bf16_t*)dx;
        static const bool bwd2 = []() { const char* e = getenv("PASSION_UPSAMPLE_BWD2"); return !(e && e[0] == '0'); }();
        const size_t dyn2 = (size_t)4 * W * k * C * sizeof(float);
        const long units2 = (long)((D + 1) / 2) * ((H + 1) / 2);
        if (bwd2 && C % 8 == 0 && one && dyn2 <= 65536 && units2 * B >= 2048) {   // 2 x 2 blocks of input rows (large planes only:
            const int rb2 = (int)hmin(units2, 2048);                              //  fewer, heavier workgroups)
            hipLaunchKernelGGL((upsample_shared_bwd2_kernel<bf16_t, 8, 1>), dim3(rb2, B), dim3(256), dyn2, st, g, o, D, H, W, C, k);
        } else
        if (C % 8 == 0) {
            if (one) hipLaunchKernelGGL((upsample_shared_bwd_kernel<bf16_t, 8, 1>), dim3(rb, B), dim3(256), dyn, st, g, o, D, H, W, C, k);
            else hipLaunchKernelGGL((upsample_shared_bwd_kernel<bf16_t, 8, 2>), dim3(rb, B), dim3(256), dyn, st, g, o, D, H, W, C, k);
        } else {
            if (one) hipLaunchKernelGGL((upsample_shared_bwd_kernel<bf16_t, 4, 1>), dim3(rb, B), dim3(256), dyn, st, g, o, D, H, W, C, k);
            else hipLaunchKernelGGL((upsample_shared_bwd_kernel<bf16_t, 4, 2>), dim3(rb, B), dim3(256), dyn, st, g, o, D, H, W, C, k);
        }
    } else
    if (k == 2 && D > 1 && H > 1 && W > 1) {
        const bool v8 = dtype == PASSION_BF16 && C % 8 == 0;
        const int blocks = (int)hmin(((v8 ? items / 2 : items) + 255) / 256, 16384);
        if (dtype == PASSION_F32)
            K2_LAUNCH(float, 4, (const float*)dy, (float*)dx);
        else if (v8)
            K2_LAUNCH(bf16_t, 8, (const bf16_t*)dy, (bf16_t*)dx);
        else if (dtype == PASSION_BF16)
            K2_LAUNCH(bf16_t, 4, (const bf16_t*)dy, (bf16_t*)dx);
        else
            return -1007;
    } else {
        const int blocks = (int)hmin((items + 3) / 4, 16384);
        if (dtype == PASSION_F32)
            hipLaunchKernelGGL((upsample_bwd_kernel<float, 64>), dim3(blocks), dim3(256), 0, st, (const float*)dy, (float*)dx, B, D, H, W, C, k);
        else if (dtype == PASSION_BF16)
            hipLaunchKernelGGL((upsample_bwd_kernel<bf16_t, 64>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)dy, (bf16_t*)dx, B, D, H, W, C, k);
        else
            return -1007;
    }
    LAUNCH_RET();
    return 0;
}


extern "C" int passion_upsample_window_fwd(const void* x, void* y, int dtype, int B, int Dl_in, int id0, int Dg, int Dl_out, int od0,
                                           int H, int W, int C, int k, void* stream) {
    if (C % 4 || k < 1 || Dl_in < 1 || Dl_out < 1 || Dg < 1) return -1011;
    const long total = (long)B * Dl_out * H * k * W * k * (C / 4);
    const int blocks = (int)hmin((total + 255) / 256, 16384);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PASSION_F32)
        hipLaunchKernelGGL(upsample_window_fwd_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)x, (float*)y, B, Dl_in, id0, Dg, Dl_out, od0, H, W, C, k);
    else if (dtype == PASSION_BF16)
        hipLaunchKernelGGL(upsample_window_fwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, B, Dl_in, id0, Dg, Dl_out, od0, H, W, C, k);
    else
        return -1007;
    LAUNCH_RET();
    return 0;
}

extern "C" int passion_upsample_window_bwd(const void* dy, void* dx, int dtype, int B, int Dl_in, int id0, int Dg, int Dl_out, int od0,
                                           int H, int W, int C, int k, void* stream) {
    if (C % 4 || k < 1 || Dl_in < 1 || Dl_out < 1 || Dg < 1) return -1011;
    const long items = (long)B * Dl_in * H * W * (C / 4);
    const int blocks = (int)hmin((items + 255) / 256, 16384);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PASSION_F32)
        hipLaunchKernelGGL(upsample_window_bwd_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)dy, (float*)dx, B, Dl_in, id0, Dg, Dl_out, od0, H, W, C, k);
    else if (dtype == PASSION_BF16)
        hipLaunchKernelGGL(upsample_window_bwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)dy, (bf16_t*)dx, B, Dl_in, id0, Dg, Dl_out, od0, H, W, C, k);
    else
        return -1007;
    LAUNCH_RET();
    return 0;
}

/* the same adjoint as three separable passes (W, H whole-axis; D windowed) through `scratch`: the one-pass gather above walks the
 * (2k + 1)^3 candidate window per input voxel -- 4 ms per call for the x16 deep-supervision map of a 240 x 240 x 160 volume */
extern "C" int64_t passion_upsample_window_bwd_scratch_floats(int B, int Dl_out, int H, int W, int C, int k) {
    return (int64_t)B * Dl_out * H * k * W * C + (int64_t)B * Dl_out * H * W * C;
}
extern "C" int passion_upsample_window_bwd_sep(const void* dy, void* dx, int dtype, int B, int Dl_in, int id0, int Dg, int Dl_out, int od0,
                                               int H, int W, int C, int k, float* scratch, void* stream) {
    if (C % 4 || k < 1 || Dl_in < 1 || Dl_out < 1 || Dg < 1 || !scratch) return -1011;
    hipStream_t st = (hipStream_t)stream;
    const int Ho = H * k, Wo = W * k;
    const long cv = C / 4;
    float* t1 = scratch;                                        // [B][Dl_out][Ho][W][C]
    float* t2 = scratch + (long)B * Dl_out * Ho * W * C;         // [B][Dl_out][H][W][C]
    int rc;
    if (dtype == PASSION_F32) rc = launch_axis<float, float>((const float*)dy, t1, (long)B * Dl_out * Ho, W, Wo, cv, st);
    else if (dtype == PASSION_BF16) rc = launch_axis<bf16_t, float>((const bf16_t*)dy, t1, (long)B * Dl_out * Ho, W, Wo, cv, st);
    else return -1007;
    if (rc) return rc;
    rc = launch_axis<float, float>(t1, t2, (long)B * Dl_out, H, Ho, (long)W * cv, st);
    if (rc) return rc;
    const long inner = (long)H * W * cv, total = (long)B * Dl_in * inner;
    const int blocks = (int)hmin((total + 255) / 256, 16384);
    if (dtype == PASSION_F32)
        hipLaunchKernelGGL(upsample_bwd_dwindow_kernel<float>, dim3(blocks), dim3(256), 0, st, t2, (float*)dx, B, Dl_in, id0, Dg, Dl_out, od0, inner, k);
    else
        hipLaunchKernelGGL(upsample_bwd_dwindow_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, t2, (bf16_t*)dx, B, Dl_in, id0, Dg, Dl_out, od0, inner, k);
    LAUNCH_RET();
    return 0;
}

extern "C" int passion_halo_pack(const void* x, void* buf, int dtype, int B, int D, int64_t P, int d, void* stream) {
    const int es = dtype == PASSION_BF16 ? 2 : 4;
    if (B < 1 || D < 1 || d < 0 || d >= D || (P * es) % 16) return -1011;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(buf)) & 15) return -1005;
    const long P8 = P * es / 16, total = (long)B * P8;
    const int blocks = (int)hmin((total + 255) / 256, 8192);
    hipLaunchKernelGGL(halo_pack_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)buf, B, D, P8, d);
    LAUNCH_RET();
    return 0;
}

extern "C" int passion_halo_extend(const void* x, const void* lo, const void* hi, void* ext, int dtype, int B, int Dl, int64_t P, int n_lo, int n_hi,
                                   int lo_mode, int hi_mode, int lo_pl, int hi_pl, void* stream) {
    const int es = dtype == PASSION_BF16 ? 2 : 4;
    if (B < 1 || Dl < 1 || n_lo < 0 || n_hi < 0 || (P * es) % 16) return -1011;
    if ((lo_mode == 1 && !lo) || (hi_mode == 1 && !hi) || (lo_mode == 2 && (lo_pl < 0 || lo_pl >= Dl)) || (hi_mode == 2 && (hi_pl < 0 || hi_pl >= Dl))) return -1011;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(ext) | reinterpret_cast<uintptr_t>(lo) | reinterpret_cast<uintptr_t>(hi)) & 15) return -1005;
    const long P8 = P * es / 16, total = (long)B * (n_lo + Dl + n_hi) * P8;
    const int blocks = (int)hmin((total + 255) / 256, 16384);
    hipLaunchKernelGGL(halo_extend_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (const uint4*)lo, (const uint4*)hi, (uint4*)ext,
                       B, Dl, P8, n_lo, n_hi, lo_mode, hi_mode, lo_pl, hi_pl);
    LAUNCH_RET();
    return 0;
}

extern "C" int passion_halo_extend_bwd(const void* g, void* gx, void* send, int dtype, int B, int Dl, int64_t P, int n_lo, int n_hi, int fold_lo, int fold_hi,
                                       void* stream) {
    const int es = dtype == PASSION_BF16 ? 2 : 4;
    if (B < 1 || Dl < 1 || n_lo < 0 || n_hi < 0 || (P * es) % 16) return -1011;
    if ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(gx) | reinterpret_cast<uintptr_t>(send)) & 15) return -1005;
    const long total = ((long)B * Dl + 2L * B) * (P * es / 16);
    const int blocks = (int)hmin((total + 255) / 256, 16384);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PASSION_F32)
        hipLaunchKernelGGL(halo_extend_bwd_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)g, (float*)gx, (float*)send, B, Dl, (long)P, n_lo, n_hi, fold_lo, fold_hi);
    else if (dtype == PASSION_BF16)
        hipLaunchKernelGGL(halo_extend_bwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)g, (bf16_t*)gx, (bf16_t*)send, B, Dl, (long)P, n_lo, n_hi, fold_lo, fold_hi);
    else
        return -1007;
    LAUNCH_RET();
    return 0;
}

extern "C" int passion_stats_to_moments(const float* stats, double* mom, int64_t BC, double n, double eps, void* stream) {
    if (BC < 1) return -1011;
    hipLaunchKernelGGL(stats_moments_kernel, dim3((unsigned)((BC + 255) / 256)), dim3(256), 0, (hipStream_t)stream, stats, mom, (long)BC, n, eps);
    LAUNCH_RET();
    return 0;
}
extern "C" int passion_moments_to_stats(const double* mom, float* stats, int64_t BC, double eps, void* stream) {
    if (BC < 1) return -1011;
    hipLaunchKernelGGL(moments_stats_kernel, dim3((unsigned)((BC + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mom, stats, (long)BC, eps);
    LAUNCH_RET();
    return 0;
}

extern "C" int passion_halo_unpack(const void* buf, void* y, int dtype, int B, int D, int64_t P, int d, int accumulate, void* stream) {
    if (B < 1 || D < 1 || d < 0 || d >= D || P < 1) return -1011;
    const long total = (long)B * P;
    const int blocks = (int)hmin((total + 255) / 256, 8192);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PASSION_F32)
        hipLaunchKernelGGL(halo_unpack_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)buf, (float*)y, B, D, (long)P, d, accumulate);
    else if (dtype == PASSION_BF16)
        hipLaunchKernelGGL(halo_unpack_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)buf, (bf16_t*)y, B, D, (long)P, d, accumulate);
    else
        return -1007;
    LAUNCH_RET();
    return 0;
}
