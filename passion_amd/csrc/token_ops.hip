// Token-volume ops of the m2ftrans cross-token blocks (reference code/models/blocks.py:32-109 DepthWiseConvBlock /
// GroupConvBlock): LayerNorm over the channels of a channels-last tensor, fused with what follows it in the reference --
//     y = act(LN(x) * gamma + beta [+ res]),   act = identity | GELU (erf form, torch's default)
// forward and backward, for G independent weight sets over G equal row groups of one launch (the four key_map_* and four
// value_map_* blocks of a MultiMaskAttentionLayer share every shape: blocks.py:246-297).  torch needs a cast, a LayerNorm and
// a cast (forward) plus five kernels (backward) per call and a step of the model has 170 of them: the model was bound by
// the host's launch rate (3800 launches per step), not by the GPU.
// Layout: a row = C channels, C / 8 lanes per row (one 16-byte access per lane and tensor for bf16), 64 / (C / 8) rows per
// wavefront; row moments by butterfly shuffles inside the row's lanes.  Backward: gamma / beta gradients as per-block partial
// sums (rows of one group only) folded in block order by a second kernel: no atomics, results repeat bit for bit.
#include "common.h"
#include "passion_hip.h"

namespace {

constexpr int LN_MAX_C = 512;
constexpr float INV_SQRT2 = 0.70710678118654752f, INV_SQRT_2PI = 0.39894228040143268f;

template <typename T> __device__ __forceinline__ void ld8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void ld8<bf16_t>(const bf16_t* p, float (&v)[8]) { VecIO<bf16_t, 8>::load(p, v); }
template <> __device__ __forceinline__ void ld8<float>(const float* p, float (&v)[8]) {
    const f32x4 a = load4(p), b = load4(p + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
}
template <typename T> __device__ __forceinline__ void st8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void st8<bf16_t>(bf16_t* p, const float (&v)[8]) { VecIO<bf16_t, 8>::store(p, v); }
template <> __device__ __forceinline__ void st8<float>(float* p, const float (&v)[8]) {
    store4(p, (f32x4){v[0], v[1], v[2], v[3]});
    store4(p + 4, (f32x4){v[4], v[5], v[6], v[7]});
}

// sum over the `lpr` (power of two) consecutive lanes that hold one row
__device__ __forceinline__ float row_sum(float v, int lpr) {
    for (int o = lpr >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float gelu_f(float z) { return 0.5f * z * (1.f + erff(z * INV_SQRT2)); }
__device__ __forceinline__ float gelu_d(float z) {
    return 0.5f * (1.f + erff(z * INV_SQRT2)) + z * INV_SQRT_2PI * __expf(-0.5f * z * z);
}

// rows [0, rows) in G = rows / rpg groups; gamma / beta: [G][C]
// TO: type of y (and of dy in the backward): f32 tokens normalised straight into the bf16 operand of the GEMM that follows
template <typename T, typename TO>
__global__ __launch_bounds__(256) void ln_act_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const T* __restrict__ res, TO* __restrict__ y, float* __restrict__ stat,
                                                         long rows, int C, long rpg, float eps, int act) {
    const int lpr = C >> 3, rpw = 64 / lpr;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int sub = lane / lpr, cl = (lane % lpr) * 8;
    const float invC = 1.f / (float)C;
    for (long r0 = ((long)blockIdx.x * 4 + wv) * rpw; r0 < rows; r0 += (long)gridDim.x * 4 * rpw) {
        const long r = r0 + sub;
        const bool ok = r < rows;
        const long rr = ok ? r : rows - 1;
        float v[8];
        ld8<T>(x + rr * C + cl, v);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
        const float mean = row_sum(s, lpr) * invC;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[j] -= mean; q = __builtin_fmaf(v[j], v[j], q); }
        const float rstd = 1.f / sqrtf(row_sum(q, lpr) * invC + eps);
        const long g = rr / rpg;
        float ga[8], be[8], rv[8];
        ld8<float>(gamma + g * C + cl, ga);
        ld8<float>(beta + g * C + cl, be);
        if (res) ld8<T>(res + rr * C + cl, rv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float z = __builtin_fmaf(v[j] * rstd, ga[j], be[j]);
            if (res) z += rv[j];
            v[j] = act ? gelu_f(z) : z;
        }
        if (ok) {
            st8<TO>(y + r * C + cl, v);
            if ((lane % lpr) == 0) { stat[2 * r] = mean; stat[2 * r + 1] = rstd; }
        }
    }
}

// grid (nblk, G): block (k, g) owns rows g * rpg + [k * chunk, (k + 1) * chunk) and leaves its column sums in part[g][k][2][C]
template <typename T, typename TO>
__global__ __launch_bounds__(256) void ln_act_bwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const T* __restrict__ res, const float* __restrict__ stat, const TO* __restrict__ dy,
                                                         T* __restrict__ dx, T* __restrict__ dres, float* __restrict__ part,
                                                         int C, long rpg, long chunk, int act) {
    __shared__ float red[4 * 64 * 16];                 // [wave][lane][2][8]
    const int lpr = C >> 3, rpw = 64 / lpr;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int sub = lane / lpr, cl = (lane % lpr) * 8;
    const float invC = 1.f / (float)C;
    const int g = blockIdx.y;
    const long lo = (long)blockIdx.x * chunk, hi = lo + chunk < rpg ? lo + chunk : rpg;
    float ga[8], be[8], ag[8], ab[8];
    ld8<float>(gamma + (long)g * C + cl, ga);
    if (act) ld8<float>(beta + (long)g * C + cl, be);
#pragma unroll
    for (int j = 0; j < 8; ++j) { ag[j] = 0.f; ab[j] = 0.f; }
    for (long r0 = lo + (long)wv * rpw; r0 < hi; r0 += 4 * rpw) {
        const long rl = r0 + sub;
        const bool ok = rl < hi;
        const long r = (long)g * rpg + (ok ? rl : hi - 1);
        float xv[8], gv[8], rv[8];
        ld8<T>(x + r * C + cl, xv);
        ld8<TO>(dy + r * C + cl, gv);
        if (act && res) ld8<T>(res + r * C + cl, rv);
        const float mean = stat[2 * r], rstd = stat[2 * r + 1];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xh = (xv[j] - mean) * rstd;
            float dz = gv[j];
            if (act) {
                float z = __builtin_fmaf(xh, ga[j], be[j]);
                if (res) z += rv[j];
                dz *= gelu_d(z);
            }
            if (!ok) dz = 0.f;
            gv[j] = dz;                                  // dz (= dres)
            xv[j] = xh;
            ag[j] = __builtin_fmaf(dz, xh, ag[j]);
            ab[j] += dz;
            const float dh = dz * ga[j];
            s1 += dh;
            s2 = __builtin_fmaf(dh, xh, s2);
        }
        const float m1 = row_sum(s1, lpr) * invC, m2 = row_sum(s2, lpr) * invC;
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = rstd * (gv[j] * ga[j] - m1 - xv[j] * m2);
        if (ok) {
            st8<T>(dx + r * C + cl, o);
            if (dres) st8<T>(dres + r * C + cl, gv);
        }
    }
    // column sums: lanes with the same channel slice (lane % lpr) over the 4 waves and rpw row slots, fixed order
    float* mine = red + ((wv * 64 + lane) * 16);
#pragma unroll
    for (int j = 0; j < 8; ++j) { mine[j] = ag[j]; mine[8 + j] = ab[j]; }
    __syncthreads();
    float* out = part + ((long)g * gridDim.x + blockIdx.x) * 2 * C;
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        const int k = i / C, c = i % C, sl = c >> 3, j = c & 7;
        float t = 0.f;
        for (int w2 = 0; w2 < 4; ++w2)
            for (int s = 0; s < rpw; ++s) t += red[((w2 * 64 + s * lpr + sl) * 16) + k * 8 + j];
        out[i] = t;
    }
}

// dgamma[g][c] = sum_k part[g][k][0][c], dbeta[g][c] = sum_k part[g][k][1][c].  A block owns 16 consecutive values of one group; 16
// slices of the partial-block axis per value (one thread per value walked up to 256 dependent-latency loads: 21 us per launch, and
// m2ftrans runs ~100 LayerNorm backwards per step), folded through LDS in slice order: the order of the adds is fixed.
__global__ __launch_bounds__(256) void ln_fold_kernel(const float* __restrict__ part, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                      int G, int nblk, int C) {
    __shared__ float red[16][17];
    const int li = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const long i = (long)blockIdx.x * 16 + li;            // value index inside (g, k, c); 2 C is a multiple of 16
    const int g = (int)(i / (2L * C)), kc = (int)(i % (2L * C));
    float t = 0.f;
    if (g < G) {
        const float* p = part + (long)g * nblk * 2 * C + kc;
        float t0 = 0.f, t1 = 0.f;
        int b = sl;
        for (; b + 16 < nblk; b += 32) { t0 += p[(long)b * 2 * C]; t1 += p[(long)(b + 16) * 2 * C]; }
        if (b < nblk) t0 += p[(long)b * 2 * C];
        t = t0 + t1;
    }
    red[sl][li] = t;
    __syncthreads();
    if (sl == 0 && g < G) {
        float a = 0.f;
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) a += red[k2][li];
        const int k = kc / C, c = kc % C;
        (k ? dbeta : dgamma)[(long)g * C + c] = a;
    }
}

bool ln_shape_ok(long rows, int C, long rpg) {
    if (rows <= 0 || rpg <= 0 || rows % rpg || C < 8 || C > LN_MAX_C || (C & 7)) return false;
    const int lpr = C >> 3;
    return (lpr & (lpr - 1)) == 0;
}

int ln_nblk(long rpg, int C) {
    const int rpw = 64 / (C >> 3);
    const long per_pass = 4L * rpw;                       // rows one block covers per loop trip
    long n = (rpg + 4 * per_pass - 1) / (4 * per_pass);   // >= 4 trips per block
    return (int)hmax(1, hmin(n, 256));
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------
// The elementwise glue of the reference's Transformer blocks (code/models/mmformer.py:230-313) as ONE pass each way:
//     out = [x +] drop_2(drop_1(act(y)))          act = identity | GELU (erf form), drop_i = nn.Dropout(p) in training mode
// -- `x + self.dropout(self.proj_drop(proj(...)))` (SelfAttention.proj_drop inside PreNormDrop.dropout inside Residual: two dropouts
// in a row, :226 / :256 / :247), `dropout(gelu(h))` (FeedForward :271-273) and `x + dropout(h)` (:275 / :247).  torch ran them as 2-4
// launches forward (GELU, fused_dropout x 1-2, add) and as many backward (masked_scale, gelu_backward) over (tokens, 512 ... 4096)
// tensors: ~1 ms of ATen launches per step in all.  Keep bits: 16-bit fields of a two-round counter hash of (seed, element index)
// (field >= thr keeps; two dropouts take the two fields of one word: independent, as the reference's two modules are); the seed is
// drawn per call from torch's generator, a replayed HIP graph adds its step words (passion_set_step_words) like the attention kernels.
extern const unsigned* g_passion_step_words;      // api.cpp: passion_set_step_words
namespace {
__device__ __forceinline__ unsigned tok_hash(unsigned i, unsigned s0, unsigned s1) {
    unsigned x = i ^ s0;
    x = __umul24(x, 0xB5297Au) + (x >> 9);
    x ^= x >> 15;
    x += s1;
    x = __umul24(x, 0x68E31Du) + (x >> 11);
    x ^= x >> 14;
    return x;
}
// scale * keep of element i (ndrop = 1: the low field; 2: both)
__device__ __forceinline__ float tok_keep(unsigned i, unsigned s0, unsigned s1, unsigned thr, int ndrop, float scale) {
    if (!thr) return 1.f;
    const unsigned u = tok_hash(i, s0, s1);
    const bool k = (u & 0xffffu) >= thr && (ndrop < 2 || (u >> 16) >= thr);
    return k ? scale : 0.f;
}

template <typename TY, typename TO>
__global__ __launch_bounds__(256) void token_drop_fwd_kernel(const TY* __restrict__ y, const float* __restrict__ x, TO* __restrict__ out, long n8,
                                                             unsigned thr, float scale, unsigned s0, unsigned s1, const unsigned* soff, int ndrop, int act) {
    if (soff) { s0 += soff[0]; s1 += soff[1]; }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        float v[8], r[8];
        ld8<TY>(y + i * 8, v);
        if (x) ld8<float>(x + i * 8, r);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float z = act ? gelu_f(v[j]) : v[j];
            z *= tok_keep((unsigned)(i * 8 + j), s0, s1, thr, ndrop, scale);
            v[j] = x ? r[j] + z : z;
        }
        st8<TO>(out + i * 8, v);
    }
}
// dy = g * keep * scale [* gelu'(y)]
template <typename TY, typename TO>
__global__ __launch_bounds__(256) void token_drop_bwd_kernel(const TO* __restrict__ g, const TY* __restrict__ y, TY* __restrict__ dy, long n8,
                                                             unsigned thr, float scale, unsigned s0, unsigned s1, const unsigned* soff, int ndrop, int act) {
    if (soff) { s0 += soff[0]; s1 += soff[1]; }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        float gv[8], v[8];
        ld8<TO>(g + i * 8, gv);
        if (act) ld8<TY>(y + i * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float z = gv[j] * tok_keep((unsigned)(i * 8 + j), s0, s1, thr, ndrop, scale);
            if (act) z *= gelu_d(v[j]);
            gv[j] = z;
        }
        st8<TY>(dy + i * 8, gv);
    }
}

}  // namespace

// y_dtype / out_dtype: PASSION_BF16 | PASSION_F32 ((bf16, bf16), (bf16, f32), (f32, f32)); x: f32 residual or NULL; n % 8 == 0; act 0 / 1 (GELU)
extern "C" int passion_token_drop_fwd(const void* y, const float* x, void* out, int64_t n, int y_dtype, int out_dtype, float p, int ndrop, int act,
                                      uint64_t seed, void* stream) {
    if (n < 8 || n % 8 || n >= (1LL << 32) || !(p >= 0.f && p < 1.f) || ndrop < 1 || ndrop > 2) return -1011;
    if ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) return -1005;
    const unsigned thr = (unsigned)(p * 65536.0f + 0.5f);
    const float keep = 1.0f - (float)thr / 65536.0f, scale = thr ? (ndrop == 2 ? 1.0f / (keep * keep) : 1.0f / keep) : 1.0f;
    const long n8 = n / 8;
    const int blocks = (int)((n8 + 255) / 256 < 4096 ? (n8 + 255) / 256 : 4096);
    const unsigned s0 = (unsigned)(seed & 0xffffffffu), s1 = (unsigned)(seed >> 32);
    hipStream_t st = (hipStream_t)stream;
    if (y_dtype == PASSION_BF16 && out_dtype == PASSION_BF16)
        hipLaunchKernelGGL((token_drop_fwd_kernel<bf16_t, bf16_t>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)y, x, (bf16_t*)out, n8, thr, scale, s0, s1, g_passion_step_words, ndrop, act);
    else if (y_dtype == PASSION_BF16 && out_dtype == PASSION_F32)
        hipLaunchKernelGGL((token_drop_fwd_kernel<bf16_t, float>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)y, x, (float*)out, n8, thr, scale, s0, s1, g_passion_step_words, ndrop, act);
    else if (y_dtype == PASSION_F32 && out_dtype == PASSION_F32)
        hipLaunchKernelGGL((token_drop_fwd_kernel<float, float>), dim3(blocks), dim3(256), 0, st, (const float*)y, x, (float*)out, n8, thr, scale, s0, s1, g_passion_step_words, ndrop, act);
    else
        return -1007;
    LAUNCH_RET();
    return 0;
}

// g: gradient of out (out_dtype); dy (y_dtype) = g * keep * scale [* gelu'(y)]; the residual's gradient is g itself
extern "C" int passion_token_drop_bwd(const void* g, const void* y, void* dy, int64_t n, int y_dtype, int out_dtype, float p, int ndrop, int act,
                                      uint64_t seed, void* stream) {
    if (n < 8 || n % 8 || n >= (1LL << 32) || !(p >= 0.f && p < 1.f) || ndrop < 1 || ndrop > 2 || (act && !y)) return -1011;
    if ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dy)) & 15) return -1005;
    const unsigned thr = (unsigned)(p * 65536.0f + 0.5f);
    const float keep = 1.0f - (float)thr / 65536.0f, scale = thr ? (ndrop == 2 ? 1.0f / (keep * keep) : 1.0f / keep) : 1.0f;
    const long n8 = n / 8;
    const int blocks = (int)((n8 + 255) / 256 < 4096 ? (n8 + 255) / 256 : 4096);
    const unsigned s0 = (unsigned)(seed & 0xffffffffu), s1 = (unsigned)(seed >> 32);
    hipStream_t st = (hipStream_t)stream;
    if (y_dtype == PASSION_BF16 && out_dtype == PASSION_BF16)
        hipLaunchKernelGGL((token_drop_bwd_kernel<bf16_t, bf16_t>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)g, (const bf16_t*)y, (bf16_t*)dy, n8, thr, scale, s0, s1, g_passion_step_words, ndrop, act);
    else if (y_dtype == PASSION_BF16 && out_dtype == PASSION_F32)
        hipLaunchKernelGGL((token_drop_bwd_kernel<bf16_t, float>), dim3(blocks), dim3(256), 0, st, (const float*)g, (const bf16_t*)y, (bf16_t*)dy, n8, thr, scale, s0, s1, g_passion_step_words, ndrop, act);
    else if (y_dtype == PASSION_F32 && out_dtype == PASSION_F32)
        hipLaunchKernelGGL((token_drop_bwd_kernel<float, float>), dim3(blocks), dim3(256), 0, st, (const float*)g, (const float*)y, (float*)dy, n8, thr, scale, s0, s1, g_passion_step_words, ndrop, act);
    else
        return -1007;
    LAUNCH_RET();
    return 0;
}

extern "C" int passion_ln_act_supported(int64_t rows, int C, int64_t rows_per_group) { return ln_shape_ok(rows, C, rows_per_group) ? 1 : 0; }

extern "C" int64_t passion_ln_act_part_floats(int64_t rows, int C, int64_t rows_per_group) {
    if (!ln_shape_ok(rows, C, rows_per_group)) return 0;
    return (rows / rows_per_group) * (long)ln_nblk(rows_per_group, C) * 2 * C;
}

// out_dtype: type of y / dy; the input's type, or bf16 for f32 input (no residual then)
#define LN_DISPATCH(KERNEL, ...)                                                                                         \
    if (dtype == PASSION_F32 && out_dtype == PASSION_F32) { KERNEL(float, float, __VA_ARGS__); }                          \
    else if (dtype == PASSION_BF16 && out_dtype == PASSION_BF16) { KERNEL(bf16_t, bf16_t, __VA_ARGS__); }                 \
    else if (dtype == PASSION_F32 && out_dtype == PASSION_BF16 && !res) { KERNEL(float, bf16_t, __VA_ARGS__); }           \
    else return -1007;

extern "C" int passion_ln_act_fwd(const void* x, const float* gamma, const float* beta, const void* res, void* y, float* stat, int dtype,
                                  int out_dtype, int64_t rows, int C, int64_t rows_per_group, float eps, int act, void* stream) {
    if (!ln_shape_ok(rows, C, rows_per_group)) return -1003;
    const int rpw = 64 / (C >> 3);
    const long groups = (rows + 4L * rpw - 1) / (4L * rpw);
    const unsigned grid = (unsigned)hmin(groups, 8192);
    hipStream_t st = (hipStream_t)stream;
#define LN_FWD(T, TO, dummy)                                                                                                                   \
    hipLaunchKernelGGL((ln_act_fwd_kernel<T, TO>), dim3(grid), dim3(256), 0, st, (const T*)x, gamma, beta, (const T*)res, (TO*)y, stat, rows, C, \
                       rows_per_group, eps, act)
    LN_DISPATCH(LN_FWD, 0)
#undef LN_FWD
    LAUNCH_RET();
    return 0;
}

extern "C" int passion_ln_act_bwd(const void* x, const float* gamma, const float* beta, const void* res, const float* stat, const void* dy,
                                  void* dx, void* dres, float* part, float* dgamma, float* dbeta, int dtype, int out_dtype, int64_t rows, int C,
                                  int64_t rows_per_group, int act, void* stream) {
    if (!ln_shape_ok(rows, C, rows_per_group)) return -1003;
    const int G = (int)(rows / rows_per_group), nblk = ln_nblk(rows_per_group, C);
    const long chunk = (rows_per_group + nblk - 1) / nblk;
    hipStream_t st = (hipStream_t)stream;
#define LN_BWD(T, TO, dummy)                                                                                                                   \
    hipLaunchKernelGGL((ln_act_bwd_kernel<T, TO>), dim3(nblk, G), dim3(256), 0, st, (const T*)x, gamma, beta, (const T*)res, stat, (const TO*)dy, \
                       (T*)dx, (T*)dres, part, C, rows_per_group, chunk, act)
    LN_DISPATCH(LN_BWD, 0)
#undef LN_BWD
    LAUNCH_RET();
    hipLaunchKernelGGL(ln_fold_kernel, dim3((unsigned)(((long)G * 2 * C + 15) / 16)), dim3(256), 0, st, part, dgamma, dbeta, G, nblk, C);
    LAUNCH_RET();
    return 0;
}
