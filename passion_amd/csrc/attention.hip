// Multi-head self-attention core: softmax(q k^T * scale) v per (batch, head)
// (reference code/models/mmformer.py:220-224; tokens N = (S/16)^3 per modality, 4N for the
// inter-modal transformer, 8 heads x 64).
//
// Round-1 structure: the score matrix is materialised in f32 (it is also what the backward needs),
// and all six contractions (QK^T, PV, P^T dO, dO V^T, dS K, dS^T Q) run through ONE strided batched
// GEMM on the matrix cores -- 64x64 output tile per workgroup, 2x2 MFMA 16x16x4 f32 tiles per wave,
// operands staged k-major through LDS so that both fragment reads are bank-conflict free.  Row
// softmax forward/backward are single-pass wavefront-per-row kernels.
#include "common.h"
#include "passion_hip.h"

namespace {

struct GemmP {
    const void* A; const void* Bm; void* C;
    int M, N, K;
    long am, ak, bk, bn, cm, cn;          // element strides
    long a_b1, a_b2, b_b1, b_b2, c_b1, c_b2;  // two-level batch strides
    int nb2;                              // batch = b1 * nb2 + b2
    float alpha;
};

constexpr int GT = 64, GK = 16, GLD = GT + 16;  // +16: lane groups kq, kq+1 hit disjoint bank halves

template <typename TA, typename TB, typename TC>
__global__ __launch_bounds__(256) void bgemm_kernel(GemmP p) {
    __shared__ float As[GK * GLD];
    __shared__ float Bs[GK * GLD];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, n = lane & 15, kq = lane >> 4;
    const int batch = blockIdx.z, b1 = batch / p.nb2, b2 = batch % p.nb2;
    const TA* A = reinterpret_cast<const TA*>(p.A) + b1 * p.a_b1 + b2 * p.a_b2;
    const TB* Bm = reinterpret_cast<const TB*>(p.Bm) + b1 * p.b_b1 + b2 * p.b_b2;
    TC* C = reinterpret_cast<TC*>(p.C) + b1 * p.c_b1 + b2 * p.c_b2;
    const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
    const int wm = (wv >> 1) * 32, wn = (wv & 1) * 32;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < p.K; k0 += GK) {
        __syncthreads();
        // stage A[m0..+64][k0..+16] -> As[k][m], B[k0..+16][n0..+64] -> Bs[k][n]; thread order follows
        // whichever global axis is contiguous
#pragma unroll
        for (int r = 0; r < (GT * GK) / 256; ++r) {
            const int e = r * 256 + tid;
            int kk, mm;
            if (p.ak == 1) { kk = e % GK; mm = e / GK; } else { mm = e % GT; kk = e / GT; }
            float v = 0.f;
            if (m0 + mm < p.M && k0 + kk < p.K) v = to_f32(A[(long)(m0 + mm) * p.am + (long)(k0 + kk) * p.ak]);
            As[kk * GLD + mm] = v;
            int kb, nn;
            if (p.bk == 1) { kb = e % GK; nn = e / GK; } else { nn = e % GT; kb = e / GT; }
            v = 0.f;
            if (n0 + nn < p.N && k0 + kb < p.K) v = to_f32(Bm[(long)(k0 + kb) * p.bk + (long)(n0 + nn) * p.bn]);
            Bs[kb * GLD + nn] = v;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < GK / 4; ++ks) {
            const int kk = ks * 4 + kq;
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[kk * GLD + wm + i * 16 + n];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = Bs[kk * GLD + wn + j * 16 + n];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    // D layout: column = lane & 15, rows = 4 * (lane >> 4) + r
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm + i * 16 + kq * 4 + r, col = n0 + wn + j * 16 + n;
                if (row < p.M && col < p.N) C[(long)row * p.cm + (long)col * p.cn] = from_f32<TC>(p.alpha * acc[i][j][r]);
            }
}

// ---- bf16 matrix-core version (used when the tensors are bf16: the bench dtype) --------------------------------
// 64x64 tile, K chunk 32, v_mfma_f32_16x16x32_bf16.  Either operand may be contiguous along k (row image, fragment =
// one ds_read_b128) or along m/n (k-major image, fragment = two ds_read_b64_tr_b16): no transposing copies, the six
// attention products only differ in strides.  f32 inputs (the probabilities) are rounded to bf16 while staging.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
constexpr int BK = 32;
constexpr int ROWS = BK * 2 + 16;      // bytes per row of a [64][32] row image (k contiguous)
constexpr int ROWT = GT * 2 + 16;      // bytes per row of a [32][64] k-major image

template <typename T>
__device__ __forceinline__ bf16x8_t load_run8(const T* base, long off, long stride, int valid) {
    // 8 elements base[off + i*stride], i < valid (others 0); vectorised when contiguous, aligned and complete
    bf16x8_t r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (__bf16)0.f;
    if (valid <= 0) return r;
    const T* p = base + off;
    if (stride == 1 && valid == 8 && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
        if (sizeof(T) == 2) return *reinterpret_cast<const bf16x8_t*>(p);
        const f32x4 lo = *reinterpret_cast<const f32x4*>(p), hi = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { r[i] = (__bf16)lo[i]; r[4 + i] = (__bf16)hi[i]; }
        return r;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) if (i < valid) r[i] = (__bf16)to_f32(p[i * stride]);
    return r;
}

__device__ __forceinline__ bf16x8_t tr8(const char* p0, const char* p1) {
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(p0));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(p1));
    return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <typename TA, typename TB, typename TC>
__global__ __launch_bounds__(256) void bgemm_bf16_kernel(GemmP p) {
    __shared__ __attribute__((aligned(16))) char As[GT * ROWS > BK * ROWT ? GT * ROWS : BK * ROWT];
    __shared__ __attribute__((aligned(16))) char Bs[GT * ROWS > BK * ROWT ? GT * ROWS : BK * ROWT];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, n = lane & 15, g = lane >> 4, q = n >> 2, pq = n & 3;
    const int batch = blockIdx.z, b1 = batch / p.nb2, b2 = batch % p.nb2;
    const TA* A = reinterpret_cast<const TA*>(p.A) + b1 * p.a_b1 + b2 * p.a_b2;
    const TB* Bm = reinterpret_cast<const TB*>(p.Bm) + b1 * p.b_b1 + b2 * p.b_b2;
    TC* C = reinterpret_cast<TC*>(p.C) + b1 * p.c_b1 + b2 * p.c_b2;
    const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
    const int wm = (wv >> 1) * 32, wn = (wv & 1) * 32;
    const bool a_row = (p.ak == 1), b_row = (p.bk == 1);     // operand contiguous along k -> row image
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < p.K; k0 += BK) {
        __syncthreads();
        if (a_row) {            // thread: row m = tid/4, 8 consecutive k
            const int m = tid >> 2, k8 = (tid & 3) * 8;
            const int valid = (m0 + m < p.M) ? min(8, p.K - (k0 + k8)) : 0;
            *reinterpret_cast<bf16x8_t*>(As + m * ROWS + k8 * 2) = load_run8(A, (long)(m0 + m) * p.am + (k0 + k8), 1, valid);
        } else {                // thread: k = tid/8, 8 consecutive m (or strided when neither axis is contiguous)
            const int k = tid >> 3, m8 = (tid & 7) * 8;
            const int valid = (k0 + k < p.K) ? min(8, p.M - (m0 + m8)) : 0;
            *reinterpret_cast<bf16x8_t*>(As + k * ROWT + m8 * 2) = load_run8(A, (long)(k0 + k) * p.ak + (long)(m0 + m8) * p.am, p.am, valid);
        }
        if (b_row) {
            const int nn = tid >> 2, k8 = (tid & 3) * 8;
            const int valid = (n0 + nn < p.N) ? min(8, p.K - (k0 + k8)) : 0;
            *reinterpret_cast<bf16x8_t*>(Bs + nn * ROWS + k8 * 2) = load_run8(Bm, (long)(n0 + nn) * p.bn + (k0 + k8), 1, valid);
        } else {
            const int k = tid >> 3, n8 = (tid & 7) * 8;
            const int valid = (k0 + k < p.K) ? min(8, p.N - (n0 + n8)) : 0;
            *reinterpret_cast<bf16x8_t*>(Bs + k * ROWT + n8 * 2) = load_run8(Bm, (long)(k0 + k) * p.bk + (long)(n0 + n8) * p.bn, p.bn, valid);
        }
        __syncthreads();
        bf16x8_t a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (a_row) a[i] = *reinterpret_cast<const bf16x8_t*>(As + (wm + i * 16 + n) * ROWS + g * 16);
            else a[i] = tr8(As + (8 * g + q) * ROWT + (wm + i * 16 + 4 * pq) * 2, As + (8 * g + 4 + q) * ROWT + (wm + i * 16 + 4 * pq) * 2);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (b_row) b[j] = *reinterpret_cast<const bf16x8_t*>(Bs + (wn + j * 16 + n) * ROWS + g * 16);
            else b[j] = tr8(Bs + (8 * g + q) * ROWT + (wn + j * 16 + 4 * pq) * 2, Bs + (8 * g + 4 + q) * ROWT + (wn + j * 16 + 4 * pq) * 2);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm + i * 16 + g * 4 + r, col = n0 + wn + j * 16 + n;
                if (row < p.M && col < p.N) C[(long)row * p.cm + (long)col * p.cn] = from_f32<TC>(p.alpha * acc[i][j][r]);
            }
}

template <typename T> struct is_bf16 { static constexpr bool value = false; };
template <> struct is_bf16<bf16_t> { static constexpr bool value = true; };

template <typename TA, typename TB, typename TC>
int launch_gemm(const GemmP& p, int nbatch, hipStream_t st) {
    dim3 grid(cdiv(p.N, GT), cdiv(p.M, GT), nbatch);
    // bf16 tensors (qkv / out / dqkv in bench mode) -> bf16 matrix cores; the all-f32 parity mode keeps the exact f32 MFMA
    if (is_bf16<TA>::value || is_bf16<TB>::value || is_bf16<TC>::value)
        hipLaunchKernelGGL((bgemm_bf16_kernel<TA, TB, TC>), grid, dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL((bgemm_kernel<TA, TB, TC>), grid, dim3(256), 0, st, p);
    LAUNCH_RET();
    return 0;
}

// one wavefront per row, in place, the row held in registers: one read and one write of the (B, H, N, N) scores instead of
// three reads and two writes (N = 512 intra-modal, 2048 inter-modal: NV = N / 256 float4 per lane)
template <int NV>
__global__ __launch_bounds__(256) void row_softmax_reg_kernel(float* __restrict__ s, long rows) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    f32x4* r = reinterpret_cast<f32x4*>(s + row * (NV * 256)) + lane;
    f32x4 v[NV];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        v[j] = r[j * 64];
        m = fmaxf(m, fmaxf(fmaxf(v[j][0], v[j][1]), fmaxf(v[j][2], v[j][3])));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) { v[j][c] = expf(v[j][c] - m); sum += v[j][c]; }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
#pragma unroll
    for (int j = 0; j < NV; ++j) r[j * 64] = v[j] * inv;
}

template <int NV>
__global__ __launch_bounds__(256) void row_softmax_bwd_reg_kernel(const float* __restrict__ pr, float* __restrict__ dp, long rows,
                                                                  float scale) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const f32x4* p = reinterpret_cast<const f32x4*>(pr + row * (NV * 256)) + lane;
    f32x4* g = reinterpret_cast<f32x4*>(dp + row * (NV * 256)) + lane;
    f32x4 pv[NV], gv[NV];
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        pv[j] = p[j * 64]; gv[j] = g[j * 64];
#pragma unroll
        for (int c = 0; c < 4; ++c) dot += pv[j][c] * gv[j][c];
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = scale * pv[j][c] * (gv[j][c] - dot);
        g[j * 64] = o;
    }
}

// any row length: three passes through memory
__global__ __launch_bounds__(256) void row_softmax_kernel(float* __restrict__ s, long rows, int n) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float* r = s + row * n;
    float m = -INFINITY;
    for (int i = lane; i < n; i += 64) m = fmaxf(m, r[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float sum = 0.f;
    for (int i = lane; i < n; i += 64) { const float e = expf(r[i] - m); r[i] = e; sum += e; }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    for (int i = lane; i < n; i += 64) r[i] *= inv;
}

// dS = scale * P o (dP - rowsum(P o dP)), in place on dP
__global__ __launch_bounds__(256) void row_softmax_bwd_kernel(const float* __restrict__ pr, float* __restrict__ dp,
                                                              long rows, int n, float scale) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* p = pr + row * n;
    float* g = dp + row * n;
    float dot = 0.f;
    for (int i = lane; i < n; i += 64) dot += p[i] * g[i];
    dot = wave_sum(dot);
    for (int i = lane; i < n; i += 64) g[i] = scale * p[i] * (g[i] - dot);
}

// in-place softmax of row (b, h, q) of scores [B][H][N][N] over the keys the block mask allows: allow[b][q / n][key / n], n = N / G
__global__ __launch_bounds__(256) void masked_softmax_kernel(float* __restrict__ s, const uint8_t* __restrict__ allow, long rows, int N,
                                                             int H, int G) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const int q = (int)(row % N);
    const int b = (int)(row / ((long)N * H));
    const int n = N / G;
    const uint8_t* al = allow + ((long)b * G + q / n) * G;
    float* r = s + row * N;
    float m = -INFINITY;
    for (int i = lane; i < N; i += 64) if (al[i / n]) m = fmaxf(m, r[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float sum = 0.f;
    for (int i = lane; i < N; i += 64) {
        const float e = al[i / n] ? expf(r[i] - m) : 0.f;
        r[i] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    for (int i = lane; i < N; i += 64) r[i] *= inv;
}

template <typename T>
int attn_scores(const void* qkv, int B, int N, int H, int Dh, float scale, float* probs, hipStream_t st, const uint8_t* allow = nullptr,
                int G = 0) {
    const long rs = 3L * H * Dh;
    GemmP p;
    p.A = qkv; p.Bm = reinterpret_cast<const T*>(qkv) + (long)H * Dh; p.C = probs;
    p.M = N; p.N = N; p.K = Dh;
    p.am = rs; p.ak = 1; p.bk = 1; p.bn = rs; p.cm = N; p.cn = 1;
    p.a_b1 = (long)N * rs; p.a_b2 = Dh; p.b_b1 = (long)N * rs; p.b_b2 = Dh;
    p.c_b1 = (long)H * N * N; p.c_b2 = (long)N * N; p.nb2 = H; p.alpha = scale;
    int rc = launch_gemm<T, T, float>(p, B * H, st);
    if (rc) return rc;
    const long rows = (long)B * H * N;
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (allow) hipLaunchKernelGGL(masked_softmax_kernel, grid, dim3(256), 0, st, probs, allow, rows, N, H, G);
    else if (N == 2048) hipLaunchKernelGGL(row_softmax_reg_kernel<8>, grid, dim3(256), 0, st, probs, rows);
    else if (N == 1024) hipLaunchKernelGGL(row_softmax_reg_kernel<4>, grid, dim3(256), 0, st, probs, rows);
    else if (N == 512) hipLaunchKernelGGL(row_softmax_reg_kernel<2>, grid, dim3(256), 0, st, probs, rows);
    else if (N == 256) hipLaunchKernelGGL(row_softmax_reg_kernel<1>, grid, dim3(256), 0, st, probs, rows);
    else hipLaunchKernelGGL(row_softmax_kernel, grid, dim3(256), 0, st, probs, rows, N);
    LAUNCH_RET();
    return 0;
}

template <typename T>
int attn_pv(const float* probs, const void* qkv, int B, int N, int H, int Dh, void* out, hipStream_t st) {
    const long rs = 3L * H * Dh;
    GemmP p;
    p.A = probs; p.Bm = reinterpret_cast<const T*>(qkv) + 2L * H * Dh; p.C = out;
    p.M = N; p.N = Dh; p.K = N;
    p.am = N; p.ak = 1; p.bk = rs; p.bn = 1; p.cm = (long)H * Dh; p.cn = 1;
    p.a_b1 = (long)H * N * N; p.a_b2 = (long)N * N; p.b_b1 = (long)N * rs; p.b_b2 = Dh;
    p.c_b1 = (long)N * H * Dh; p.c_b2 = Dh; p.nb2 = H; p.alpha = 1.f;
    return launch_gemm<float, T, T>(p, B * H, st);
}

template <typename T>
int attn_pv_bwd(const float* probs, const void* qkv, const void* dout, int B, int N, int H, int Dh, float* dprobs,
                void* dqkv, hipStream_t st) {
    const long rs = 3L * H * Dh, os = (long)H * Dh;
    GemmP p;
    // dV[key][d] = sum_q P[q][key] dO[q][d]
    p.A = probs; p.Bm = dout; p.C = reinterpret_cast<T*>(dqkv) + 2L * H * Dh;
    p.M = N; p.N = Dh; p.K = N;
    p.am = 1; p.ak = N; p.bk = os; p.bn = 1; p.cm = rs; p.cn = 1;
    p.a_b1 = (long)H * N * N; p.a_b2 = (long)N * N; p.b_b1 = (long)N * os; p.b_b2 = Dh;
    p.c_b1 = (long)N * rs; p.c_b2 = Dh; p.nb2 = H; p.alpha = 1.f;
    int rc = launch_gemm<float, T, T>(p, B * H, st);
    if (rc) return rc;
    // dP[q][key] = sum_d dO[q][d] V[key][d]
    p.A = dout; p.Bm = reinterpret_cast<const T*>(qkv) + 2L * H * Dh; p.C = dprobs;
    p.M = N; p.N = N; p.K = Dh;
    p.am = os; p.ak = 1; p.bk = 1; p.bn = rs; p.cm = N; p.cn = 1;
    p.a_b1 = (long)N * os; p.a_b2 = Dh; p.b_b1 = (long)N * rs; p.b_b2 = Dh;
    p.c_b1 = (long)H * N * N; p.c_b2 = (long)N * N;
    return launch_gemm<T, T, float>(p, B * H, st);
}

template <typename T>
int attn_scores_bwd(const float* probs, float* dprobs, const void* qkv, int B, int N, int H, int Dh, float scale,
                    void* dqkv, hipStream_t st) {
    const long rs = 3L * H * Dh;
    const long rows = (long)B * H * N;
    const dim3 sgrid((unsigned)((rows + 3) / 4));
    if (N == 2048) hipLaunchKernelGGL(row_softmax_bwd_reg_kernel<8>, sgrid, dim3(256), 0, st, probs, dprobs, rows, scale);
    else if (N == 1024) hipLaunchKernelGGL(row_softmax_bwd_reg_kernel<4>, sgrid, dim3(256), 0, st, probs, dprobs, rows, scale);
    else if (N == 512) hipLaunchKernelGGL(row_softmax_bwd_reg_kernel<2>, sgrid, dim3(256), 0, st, probs, dprobs, rows, scale);
    else if (N == 256) hipLaunchKernelGGL(row_softmax_bwd_reg_kernel<1>, sgrid, dim3(256), 0, st, probs, dprobs, rows, scale);
    else hipLaunchKernelGGL(row_softmax_bwd_kernel, sgrid, dim3(256), 0, st, probs, dprobs, rows, N, scale);
    LAUNCH_RET();
    GemmP p;
    // dQ[q][d] = sum_key dS[q][key] K[key][d]
    p.A = dprobs; p.Bm = reinterpret_cast<const T*>(qkv) + (long)H * Dh; p.C = dqkv;
    p.M = N; p.N = Dh; p.K = N;
    p.am = N; p.ak = 1; p.bk = rs; p.bn = 1; p.cm = rs; p.cn = 1;
    p.a_b1 = (long)H * N * N; p.a_b2 = (long)N * N; p.b_b1 = (long)N * rs; p.b_b2 = Dh;
    p.c_b1 = (long)N * rs; p.c_b2 = Dh; p.nb2 = H; p.alpha = 1.f;
    int rc = launch_gemm<float, T, T>(p, B * H, st);
    if (rc) return rc;
    // dK[key][d] = sum_q dS[q][key] Q[q][d]
    p.A = dprobs; p.Bm = qkv; p.C = reinterpret_cast<T*>(dqkv) + (long)H * Dh;
    p.am = 1; p.ak = N;
    return launch_gemm<float, T, T>(p, B * H, st);
}

}  // namespace

extern "C" int passion_attn_scores_fwd(const void* qkv, int dtype, int B, int N, int H, int Dh, float scale, float* probs,
                                       void* stream) {
    if (dtype == PASSION_F32) return attn_scores<float>(qkv, B, N, H, Dh, scale, probs, (hipStream_t)stream);
    if (dtype == PASSION_BF16) return attn_scores<bf16_t>(qkv, B, N, H, Dh, scale, probs, (hipStream_t)stream);
    return -1007;
}
extern "C" int passion_attn_scores_masked_fwd(const void* qkv, int dtype, int B, int N, int H, int Dh, float scale, const uint8_t* allow, int G,
                                              float* probs, void* stream) {
    if (!allow || G <= 0 || N % G) return -1003;
    if (dtype == PASSION_F32) return attn_scores<float>(qkv, B, N, H, Dh, scale, probs, (hipStream_t)stream, allow, G);
    if (dtype == PASSION_BF16) return attn_scores<bf16_t>(qkv, B, N, H, Dh, scale, probs, (hipStream_t)stream, allow, G);
    return -1007;
}
extern "C" int passion_attn_pv_fwd(const float* probs, const void* qkv, int dtype, int B, int N, int H, int Dh, void* out,
                                   void* stream) {
    if (dtype == PASSION_F32) return attn_pv<float>(probs, qkv, B, N, H, Dh, out, (hipStream_t)stream);
    if (dtype == PASSION_BF16) return attn_pv<bf16_t>(probs, qkv, B, N, H, Dh, out, (hipStream_t)stream);
    return -1007;
}
extern "C" int passion_attn_pv_bwd(const float* probs, const void* qkv, const void* dout, int dtype, int B, int N, int H,
                                   int Dh, float* dprobs, void* dqkv, void* stream) {
    if (dtype == PASSION_F32) return attn_pv_bwd<float>(probs, qkv, dout, B, N, H, Dh, dprobs, dqkv, (hipStream_t)stream);
    if (dtype == PASSION_BF16) return attn_pv_bwd<bf16_t>(probs, qkv, dout, B, N, H, Dh, dprobs, dqkv, (hipStream_t)stream);
    return -1007;
}
extern "C" int passion_attn_scores_bwd(const float* probs, float* dprobs, const void* qkv, int dtype, int B, int N, int H,
                                       int Dh, float scale, void* dqkv, void* stream) {
    if (dtype == PASSION_F32) return attn_scores_bwd<float>(probs, dprobs, qkv, B, N, H, Dh, scale, dqkv, (hipStream_t)stream);
    if (dtype == PASSION_BF16) return attn_scores_bwd<bf16_t>(probs, dprobs, qkv, B, N, H, Dh, scale, dqkv, (hipStream_t)stream);
    return -1007;
}

// ---------------------------------------------------------------------------------------------------------------
extern const unsigned* g_passion_step_words;      // api.cpp
// Attention dropout, mask regenerated from (seed, index): torch's rand_like / >= / two multiplies were four passes over
// the (B, H, N, N) f32 probabilities forward (168 M elements for the inter-modal transformer with the five PASSION
// passes) plus a stored mask; this is one pass each way.  splitmix64 of (seed, index / 4) yields four 16-bit uniforms.
namespace {
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long n4, long n, unsigned thr,
                                                      float inv_keep, unsigned long long seed, const unsigned* soff) {
    if (soff) seed += (unsigned long long)soff[0] | ((unsigned long long)soff[1] << 32);     // replayed HIP graph: passion_set_step_words
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        unsigned long long z = (unsigned long long)i + seed * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        const long e = i * 4;
        if (e + 3 < n) {
            f32x4 v = *reinterpret_cast<const f32x4*>(x + e);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (((unsigned)(z >> (16 * j)) & 0xffffu) >= thr) ? v[j] * inv_keep : 0.f;
            *reinterpret_cast<f32x4*>(y + e) = v;
        } else {
            for (int j = 0; j < 4 && e + j < n; ++j)
                y[e + j] = (((unsigned)(z >> (16 * j)) & 0xffffu) >= thr) ? x[e + j] * inv_keep : 0.f;
        }
    }
}
}  // namespace

extern "C" int passion_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, void* stream) {
    if (!(p >= 0.f && p < 1.f) || n < 0) return -1003;
    if (n == 0) return 0;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) return -1005;
    const long n4 = (n + 3) / 4;
    const unsigned thr = (unsigned)(p * 65536.0f + 0.5f);          // keep when the 16-bit uniform >= p * 2^16
    long gx = (n4 + 255) / 256;
    if (gx > 65536) gx = 65536;
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, x, y, n4, (long)n, thr, 1.0f / (1.0f - p),
                       (unsigned long long)seed, g_passion_step_words);
    LAUNCH_RET();
    return 0;
}
