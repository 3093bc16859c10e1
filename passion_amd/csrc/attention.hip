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

template <typename TA, typename TB, typename TC>
int launch_gemm(const GemmP& p, int nbatch, hipStream_t st) {
    dim3 grid(cdiv(p.N, GT), cdiv(p.M, GT), nbatch);
    hipLaunchKernelGGL((bgemm_kernel<TA, TB, TC>), grid, dim3(256), 0, st, p);
    LAUNCH_RET();
    return 0;
}

// one wavefront per row, in place
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

template <typename T>
int attn_scores(const void* qkv, int B, int N, int H, int Dh, float scale, float* probs, hipStream_t st) {
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
    hipLaunchKernelGGL(row_softmax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, probs, rows, N);
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
    hipLaunchKernelGGL(row_softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, probs, dprobs, rows,
                       N, scale);
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
