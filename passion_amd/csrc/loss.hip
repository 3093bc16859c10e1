// Class-axis softmax, layout conversion and the segmentation / self-distillation losses.
//   softmax                         <- nn.Softmax(dim=1) / F.softmax(x/T, dim=1)   (mmformer.py:91,543,635; criterions.py:93-94)
//   wce_dice                        <- softmax_weighted_loss_bs + dice_loss_bs     (criterions.py:59-76, 25-38)
//   kl                              <- temp_kl_loss_bs                             (criterions.py:92-103)
//   proto                           <- prototype_passion_loss_bs                   (criterions.py:144-180)
// All are HBM-bound streaming reductions over D*H*W: one pass, 16-byte loads per voxel, per-thread
// register partials, wavefront shuffle reduction, one atomic per workgroup per scalar.
#include <cstdlib>
#include "common.h"
#include "conv_common.h"
#include "passion_hip.h"

namespace {

constexpr int NC = 4;  // classes

// ---------------------------------------------------------------- x k trilinear sampling of a coarse 4-class tensor (align_corners=True)
// The deep-supervision terms (reference mmformer.py:366-370, 540-544; criterions.py:59-76, 92-103) score nn.Upsample(x k) of a
// coarse probability map against the full-size label.  The kernels below read the COARSE map (cache resident: at most 64^3 x 16
// bytes) and blend the eight neighbours in registers -- same source index arithmetic and association as resample.hip's forward
// -- so the x k map is never written or read (k = 2 .. 16: 33 MB per map and sample at 128^3, forward and backward).
struct UpG { int D, H, W, k, Do, Ho, Wo; float sd, sh, sw, inv_wo, inv_ho; };      // coarse dims, factor, fine dims, axis scales
struct USrc { int i0, i1; float l0, l1; };
__device__ __forceinline__ USrc u_src(int o, float scale, int in) {
    USrc s;
    const float r = scale * (float)o;
    s.i0 = (int)r;
    if (s.i0 > in - 1) s.i0 = in - 1;
    s.i1 = s.i0 + (s.i0 < in - 1 ? 1 : 0);
    s.l1 = r - (float)s.i0;
    s.l0 = 1.f - s.l1;
    return s;
}
__device__ __forceinline__ float u_weight(int o, int i, float scale, int in) {      // weight with which output o reads input i
    const USrc s = u_src(o, scale, in);
    float w = 0.f;
    if (s.i0 == i) w += s.l0;
    if (s.i1 == i) w += s.l1;
    return w;
}
__device__ __forceinline__ void u_range(int i, float scale, int out, int& lo, int& hi) {
    if (scale <= 0.f) { lo = 0; hi = out - 1; return; }
    lo = (int)floorf((float)(i - 1) / scale) - 1;
    hi = (int)ceilf((float)(i + 1) / scale) + 1;
    if (lo < 0) lo = 0;
    if (hi > out - 1) hi = out - 1;
}
// fine voxel index -> (od, oh, ow); fine volumes here have < 2^23 voxels (checked by the launchers): (v + 0.5) / n through the reciprocal is exact
__device__ __forceinline__ void u_split(const UpG& g, long v, int& od, int& oh, int& ow) {
    const int t = (int)(((float)v + 0.5f) * g.inv_wo);
    ow = (int)v - t * g.Wo;
    od = (int)(((float)t + 0.5f) * g.inv_ho);
    oh = t - od * g.Ho;
}
// value at fine voxel (od, oh, ow) of sample base xb: ((w-blend) h-blend) d-blend, as upsample_fwd_kernel
__device__ __forceinline__ f32x4 u_sample(const float* __restrict__ xb, const UpG& g, const USrc& a, const USrc& e, const USrc& f) {
    const int di[2] = {a.i0, a.i1}, hi[2] = {e.i0, e.i1};
    const float dl[2] = {a.l0, a.l1}, hl[2] = {e.l0, e.l1};
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        f32x4 rh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float* row = xb + (((long)di[p] * g.H + hi[q]) * g.W) * 4;
            const f32x4 t0 = load4(row + f.i0 * 4), t1 = load4(row + f.i1 * 4);
            rh += hl[q] * (f.l0 * t0 + f.l1 * t1);
        }
        r += dl[p] * rh;
    }
    return r;
}

// ---------------------------------------------------------------- softmax
__global__ void softmax_fwd_kernel(const float* __restrict__ lg, long lsb, long lsv, long lsc, float* __restrict__ pr,
                                   long psb, long psv, long psc, int B, long V, float inv_temp) {
    const long total = (long)B * V;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / V, v = i % V;
        float z[NC], m = -INFINITY;
        if (lsc == 1) {
            f32x4 t = load4(lg + b * lsb + v * lsv);
#pragma unroll
            for (int c = 0; c < NC; ++c) z[c] = t[c] * inv_temp;
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c) z[c] = lg[b * lsb + v * lsv + c * lsc] * inv_temp;
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) m = fmaxf(m, z[c]);
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) { z[c] = expf(z[c] - m); s += z[c]; }
        const float inv = 1.f / s;
        if (psc == 1) {
            f32x4 o = {z[0] * inv, z[1] * inv, z[2] * inv, z[3] * inv};
            store4(pr + b * psb + v * psv, o);
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c) pr[b * psb + v * psv + c * psc] = z[c] * inv;
        }
    }
}

__global__ void softmax_bwd_kernel(const float* __restrict__ pr, long psb, long psv, long psc,
                                   const float* __restrict__ dp, long gsb, long gsv, long gsc, float* __restrict__ dl,
                                   long lsb, long lsv, long lsc, int B, long V, float inv_temp) {
    const long total = (long)B * V;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / V, v = i % V;
        float p[NC], g[NC], dot = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            p[c] = pr[b * psb + v * psv + c * psc];
            g[c] = dp[b * gsb + v * gsv + c * gsc];
            dot += p[c] * g[c];
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) dl[b * lsb + v * lsv + c * lsc] = p[c] * (g[c] - dot) * inv_temp;
    }
}

template <typename TS>
__global__ void target_to_cl_kernel(const TS* __restrict__ t, float* __restrict__ out, int B, long V) {
    const long total = (long)B * V;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / V, v = i % V;
        f32x4 o;
#pragma unroll
        for (int c = 0; c < NC; ++c) o[c] = (float)t[(b * NC + c) * V + v];
        store4(out + i * 4, o);
    }
}

template <typename TS, typename TD>
__global__ void copy_strided_kernel(const TS* __restrict__ s, long ssb, long ssv, long ssc, TD* __restrict__ d,
                                    long dsb, long dsv, long dsc, int B, long V, int C) {
    const long total = (long)B * V * C;
    // iterate with the destination-contiguous index fastest
    const bool c_fast = (dsc == 1);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long b, v; int c;
        if (c_fast) { c = (int)(i % C); v = (i / C) % V; b = i / ((long)C * V); }
        else { v = i % V; c = (int)((i / V) % C); b = i / ((long)C * V); }
        d[b * dsb + v * dsv + c * dsc] = from_f32<TD>(to_f32(s[b * ssb + v * ssv + c * ssc]));
    }
}

// ---------------------------------------------------------------- deterministic grid reductions
// The per-sample sums of the loss kernels used to end with one float atomicAdd per block: the order of those adds changes run
// to run, the sums differ in their last bit, and a bf16 rounding downstream then flips -- gradients did not repeat.  Now every
// block stores its J partial sums in its own slot (`part`: [gridDim.y][gridDim.x][J] floats), takes a ticket from a device
// counter, and the block that arrives LAST adds the slots in block order: the same additions in the same order every run.
// Returns true in that last block only, with column j's total in thread j (j < J <= 128): all 256 threads of the last block
// share the slots (column j x slot group) with eight loads in flight each -- a serial walk over 512 slots would cost more than
// the reduction itself.  The counter is left at zero again for the next launch on the stream.
template <int J>
__device__ __forceinline__ bool fold_blocks(float s, float* __restrict__ part, unsigned* __restrict__ counter,
                                            float* __restrict__ /*red*/, float& total) {
    constexpr int JP = J <= 1 ? 1 : J <= 2 ? 2 : J <= 4 ? 4 : J <= 8 ? 8 : J <= 16 ? 16 : J <= 32 ? 32 : J <= 64 ? 64 : 128;
    constexpr int G = 256 / JP;                             // slot groups: thread (gq, j) adds slots gq, gq + G, ... of column j
    static_assert(J <= 128, "fold_blocks: at most 128 columns");
    __shared__ int last;
    __shared__ float fred[G * JP];
    const int nb = gridDim.x, b = blockIdx.y, tid = threadIdx.x;
    if (tid < J) st_agent(&part[((long)b * nb + blockIdx.x) * J + tid], s);      // agent-scope atomics, no fence: see common.h
    stores_done();
    __syncthreads();                                        // the slot is stored (vmcnt 0) before the ticket is taken
    if (tid == 0) last = atomicAdd(&counter[b], 1u) == (unsigned)(nb - 1);
    __syncthreads();
    if (!last) return false;
    acquire_agent();
    const int j = tid % JP, gq = tid / JP;
    float a = 0.f;
    if (j < J) {
        const float* col = part + (long)b * nb * J + j;
        int r = gq;
        constexpr int U = JP >= 64 ? 16 : 8;                // independent loads in flight, added in slot order
        for (; r + (U - 1) * G < nb; r += U * G) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = ld_agent(&col[(long)(r + u * G) * J]);
#pragma unroll
            for (int u = 0; u < U; ++u) a += v[u];
        }
        for (; r < nb; r += G) a += ld_agent(&col[(long)r * J]);
    }
    fred[gq * JP + j] = a;
    __syncthreads();
    total = 0.f;
    if (tid < J) {
#pragma unroll 8
        for (int q = 0; q < G; ++q) total += fred[q * JP + tid];
    }
    if (tid == 0) __hip_atomic_store(&counter[b], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}
constexpr int RED_BLOCKS_MAX = 1024;

// ---------------------------------------------------------------- weighted CE + Dice
// sums[b][0..3]=num_c, [4..7]=l_c (sum p), [8..11]=r_c (sum t), [12..15]=sum t_c*log(clamp p_c)
// the class probabilities of one voxel from its logits: the arithmetic of softmax_fwd_kernel, operation for operation (a fused pass
// sees bit for bit the probabilities the two-pass form stores)
__device__ __forceinline__ void softmax4(const f32x4 t, float inv_temp, float (&p)[NC]) {
    float z[NC], m = -INFINITY;
#pragma unroll
    for (int c = 0; c < NC; ++c) z[c] = t[c] * inv_temp;
#pragma unroll
    for (int c = 0; c < NC; ++c) m = fmaxf(m, z[c]);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) { z[c] = expf(z[c] - m); s += z[c]; }
    const float inv = 1.f / s;
#pragma unroll
    for (int c = 0; c < NC; ++c) p[c] = z[c] * inv;
}

// LG: `pr` holds channels-last LOGITS (psc == 1) and the probabilities are formed here (softmax4, inv_temp) instead of being read:
// decoder_sep's 4 B predictions never exist in memory (passion_softmax_wce_dice_fwd)
template <bool LG = false>
__global__ __launch_bounds__(256) void wce_dice_sums_kernel(const float* __restrict__ pr, long psb, long psv, long psc,
                                                            const float* __restrict__ tg, long V,
                                                            float* __restrict__ sums, float* __restrict__ part,
                                                            unsigned* __restrict__ counter, float* __restrict__ out, int Bt, UpG ug,
                                                            float inv_temp = 1.f) {
    __shared__ float red[4][16];
    const int b = blockIdx.y, bt = b % Bt;              // target of sample b (several predictions per target: Bt < B)
    float a[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = 0.f;
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (long)gridDim.x * blockDim.x) {
        float p[NC];
        if (LG) {
            softmax4(load4(pr + b * psb + v * psv), inv_temp, p);
        } else
        if (ug.k > 1) {                                     // pr: the coarse map [B][D][H][W][4]; V: fine voxels
            int od, oh, ow;
            u_split(ug, v, od, oh, ow);
            const f32x4 t = u_sample(pr + (long)b * ug.D * ug.H * ug.W * 4, ug, u_src(od, ug.sd, ug.D), u_src(oh, ug.sh, ug.H), u_src(ow, ug.sw, ug.W));
#pragma unroll
            for (int c = 0; c < NC; ++c) p[c] = t[c];
        } else if (psc == 1) {
            f32x4 t = load4(pr + b * psb + v * psv);
#pragma unroll
            for (int c = 0; c < NC; ++c) p[c] = t[c];
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c) p[c] = pr[b * psb + v * psv + c * psc];
        }
        const f32x4 t = load4(tg + ((long)bt * V + v) * 4);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            a[c] += p[c] * t[c];
            a[4 + c] += p[c];
            a[8 + c] += t[c];
            a[12 + c] += t[c] * logf(fminf(fmaxf(p[c], 0.005f), 1.0f));
        }
    }
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float s = wave_sum(a[j]);
        if (lane == 0) red[wv][j] = s;
    }
    __syncthreads();
    float s_blk = 0.f, tot;
    if (threadIdx.x < 16) s_blk = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (!fold_blocks<16>(s_blk, part, counter, &red[0][0], tot)) return;
    // last block: the sixteen totals (kept for the backward / the depth-sharded all-reduce) and the two loss scalars
    if (threadIdx.x < 16) { sums[b * 16 + threadIdx.x] = tot; red[0][threadIdx.x] = tot; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const float* s = &red[0][0];
    const float rt = s[8] + s[9] + s[10] + s[11];
    float ce = 0.f, dice = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const float w = 1.0f - s[8 + c] / rt;
        ce += -w * s[12 + c];
        dice += 2.0f * s[c] / (s[4 + c] + s[8 + c] + 1e-7f);
    }
    out[b * 2 + 0] = ce / (float)V;
    out[b * 2 + 1] = 1.0f - dice / (float)NC;
}

__global__ void wce_dice_bwd_kernel(const float* __restrict__ pr, long psb, long psv, long psc,
                                    const float* __restrict__ tg, const float* __restrict__ sums,
                                    const float* __restrict__ g, float* __restrict__ dp, long V, int Bt) {
    const int b = blockIdx.y, bt = b % Bt;
    const float* s = sums + b * 16;
    const float rt = s[8] + s[9] + s[10] + s[11];
    const float g_ce = g[b * 2 + 0], g_dice = g[b * 2 + 1];
    float kce[NC], kd1[NC], kd2[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const float w = 1.0f - s[8 + c] / rt;
        const float den = s[4 + c] + s[8 + c] + 1e-7f;
        kce[c] = -g_ce * w / (float)V;                 // * t / p (inside the clamp window)
        kd1[c] = -g_dice * 2.0f / ((float)NC * den);   // * t
        kd2[c] = g_dice * 2.0f * s[c] / ((float)NC * den * den);
    }
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (long)gridDim.x * blockDim.x) {
        const f32x4 t = load4(tg + ((long)bt * V + v) * 4);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float p = pr[b * psb + v * psv + c * psc];
            float d = kd1[c] * t[c] + kd2[c];
            if (p >= 0.005f && p <= 1.0f) d += kce[c] * t[c] / p;
            dp[b * psb + v * psv + c * psc] = d;
        }
    }
}

// d(loss)/d(logits) in one pass: the probabilities from the logits again (softmax4), the gradient wrt them as wce_dice_bwd_kernel forms
// it, through the softmax Jacobian as softmax_bwd_kernel applies it -- the same operations in the same order, without the two f32
// tensors (probabilities, their gradient) in between
__global__ __launch_bounds__(256) void softmax_wce_dice_bwd_kernel(const float* __restrict__ lg, const float* __restrict__ tg,
                                                                   const float* __restrict__ sums, const float* __restrict__ g,
                                                                   float* __restrict__ dl, long V, int Bt, float inv_temp) {
    const int b = blockIdx.y, bt = b % Bt;
    const float* s = sums + b * 16;
    const float rt = s[8] + s[9] + s[10] + s[11];
    const float g_ce = g[b * 2 + 0], g_dice = g[b * 2 + 1];
    float kce[NC], kd1[NC], kd2[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const float w = 1.0f - s[8 + c] / rt;
        const float den = s[4 + c] + s[8 + c] + 1e-7f;
        kce[c] = -g_ce * w / (float)V;
        kd1[c] = -g_dice * 2.0f / ((float)NC * den);
        kd2[c] = g_dice * 2.0f * s[c] / ((float)NC * den * den);
    }
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (long)gridDim.x * blockDim.x) {
        const f32x4 t = load4(tg + ((long)bt * V + v) * 4);
        float p[NC], d[NC], dot = 0.f;
        softmax4(load4(lg + ((long)b * V + v) * 4), inv_temp, p);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float dc = kd1[c] * t[c] + kd2[c];
            if (p[c] >= 0.005f && p[c] <= 1.0f) dc += kce[c] * t[c] / p[c];
            d[c] = dc;
            dot += p[c] * dc;
        }
        f32x4 o;
#pragma unroll
        for (int c = 0; c < NC; ++c) o[c] = p[c] * (d[c] - dot) * inv_temp;
        store4(dl + ((long)b * V + v) * 4, o);
    }
}

// ---------------------------------------------------------------- temperature KL
// Bt: samples of the teacher tensor; student sample b is held against teacher sample b % Bt (the four PASSION students batched
// against one teacher)
__global__ __launch_bounds__(256) void kl_fwd_kernel(const float* __restrict__ ps, const float* __restrict__ pt, long V,
                                                     float k, float* __restrict__ out, float* __restrict__ part,
                                                     unsigned* __restrict__ counter, int Bt, UpG ug) {
    __shared__ float red[4];
    const int b = blockIdx.y, bt = b % Bt;
    float a = 0.f;
    const long vc = (long)ug.D * ug.H * ug.W * 4;           // ug.k > 1: ps / pt are the coarse maps, V the fine voxels
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (long)gridDim.x * blockDim.x) {
        f32x4 s, t;
        if (ug.k > 1) {
            int od, oh, ow;
            u_split(ug, v, od, oh, ow);
            const USrc sa = u_src(od, ug.sd, ug.D), se = u_src(oh, ug.sh, ug.H), sf = u_src(ow, ug.sw, ug.W);
            s = u_sample(ps + b * vc, ug, sa, se, sf);
            t = u_sample(pt + bt * vc, ug, sa, se, sf);
        } else {
            s = load4(ps + ((long)b * V + v) * 4);
            t = load4(pt + ((long)bt * V + v) * 4);
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float sc = fminf(fmaxf(s[c], 0.005f), 1.0f), tc = fminf(fmaxf(t[c], 0.005f), 1.0f);
            a += tc * (logf(tc) - logf(sc));
        }
    }
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    float s_blk = 0.f, tot;
    if (threadIdx.x == 0) s_blk = red[0] + red[1] + red[2] + red[3];
    if (fold_blocks<1>(s_blk, part, counter, red, tot) && threadIdx.x == 0) out[b] = k * tot;
}

__global__ void kl_bwd_kernel(const float* __restrict__ ps, const float* __restrict__ pt, const float* __restrict__ g,
                              float* __restrict__ dps, long V, float k, int Bt) {
    const int b = blockIdx.y, bt = b % Bt;
    const float gk = g[b] * k;
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (long)gridDim.x * blockDim.x) {
        const f32x4 s = load4(ps + ((long)b * V + v) * 4), t = load4(pt + ((long)bt * V + v) * 4);
        f32x4 d;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float tc = fminf(fmaxf(t[c], 0.005f), 1.0f);
            d[c] = (s[c] >= 0.005f && s[c] <= 1.0f) ? -gk * tc / s[c] : 0.f;
        }
        store4(dps + ((long)b * V + v) * 4, d);
    }
}


// Backward of the x k terms: the adjoint of the interpolation is separable (A_d x A_h x A_w, resample.hip).  Its first pass -- the W
// axis, the contiguous and largest reduction -- is fused with the loss gradient: a thread owns (b, od, oh, iw), walks the fine ow
// that read coarse column iw, forms the fine probability (eight coarse loads, the D / H taps fixed for the thread), the
// gradient of the loss wrt it, and sums with the W weights.  tmp [B][Do][Ho][W][4] goes through the H and D axis passes.
// MODE 0: weighted CE + Dice (coef = the 16 sums of the forward, g = (B, 2)); MODE 1: temperature KL (pt = coarse teacher map, g = (B,)).
template <int MODE>
__global__ __launch_bounds__(256) void up_loss_bwdw_kernel(const float* __restrict__ pc, const float* __restrict__ pt,
                                                           const float* __restrict__ tg, const float* __restrict__ sums,
                                                           const float* __restrict__ g, float* __restrict__ tmp, UpG ug, int Bt,
                                                           float kk) {
    const int b = blockIdx.y, bt = b % Bt;
    const long V = (long)ug.Do * ug.Ho * ug.Wo, vc = (long)ug.D * ug.H * ug.W * 4;
    float kce[NC], kd1[NC], kd2[NC], gk = 0.f;
    if (MODE == 0) {
        const float* s = sums + b * 16;
        const float rt = s[8] + s[9] + s[10] + s[11];
        const float g_ce = g[b * 2 + 0], g_dice = g[b * 2 + 1];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float w = 1.0f - s[8 + c] / rt;
            const float den = s[4 + c] + s[8 + c] + 1e-7f;
            kce[c] = -g_ce * w / (float)V;
            kd1[c] = -g_dice * 2.0f / ((float)NC * den);
            kd2[c] = g_dice * 2.0f * s[c] / ((float)NC * den * den);
        }
    } else {
        gk = g[b] * kk;
    }
    const float* xb = pc + b * vc;
    const float* tb = pt + bt * vc;
    const long items = (long)ug.Do * ug.Ho * ug.W;
    const float inv_w = 1.0f / (float)ug.W;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (long)gridDim.x * blockDim.x) {
        const int r = (int)(((float)it + 0.5f) * inv_w), iw = (int)it - r * ug.W;       // r = od * Ho + oh (< 2^23 items)
        const int od = (int)(((float)r + 0.5f) * ug.inv_ho), oh = r - od * ug.Ho;
        const USrc a = u_src(od, ug.sd, ug.D), e = u_src(oh, ug.sh, ug.H);
        int lo, hi;
        u_range(iw, ug.sw, ug.Wo, lo, hi);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int ow = lo; ow <= hi; ++ow) {
            const float ww = u_weight(ow, iw, ug.sw, ug.W);
            if (ww == 0.f) continue;
            const USrc f = u_src(ow, ug.sw, ug.W);
            const f32x4 p = u_sample(xb, ug, a, e, f);
            const long v = (long)r * ug.Wo + ow;
            f32x4 d;
            if (MODE == 0) {
                const f32x4 t = load4(tg + ((long)bt * V + v) * 4);
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    float dc = kd1[c] * t[c] + kd2[c];
                    if (p[c] >= 0.005f && p[c] <= 1.0f) dc += kce[c] * t[c] / p[c];
                    d[c] = dc;
                }
            } else {
                const f32x4 t = u_sample(tb, ug, a, e, f);
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const float tc = fminf(fmaxf(t[c], 0.005f), 1.0f);
                    d[c] = (p[c] >= 0.005f && p[c] <= 1.0f) ? -gk * tc / p[c] : 0.f;
                }
            }
            acc += ww * d;
        }
        store4(tmp + ((long)b * items + it) * 4, acc);
    }
}

// The same pass with one thread per FINE voxel: a workgroup takes whole rows (od, oh), every thread forms the loss gradient of one
// fine voxel (eight coarse loads, one label load) into LDS, then the row's W coarse columns gather their fine columns from LDS -- in
// up_loss_bwdw_kernel's order and with its weights, so the result is bit for bit the same.  That kernel walks a thread's 2k + 1 fine
// columns one after the other (k = 16: 131 072 threads with 33 dependent iterations each, 100 us for 33 MB of labels).
template <int MODE>
__global__ __launch_bounds__(256) void up_loss_bwdw_rows_kernel(const float* __restrict__ pc, const float* __restrict__ pt,
                                                                const float* __restrict__ tg, const float* __restrict__ sums,
                                                                const float* __restrict__ g, float* __restrict__ tmp, UpG ug, int Bt,
                                                                float kk, int rpb) {
    extern __shared__ __attribute__((aligned(16))) float drow[];          // [rpb][Wo][4]
    const int b = blockIdx.y, bt = b % Bt;
    const long V = (long)ug.Do * ug.Ho * ug.Wo, vc = (long)ug.D * ug.H * ug.W * 4;
    float kce[NC], kd1[NC], kd2[NC], gk = 0.f;
    if (MODE == 0) {
        const float* s = sums + b * 16;
        const float rt = s[8] + s[9] + s[10] + s[11];
        const float g_ce = g[b * 2 + 0], g_dice = g[b * 2 + 1];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float w = 1.0f - s[8 + c] / rt;
            const float den = s[4 + c] + s[8 + c] + 1e-7f;
            kce[c] = -g_ce * w / (float)V;
            kd1[c] = -g_dice * 2.0f / ((float)NC * den);
            kd2[c] = g_dice * 2.0f * s[c] / ((float)NC * den * den);
        }
    } else {
        gk = g[b] * kk;
    }
    const float* xb = pc + b * vc;
    const float* tb = pt + bt * vc;
    const int nrows = ug.Do * ug.Ho, tid = threadIdx.x;
    for (int r0 = blockIdx.x * rpb; r0 < nrows; r0 += gridDim.x * rpb) {
        const int nr = nrows - r0 < rpb ? nrows - r0 : rpb;
        __syncthreads();                                               // the previous rows' readers are done
        for (int i = tid; i < nr * ug.Wo; i += 256) {
            const int rr = i / ug.Wo, ow = i - rr * ug.Wo, r = r0 + rr;
            const int od = (int)(((float)r + 0.5f) * ug.inv_ho), oh = r - od * ug.Ho;
            const USrc a = u_src(od, ug.sd, ug.D), e = u_src(oh, ug.sh, ug.H), f = u_src(ow, ug.sw, ug.W);
            const f32x4 p = u_sample(xb, ug, a, e, f);
            const long v = (long)r * ug.Wo + ow;
            f32x4 d;
            if (MODE == 0) {
                const f32x4 t = load4(tg + ((long)bt * V + v) * 4);
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    float dc = kd1[c] * t[c] + kd2[c];
                    if (p[c] >= 0.005f && p[c] <= 1.0f) dc += kce[c] * t[c] / p[c];
                    d[c] = dc;
                }
            } else {
                const f32x4 t = u_sample(tb, ug, a, e, f);
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const float tc = fminf(fmaxf(t[c], 0.005f), 1.0f);
                    d[c] = (p[c] >= 0.005f && p[c] <= 1.0f) ? -gk * tc / p[c] : 0.f;
                }
            }
            *reinterpret_cast<f32x4*>(drow + (long)i * 4) = d;
        }
        __syncthreads();
        for (int i = tid; i < nr * ug.W; i += 256) {
            const int rr = i / ug.W, iw = i - rr * ug.W;
            int lo, hi;
            u_range(iw, ug.sw, ug.Wo, lo, hi);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int ow = lo; ow <= hi; ++ow) {
                const float ww = u_weight(ow, iw, ug.sw, ug.W);
                if (ww == 0.f) continue;
                acc += ww * *reinterpret_cast<const f32x4*>(drow + ((long)rr * ug.Wo + ow) * 4);
            }
            store4(tmp + ((long)b * nrows * ug.W + (long)(r0 + rr) * ug.W + iw) * 4, acc);
        }
    }
}

// ---------------------------------------------------------------- prototype loss (Cf = 8 feature channels)
constexpr int CF = 8;
// work layout (floats), per batch b: PW = 4 + 2*4*CF + 4*CF + 8
//   [0..3] count_c | [4 .. 4+4CF) sum_s[c][ch] | [.. +4CF) sum_t[c][ch] | [.. +4CF) Gp[c][ch] | loss, dist, pad..
// followed (after B*PW) by 4 floats: class validity flags, then 1 float: number of valid classes.
constexpr int PW = 4 + 3 * 4 * CF + 8;
__host__ __device__ inline int pw_sums(int c, int ch) { return 4 + c * CF + ch; }
__host__ __device__ inline int pw_sumt(int c, int ch) { return 4 + 4 * CF + c * CF + ch; }
__host__ __device__ inline int pw_gp(int c, int ch) { return 4 + 8 * CF + c * CF + ch; }
constexpr int PW_LOSS = 4 + 12 * CF, PW_DIST = PW_LOSS + 1;

// features have cf = 4 or 8 channels; the arithmetic always runs on 8 lanes with the tail zero-filled
// (zero channels change no dot product, norm or prototype)
template <typename T> __device__ __forceinline__ void load8(const T* p, float* o, int cf) {
    f32x4 a = load4(p), b = {0.f, 0.f, 0.f, 0.f};
    if (cf == 8) b = load4(p + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[j] = a[j]; o[4 + j] = b[j]; }
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float* o, int cf) {
    f32x4 a = {o[0], o[1], o[2], o[3]}, b = {o[4], o[5], o[6], o[7]};
    store4(p, a);
    if (cf == 8) store4(p + 4, b);
}

template <typename T>
__global__ __launch_bounds__(256) void proto_sums_kernel(const T* __restrict__ fs, const T* __restrict__ ft,
                                                         const float* __restrict__ tg, long V, float* __restrict__ work, int cf,
                                                         float* __restrict__ part, unsigned* __restrict__ counter, int Bt) {
    __shared__ float red[4][4 + 8 * CF];
    const int b = blockIdx.y, bt = b % Bt;      // teacher features / target of student sample b
    float cnt[NC], ss[NC][CF], st[NC][CF];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        cnt[c] = 0.f;
#pragma unroll
        for (int h = 0; h < CF; ++h) { ss[c][h] = 0.f; st[c][h] = 0.f; }
    }
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (long)gridDim.x * blockDim.x) {
        float a[CF], e[CF];
        load8(fs + ((long)b * V + v) * cf, a, cf);
        load8(ft + ((long)bt * V + v) * cf, e, cf);
        const f32x4 t = load4(tg + ((long)bt * V + v) * 4);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            cnt[c] += t[c];
#pragma unroll
            for (int h = 0; h < CF; ++h) { ss[c][h] += a[h] * t[c]; st[c][h] += e[h] * t[c]; }
        }
    }
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        float s = wave_sum(cnt[c]);
        if (lane == 0) red[wv][c] = s;
#pragma unroll
        for (int h = 0; h < CF; ++h) {
            s = wave_sum(ss[c][h]);
            if (lane == 0) red[wv][4 + c * CF + h] = s;
            s = wave_sum(st[c][h]);
            if (lane == 0) red[wv][4 + 4 * CF + c * CF + h] = s;
        }
    }
    __syncthreads();
    const int j = threadIdx.x;
    float s_blk = 0.f, tot;
    if (j < 4 + 8 * CF) s_blk = red[0][j] + red[1][j] + red[2][j] + red[3][j];
    if (fold_blocks<4 + 8 * CF>(s_blk, part, counter, &red[0][0], tot) && j < 4 + 8 * CF) work[(long)b * PW + j] = tot;
}

// class c counts only if every sample of the (local) batch contains it (criterions.py:157)
__global__ void proto_valid_kernel(float* __restrict__ work, int B) {
    if (threadIdx.x < NC) {
        bool ok = true;
        for (int b = 0; b < B; ++b) ok = ok && (work[(long)b * PW + threadIdx.x] > 0.f);
        work[(long)B * PW + threadIdx.x] = ok ? 1.f : 0.f;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float n = 0.f;
        for (int c = 0; c < NC; ++c) n += work[(long)B * PW + c];
        work[(long)B * PW + 4] = n;
    }
}

struct Protos { float ps[NC][CF], pt[NC][CF], nps[NC], npt[NC], cnt[NC], valid[NC], nvalid; };

__device__ __forceinline__ void load_protos(const float* __restrict__ work, int b, int B, Protos& P) {
    const float eps = 1e-5f;
    P.nvalid = work[(long)B * PW + 4];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        P.cnt[c] = work[(long)b * PW + c];
        P.valid[c] = work[(long)B * PW + c];
        float a = 0.f, e = 0.f;
#pragma unroll
        for (int h = 0; h < CF; ++h) {
            P.ps[c][h] = work[(long)b * PW + pw_sums(c, h)] / (P.cnt[c] + eps);
            P.pt[c][h] = work[(long)b * PW + pw_sumt(c, h)] / (P.cnt[c] + eps);
            a += P.ps[c][h] * P.ps[c][h];
            e += P.pt[c][h] * P.pt[c][h];
        }
        P.nps[c] = sqrtf(a); P.npt[c] = sqrtf(e);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void proto_loss_kernel(const T* __restrict__ fs, const T* __restrict__ ft, long V,
                                                         int B, float* __restrict__ work, int cf,
                                                         float* __restrict__ part, unsigned* __restrict__ counter, int Bt) {
    __shared__ float red[4][2];
    const int b = blockIdx.y, bt = b % Bt;      // teacher features / target of student sample b
    const float eps = 1e-5f;
    Protos P;
    load_protos(work, b, B, P);
    float al = 0.f, ad = 0.f;
    float inp[NC], inpt[NC];                   // bf16 storage: the quotients by the prototype norms as products (as proto_bwd1_kernel)
#pragma unroll
    for (int c = 0; c < NC; ++c) { inp[c] = 1.f / fmaxf(P.nps[c], eps); inpt[c] = 1.f / fmaxf(P.npt[c], eps); }
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (long)gridDim.x * blockDim.x) {
        float a[CF], e[CF];
        load8(fs + ((long)b * V + v) * cf, a, cf);
        load8(ft + ((long)bt * V + v) * cf, e, cf);
        float na = 0.f, ne = 0.f;
#pragma unroll
        for (int h = 0; h < CF; ++h) { na += a[h] * a[h]; ne += e[h] * e[h]; }
        const float ia = 1.f / fmaxf(sqrtf(na), eps), ie = 1.f / fmaxf(sqrtf(ne), eps);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (P.valid[c] == 0.f) continue;
            float da = 0.f, de = 0.f;
#pragma unroll
            for (int h = 0; h < CF; ++h) { da += a[h] * P.ps[c][h]; de += e[h] * P.pt[c][h]; }
            const float sim_s = sizeof(T) == 4 ? da * ia / fmaxf(P.nps[c], eps) : da * (ia * inp[c]);
            const float sim_t = sizeof(T) == 4 ? de * ie / fmaxf(P.npt[c], eps) : de * ie * inpt[c];
            const float d = sim_s - sim_t;
            al += d * d;
            ad += sqrtf(d * d);
        }
    }
    al = wave_sum(al); ad = wave_sum(ad);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = al; red[threadIdx.x >> 6][1] = ad; }
    __syncthreads();
    float s_blk = 0.f, tot;
    if (threadIdx.x < 2) s_blk = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (fold_blocks<2>(s_blk, part, counter, &red[0][0], tot) && threadIdx.x < 2)
        work[(long)b * PW + PW_LOSS + threadIdx.x] = tot / (P.nvalid * (float)V);
}

__global__ void proto_out_kernel(const float* __restrict__ work, int B, float* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    out[b * 2 + 0] = work[(long)b * PW + PW_LOSS];
    out[b * 2 + 1] = work[(long)b * PW + PW_DIST];
}

// backward pass 1: direct term of d loss / d fs and the prototype-gradient sums Gp[c][ch]
template <typename T>
__global__ __launch_bounds__(256) void proto_bwd1_kernel(const T* __restrict__ fs, const T* __restrict__ ft,
                                                         const float* __restrict__ g, long V, int B,
                                                         float* __restrict__ work, T* __restrict__ dfs, int cf,
                                                         float* __restrict__ part, unsigned* __restrict__ counter, int Bt) {
    __shared__ float red[4][NC * CF];
    const int b = blockIdx.y, bt = b % Bt;      // teacher features / target of student sample b
    const float eps = 1e-5f;
    Protos P;
    load_protos(work, b, B, P);
    const float kk = g[b] * 2.0f / (P.nvalid * (float)V);
    float gp[NC][CF];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int h = 0; h < CF; ++h) gp[c][h] = 0.f;
    // per-class constants of this sample: every quotient by a prototype norm is a product from here on (the loop below had 82 f32
    // divisions per voxel, ~12 instructions each: 534 us for 340 MB at 4 x 128^3; three are left, those by the voxel's own norm: 373 us.
    // Tried on top and slower: prototypes in LDS + two voxels per iteration (439 us; the compiler fills whatever register budget the
    // occupancy target leaves and spills inside the loop when told to fit three waves))
    float inp[NC], inpt[NC], cpk[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const float Np = fmaxf(P.nps[c], eps);
        inp[c] = 1.f / Np;
        inpt[c] = 1.f / fmaxf(P.npt[c], eps);
        cpk[c] = (P.nps[c] > eps) ? 1.f / (P.nps[c] * Np) : 0.f;
    }
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (long)gridDim.x * blockDim.x) {
        float a[CF], e[CF], da_[CF];
        load8(fs + ((long)b * V + v) * cf, a, cf);
        load8(ft + ((long)bt * V + v) * cf, e, cf);
        float na = 0.f, ne = 0.f;
#pragma unroll
        for (int h = 0; h < CF; ++h) { na += a[h] * a[h]; ne += e[h] * e[h]; da_[h] = 0.f; }
        const float nra = sqrtf(na), nre = sqrtf(ne);
        const float Na = fmaxf(nra, eps), ia = 1.f / Na, ie = 1.f / fmaxf(nre, eps);
        const float cak = (nra > eps) ? 1.f / (nra * Na) : 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (P.valid[c] == 0.f) continue;
            float da = 0.f, de = 0.f;
#pragma unroll
            for (int h = 0; h < CF; ++h) { da += a[h] * P.ps[c][h]; de += e[h] * P.pt[c][h]; }
            // d sim / d fs = p/(Na Np) - [|fs|>eps] sim fs/(|fs| Na) ;  d sim / d p = fs/(Na Np) - [|p|>eps] sim p/(|p| Np)
            if (sizeof(T) == 4) {
                // f32 parity mode: the quotients as the forward forms them (its 20-step PASSION trajectory against the imported reference
                // sits at 9.5e-4 of a 1e-3 bound, and one ulp here moves a later step by 3e-4)
                const float Np = fmaxf(P.nps[c], eps);
                const float sim_s = da * ia / Np, sim_t = de * ie / fmaxf(P.npt[c], eps);
                const float gs = kk * (sim_s - sim_t);  // dL/dsim_s
                const float ca = (nra > eps) ? sim_s / (nra * Na) : 0.f;
                const float cp = (P.nps[c] > eps) ? sim_s / (P.nps[c] * Np) : 0.f;
#pragma unroll
                for (int h = 0; h < CF; ++h) {
                    da_[h] += gs * (P.ps[c][h] * ia / Np - ca * a[h]);
                    gp[c][h] += gs * (a[h] * ia / Np - cp * P.ps[c][h]);
                }
                continue;
            }
            const float iaNp = ia * inp[c];
            const float sim_s = da * iaNp, sim_t = de * ie * inpt[c];
            const float gs = kk * (sim_s - sim_t);  // dL/dsim_s
            const float w1 = gs * iaNp, wa = gs * (sim_s * cak), wp = gs * (sim_s * cpk[c]);
#pragma unroll
            for (int h = 0; h < CF; ++h) {
                da_[h] += w1 * P.ps[c][h] - wa * a[h];
                gp[c][h] += w1 * a[h] - wp * P.ps[c][h];
            }
        }
        store8(dfs + ((long)b * V + v) * cf, da_, cf);
    }
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int h = 0; h < CF; ++h) {
            const float s = wave_sum(gp[c][h]);
            if (lane == 0) red[wv][c * CF + h] = s;
        }
    __syncthreads();
    const int j = threadIdx.x;
    float s_blk = 0.f, tot;
    if (j < NC * CF) s_blk = red[0][j] + red[1][j] + red[2][j] + red[3][j];
    if (fold_blocks<NC * CF>(s_blk, part, counter, &red[0][0], tot) && j < NC * CF) work[(long)b * PW + 4 + 8 * CF + j] = tot;
}

// backward pass 2: prototypes are masked means of fs, so every voxel of class c receives Gp[c]/(count_c+eps)
template <typename T>
__global__ void proto_bwd2_kernel(const float* __restrict__ tg, long V, const float* __restrict__ work,
                                  T* __restrict__ dfs, int cf, int Bt) {
    const int b = blockIdx.y, bt = b % Bt;      // teacher features / target of student sample b
    const float eps = 1e-5f;
    float gpn[NC][CF];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int h = 0; h < CF; ++h) gpn[c][h] = work[(long)b * PW + pw_gp(c, h)] / (work[(long)b * PW + c] + eps);
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (long)gridDim.x * blockDim.x) {
        float d[CF];
        load8(dfs + ((long)b * V + v) * cf, d, cf);
        const f32x4 t = load4(tg + ((long)bt * V + v) * 4);
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int h = 0; h < CF; ++h) d[h] += t[c] * gpn[c][h];
        store8(dfs + ((long)b * V + v) * cf, d, cf);
    }
}


// ---------------------------------------------------------------- column sums (bias gradient of the token GEMMs)
// out[c] = sum_r g[r][c], g: [R][C] row-major (bf16 or f32), out f32.  Stands in for torch's `grad.sum(0)`: that reduce kernel
// (global reduce through a staging buffer) was measured NOT reproducible on this stack while a second stream keeps the GPU
// busy -- 20 % of 1000 column sums of a 1280 x 4096 bf16 matrix differed from the first, by up to the size of the sum itself
// (tools/linear_bias_determinism.py) -- and the encoders run on side streams next to the transformers' backward.
// One block: 128 columns (a wave reads 256 B of a bf16 row), its four waves take rows r = 4*by + w, + 4*gridDim.y, ...; the
// row slices of gridDim.y blocks are folded by the last block in slice order (same ticket scheme as fold_blocks).
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ g, long R, int C, float* __restrict__ out,
                                                     float* __restrict__ part, unsigned* __restrict__ counter) {
    __shared__ float red[4][128];
    __shared__ int last;
    // blockIdx.z: matrix of a batch (the grouped token GEMMs: one bias gradient per weight set)
    g += (long)blockIdx.z * R * C; out += (long)blockIdx.z * C;
    part += (long)blockIdx.z * gridDim.y * C; counter += blockIdx.z * gridDim.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int col = blockIdx.x * 128 + lane * 2;
    float a0 = 0.f, a1 = 0.f;
    if (col < C) {
        const long step = 4L * gridDim.y;
        long r = 4L * blockIdx.y + w;
        for (; r + 3 * step < R; r += 4 * step) {            // four rows in flight
            float v[4][2];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const T* q = g + (r + u * step) * C + col;
                v[u][0] = to_f32(q[0]); v[u][1] = to_f32(q[1]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { a0 += v[u][0]; a1 += v[u][1]; }
        }
        for (; r < R; r += step) { a0 += to_f32(g[r * C + col]); a1 += to_f32(g[r * C + col + 1]); }
    }
    red[w][lane * 2] = a0; red[w][lane * 2 + 1] = a1;
    __syncthreads();
    const int t = threadIdx.x, c = blockIdx.x * 128 + t;
    float s = 0.f;
    if (t < 128) s = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
    if (gridDim.y == 1) {
        if (t < 128 && c < C) out[c] = s;
        return;
    }
    if (t < 128 && c < C) st_agent(&part[(long)blockIdx.y * C + c], s);
    stores_done();
    __syncthreads();
    if (t == 0) last = atomicAdd(&counter[blockIdx.x], 1u) == gridDim.y - 1;
    __syncthreads();
    if (!last) return;
    acquire_agent();
    if (t < 128 && c < C) {
        float tot = 0.f;
        int y = 0;
        for (; y + 8 <= (int)gridDim.y; y += 8) {           // eight loads in flight, added in slice order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ld_agent(&part[(long)(y + u) * C + c]);
#pragma unroll
            for (int u = 0; u < 8; ++u) tot += v[u];
        }
        for (; y < (int)gridDim.y; ++y) tot += ld_agent(&part[(long)y * C + c]);
        out[c] = tot;
    }
    if (t == 0) __hip_atomic_store(&counter[blockIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
constexpr int COLSUM_SLICES = 32;

int g_red_blocks = []() { const char* e = getenv("PASSION_RED_BLOCKS"); return e && atoi(e) > 0 ? atoi(e) : 512; }();
inline int vgrid(long V) { return (int)hmin((V + 255) / 256, 2048); }
// reductions end with one atomic per block per scalar on a handful of addresses: same-address atomics serialise
// (~50-100 ns each), which argues for few blocks -- but 128 blocks left half of the 256 CUs idle and one workgroup per CU
// cannot hide its load latency (kl_fwd 46 us, proto_bwd1 231 us per launch at 128^3): 512 blocks measured +5 % on the whole
// PASSION step, 1024 the same (PASSION_RED_BLOCKS to override)
inline int vgrid_red(long V) { return (int)hmin((V + 256 * 4 - 1) / (256 * 4), hmin((long)g_red_blocks, (long)RED_BLOCKS_MAX)); }
// the prototype kernels pay per THREAD for their 2 x 32 prototype components (64 quotients each) and per BLOCK for 32 / 68 column
// sums (wave reductions, a slot, the last block's fold over all slots): at 4 x 128^3 proto_sums 177 / 113 / 84 us and proto_bwd1
// 408 / 373 / 297 us with 1024 / 512 / 256 blocks per sample (PASSION_RED_BLOCKS_PROTO to override)
// (128: proto_bwd1 265 us, proto_sums 108; 64: 379 / 175) -> 256 for the forward's two kernels, half of that for proto_bwd1
int g_red_blocks_proto = []() { const char* e = getenv("PASSION_RED_BLOCKS_PROTO"); return e && atoi(e) > 0 ? atoi(e) : 256; }();
inline int vgrid_red_proto(long V, int div = 1) {
    return (int)hmin((V + 256 * 4 - 1) / (256 * 4), hmin((long)hmax(1, g_red_blocks_proto / div), (long)RED_BLOCKS_MAX));
}
// scratch behind a reduction's own floats: B ticket counters (+ padding to 4), then B * RED_BLOCKS_MAX * J partial-sum slots
inline long red_extra_floats(int B, int J) { return (long)((B + 3) & ~3) + (long)B * RED_BLOCKS_MAX * J; }

}  // namespace


// ticket counters of the fold kernels: n zeroed words of the stream's library scratch (api.cpp); the winner leaves them at zero,
// kernels of one stream run one after the other -- no memset launch per call
static int tickets(hipStream_t st, size_t n, unsigned** out) {
    void* z = nullptr;
    const int rc = pconv::stream_scratch(st, n * sizeof(unsigned), 0, &z, nullptr);
    *out = static_cast<unsigned*>(z);
    return rc;
}

extern "C" int passion_softmax_fwd(const float* logits, int64_t lsb, int64_t lsv, int64_t lsc, float* probs, int64_t psb,
                                   int64_t psv, int64_t psc, int B, int64_t V, int C, float inv_temp, void* stream) {
    if (C != NC) return -1012;
    hipLaunchKernelGGL(softmax_fwd_kernel, dim3(vgrid((long)B * V)), dim3(256), 0, (hipStream_t)stream, logits, (long)lsb,
                       (long)lsv, (long)lsc, probs, (long)psb, (long)psv, (long)psc, B, (long)V, inv_temp);
    LAUNCH_RET();
    return 0;
}

extern "C" int passion_softmax_bwd(const float* probs, int64_t psb, int64_t psv, int64_t psc, const float* dprobs,
                                   int64_t gsb, int64_t gsv, int64_t gsc, float* dlogits, int64_t lsb, int64_t lsv,
                                   int64_t lsc, int B, int64_t V, int C, float inv_temp, void* stream) {
    if (C != NC) return -1012;
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3(vgrid((long)B * V)), dim3(256), 0, (hipStream_t)stream, probs, (long)psb,
                       (long)psv, (long)psc, dprobs, (long)gsb, (long)gsv, (long)gsc, dlogits, (long)lsb, (long)lsv,
                       (long)lsc, B, (long)V, inv_temp);
    LAUNCH_RET();
    return 0;
}

extern "C" int passion_target_to_cl(const void* target, int is_f64, float* out, int B, int64_t V, int C, void* stream) {
    if (C != NC) return -1012;
    if (is_f64)
        hipLaunchKernelGGL(target_to_cl_kernel<double>, dim3(vgrid((long)B * V)), dim3(256), 0, (hipStream_t)stream,
                           (const double*)target, out, B, (long)V);
    else
        hipLaunchKernelGGL(target_to_cl_kernel<float>, dim3(vgrid((long)B * V)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)target, out, B, (long)V);
    LAUNCH_RET();
    return 0;
}

extern "C" int passion_copy_strided(const void* src, int sdtype, int64_t ssb, int64_t ssv, int64_t ssc, void* dst,
                                    int ddtype, int64_t dsb, int64_t dsv, int64_t dsc, int B, int64_t V, int C,
                                    void* stream) {
    const int g = vgrid((long)B * V * C);
    hipStream_t st = (hipStream_t)stream;
#define CPY(TS, TD)                                                                                               \
    hipLaunchKernelGGL((copy_strided_kernel<TS, TD>), dim3(g), dim3(256), 0, st, (const TS*)src, (long)ssb, (long)ssv, \
                       (long)ssc, (TD*)dst, (long)dsb, (long)dsv, (long)dsc, B, (long)V, C)
    if (sdtype == PASSION_F32 && ddtype == PASSION_F32) CPY(float, float);
    else if (sdtype == PASSION_F32 && ddtype == PASSION_BF16) CPY(float, bf16_t);
    else if (sdtype == PASSION_BF16 && ddtype == PASSION_F32) CPY(bf16_t, float);
    else if (sdtype == PASSION_BF16 && ddtype == PASSION_BF16) CPY(bf16_t, bf16_t);
    else return -1007;
#undef CPY
    LAUNCH_RET();
    return 0;
}

extern "C" int passion_wce_dice_fwd(const float* probs, int64_t psb, int64_t psv, int64_t psc, const float* target,
                                    int B, int Bt, int64_t V, int C, float* sums, float* out, void* stream) {
    if (C != NC) return -1012;
    if (Bt <= 0) Bt = B;
    if (B % Bt) return -1017;
    hipStream_t st = (hipStream_t)stream;
    unsigned* counter = nullptr;
    float* part = sums + 16 * (size_t)B + ((B + 3) & ~3);
    { const int rc_ = tickets(st, B, &counter); if (rc_) return rc_; }
    hipLaunchKernelGGL(wce_dice_sums_kernel<false>, dim3(vgrid_red(V), B), dim3(256), 0, st, probs, (long)psb, (long)psv,
                       (long)psc, target, (long)V, sums, part, counter, out, Bt, UpG{1, 1, 1, 1, 1, 1, 1, 0.f, 0.f, 0.f, 1.f, 1.f});
    LAUNCH_RET();
    return 0;
}

namespace pconv { int upsample_axis_f32(const float* in, float* out, long outer, int Ni, int No, long inner, hipStream_t st); }   // resample.hip
static int up_geo(int D, int H, int W, int k, UpG& g) {
    if (k < 2 || D < 1 || H < 1 || W < 1) return -1011;
    g.D = D; g.H = H; g.W = W; g.k = k; g.Do = D * k; g.Ho = H * k; g.Wo = W * k;
    if ((long)g.Do * g.Ho * g.Wo >= (1L << 23)) return -1021;       // u_split's reciprocal arithmetic
    g.sd = g.Do > 1 ? (float)(D - 1) / (float)(g.Do - 1) : 0.f;
    g.sh = g.Ho > 1 ? (float)(H - 1) / (float)(g.Ho - 1) : 0.f;
    g.sw = g.Wo > 1 ? (float)(W - 1) / (float)(g.Wo - 1) : 0.f;
    g.inv_wo = 1.0f / (float)g.Wo; g.inv_ho = 1.0f / (float)g.Ho;
    return 0;
}

extern "C" int passion_wce_dice_up_fwd(const float* probs_coarse, const float* target, int B, int Bt, int D, int H, int W, int k,
                                       float* sums, float* out, void* stream) {
    UpG ug;
    { const int rc_ = up_geo(D, H, W, k, ug); if (rc_) return rc_; }
    if (Bt <= 0) Bt = B;
    if (B % Bt) return -1017;
    hipStream_t st = (hipStream_t)stream;
    unsigned* counter = nullptr;
    float* part = sums + 16 * (size_t)B + ((B + 3) & ~3);
    { const int rc_ = tickets(st, B, &counter); if (rc_) return rc_; }
    const long V = (long)ug.Do * ug.Ho * ug.Wo;
    hipLaunchKernelGGL(wce_dice_sums_kernel<false>, dim3(vgrid_red(V), B), dim3(256), 0, st, probs_coarse, 0L, 0L, 1L, target, V, sums, part, counter,
                       out, Bt, ug);
    LAUNCH_RET();
    return 0;
}

extern "C" int64_t passion_up_loss_bwd_scratch_floats(int B, int D, int H, int W, int k) {
    return (int64_t)B * D * k * H * k * W * 4 + (int64_t)B * D * k * H * W * 4;
}

// tmp [B][Do][Ho][W][4] <- fused loss gradient + W pass; then the H and D passes of the separable adjoint -> d(coarse map)
template <int MODE>
static int up_loss_bwd(const float* pc, const float* pt, const float* tg, const float* sums, const float* g, float* dpc, int B, int Bt,
                       const UpG& ug, float kk, float* scratch, hipStream_t st) {
    float* t1 = scratch;
    float* t2 = scratch + (long)B * ug.Do * ug.Ho * ug.W * 4;
    const long items = (long)ug.Do * ug.Ho * ug.W;
    if (items >= (1L << 23)) return -1021;
    static const bool rows = []() { const char* e = getenv("PASSION_UPLOSS_ROWS"); return !(e && e[0] == '0'); }();
    if (rows && ug.Wo <= 1024) {                      // one thread per fine voxel, whole rows per workgroup (16 KB of LDS at most)
        const int rpb = ug.Wo >= 256 ? 1 : 256 / ug.Wo;
        const long nrows = (long)ug.Do * ug.Ho;
        hipLaunchKernelGGL(up_loss_bwdw_rows_kernel<MODE>, dim3((unsigned)hmin((nrows + rpb - 1) / rpb, 8192), B), dim3(256),
                           (size_t)rpb * ug.Wo * 4 * sizeof(float), st, pc, pt, tg, sums, g, t1, ug, Bt, kk, rpb);
    } else
    hipLaunchKernelGGL(up_loss_bwdw_kernel<MODE>, dim3((unsigned)hmin((items + 255) / 256, 4096), B), dim3(256), 0, st, pc, pt, tg, sums, g, t1, ug,
                       Bt, kk);
    LAUNCH_RET();
    int rc = pconv::upsample_axis_f32(t1, t2, (long)B * ug.Do, ug.H, ug.Ho, (long)ug.W, st);
    if (rc) return rc;
    return pconv::upsample_axis_f32(t2, dpc, (long)B, ug.D, ug.Do, (long)ug.H * ug.W, st);
}

extern "C" int passion_wce_dice_up_bwd(const float* probs_coarse, const float* target, const float* sums, const float* g, float* dprobs_coarse,
                                       int B, int Bt, int D, int H, int W, int k, float* scratch, void* stream) {
    UpG ug;
    { const int rc_ = up_geo(D, H, W, k, ug); if (rc_) return rc_; }
    if (Bt <= 0) Bt = B;
    if (B % Bt || !scratch) return -1017;
    return up_loss_bwd<0>(probs_coarse, probs_coarse, target, sums, g, dprobs_coarse, B, Bt, ug, 0.f, scratch, (hipStream_t)stream);
}

extern "C" int passion_kl_up_fwd(const float* ps_coarse, const float* pt_coarse, int B, int Bt, int D, int H, int W, int k, float temp, float* out,
                                 float* work, void* stream) {
    UpG ug;
    { const int rc_ = up_geo(D, H, W, k, ug); if (rc_) return rc_; }
    if (Bt <= 0) Bt = B;
    if (B % Bt) return -1017;
    hipStream_t st = (hipStream_t)stream;
    unsigned* counter = nullptr;
    { const int rc_ = tickets(st, B, &counter); if (rc_) return rc_; }
    const long V = (long)ug.Do * ug.Ho * ug.Wo;
    const float kk = temp * temp / ((float)NC * (float)V);
    hipLaunchKernelGGL(kl_fwd_kernel, dim3(vgrid_red(V), B), dim3(256), 0, st, ps_coarse, pt_coarse, V, kk, out, work + ((B + 3) & ~3), counter, Bt, ug);
    LAUNCH_RET();
    return 0;
}

extern "C" int passion_kl_up_bwd(const float* ps_coarse, const float* pt_coarse, const float* g, float* dps_coarse, int B, int Bt, int D, int H,
                                 int W, int k, float temp, float* scratch, void* stream) {
    UpG ug;
    { const int rc_ = up_geo(D, H, W, k, ug); if (rc_) return rc_; }
    if (Bt <= 0) Bt = B;
    if (B % Bt || !scratch) return -1017;
    const long V = (long)ug.Do * ug.Ho * ug.Wo;
    const float kk = temp * temp / ((float)NC * (float)V);
    return up_loss_bwd<1>(ps_coarse, pt_coarse, nullptr, nullptr, g, dps_coarse, B, Bt, ug, kk, scratch, (hipStream_t)stream);
}
extern "C" int64_t passion_wce_dice_sums_floats(int B) { return 16 * (int64_t)B + red_extra_floats(B, 16); }
extern "C" int64_t passion_kl_work_floats(int B) { return red_extra_floats(B, 1); }

extern "C" int passion_wce_dice_bwd(const float* probs, int64_t psb, int64_t psv, int64_t psc, const float* target,
                                    const float* sums, const float* g, float* dprobs, int B, int Bt, int64_t V, int C,
                                    void* stream) {
    if (C != NC) return -1012;
    if (Bt <= 0) Bt = B;
    if (B % Bt) return -1017;
    hipLaunchKernelGGL(wce_dice_bwd_kernel, dim3(vgrid(V), B), dim3(256), 0, (hipStream_t)stream, probs, (long)psb,
                       (long)psv, (long)psc, target, sums, g, dprobs, (long)V, Bt);
    LAUNCH_RET();
    return 0;
}

extern "C" int passion_softmax_wce_dice_fwd(const float* logits, const float* target, int B, int Bt, int64_t V, int C, float inv_temp,
                                            float* sums, float* out, void* stream) {
    if (C != NC) return -1012;
    if (Bt <= 0) Bt = B;
    if (B % Bt) return -1017;
    hipStream_t st = (hipStream_t)stream;
    unsigned* counter = nullptr;
    float* part = sums + 16 * (size_t)B + ((B + 3) & ~3);
    { const int rc_ = tickets(st, B, &counter); if (rc_) return rc_; }
    hipLaunchKernelGGL(wce_dice_sums_kernel<true>, dim3(vgrid_red(V), B), dim3(256), 0, st, logits, (long)V * NC, (long)NC, 1L, target, (long)V,
                       sums, part, counter, out, Bt, UpG{1, 1, 1, 1, 1, 1, 1, 0.f, 0.f, 0.f, 1.f, 1.f}, inv_temp);
    LAUNCH_RET();
    return 0;
}

extern "C" int passion_softmax_wce_dice_bwd(const float* logits, const float* target, const float* sums, const float* g, float* dlogits,
                                            int B, int Bt, int64_t V, int C, float inv_temp, void* stream) {
    if (C != NC) return -1012;
    if (Bt <= 0) Bt = B;
    if (B % Bt) return -1017;
    hipLaunchKernelGGL(softmax_wce_dice_bwd_kernel, dim3(vgrid(V), B), dim3(256), 0, (hipStream_t)stream, logits, target, sums, g, dlogits,
                       (long)V, Bt, inv_temp);
    LAUNCH_RET();
    return 0;
}

extern "C" int passion_kl_fwd(const float* ps, const float* pt, int B, int Bt, int64_t V, int C, float temp, float* out,
                              float* work, void* stream) {
    if (C != NC) return -1012;
    if (Bt <= 0) Bt = B;
    if (B % Bt) return -1017;
    hipStream_t st = (hipStream_t)stream;
    unsigned* counter = nullptr;
    { const int rc_ = tickets(st, B, &counter); if (rc_) return rc_; }
    const float k = temp * temp / ((float)NC * (float)V);
    hipLaunchKernelGGL(kl_fwd_kernel, dim3(vgrid_red(V), B), dim3(256), 0, st, ps, pt, (long)V, k, out,
                       work + ((B + 3) & ~3), counter, Bt, UpG{1, 1, 1, 1, 1, 1, 1, 0.f, 0.f, 0.f, 1.f, 1.f});
    LAUNCH_RET();
    return 0;
}

extern "C" int passion_kl_bwd(const float* ps, const float* pt, const float* g, float* dps, int B, int Bt, int64_t V, int C,
                              float temp, void* stream) {
    if (C != NC) return -1012;
    if (Bt <= 0) Bt = B;
    if (B % Bt) return -1017;
    const float k = temp * temp / ((float)NC * (float)V);
    hipLaunchKernelGGL(kl_bwd_kernel, dim3(vgrid(V), B), dim3(256), 0, (hipStream_t)stream, ps, pt, g, dps, (long)V, k, Bt);
    LAUNCH_RET();
    return 0;
}

extern "C" int64_t passion_proto_work_floats(int B, int Cf) { (void)Cf; return (int64_t)B * PW + 8 + red_extra_floats(B, 4 + 8 * CF); }

extern "C" int passion_proto_fwd(const void* fs, const void* ft, int dtype, const float* target, int B, int Bt, int64_t V,
                                 int Cf, float* work, float* out, void* stream) {
    if (Cf != 4 && Cf != 8) return -1013;
    if (Bt <= 0) Bt = B;
    if (B % Bt) return -1017;
    hipStream_t st = (hipStream_t)stream;
    unsigned* counter = nullptr;                     // every word of `work` that is read is written first by these kernels
    float* part = work + (long)B * PW + 8 + ((B + 3) & ~3);
    { const int rc_ = tickets(st, B, &counter); if (rc_) return rc_; }
    dim3 grid(vgrid_red_proto(V), B);
    if (dtype == PASSION_F32)
        hipLaunchKernelGGL(proto_sums_kernel<float>, grid, dim3(256), 0, st, (const float*)fs, (const float*)ft, target, (long)V, work, Cf, part, counter, Bt);
    else if (dtype == PASSION_BF16)
        hipLaunchKernelGGL(proto_sums_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)fs, (const bf16_t*)ft, target, (long)V, work, Cf, part, counter, Bt);
    else
        return -1007;
    LAUNCH_RET();
    hipLaunchKernelGGL(proto_valid_kernel, dim3(1), dim3(64), 0, st, work, B);
    LAUNCH_RET();
    if (dtype == PASSION_F32)
        hipLaunchKernelGGL(proto_loss_kernel<float>, grid, dim3(256), 0, st, (const float*)fs, (const float*)ft, (long)V, B, work, Cf, part, counter, Bt);
    else
        hipLaunchKernelGGL(proto_loss_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)fs, (const bf16_t*)ft, (long)V, B, work, Cf, part, counter, Bt);
    LAUNCH_RET();
    hipLaunchKernelGGL(proto_out_kernel, dim3((B + 63) / 64), dim3(64), 0, st, work, B, out);
    LAUNCH_RET();
    return 0;
}

extern "C" int passion_proto_bwd(const void* fs, const void* ft, int dtype, const float* target, const float* g, int B, int Bt,
                                 int64_t V, int Cf, float* work, void* dfs, void* stream) {
    if (Cf != 4 && Cf != 8) return -1013;
    if (Bt <= 0) Bt = B;
    if (B % Bt) return -1017;
    hipStream_t st = (hipStream_t)stream;
    // `work` still holds the forward's sums / validity; the Gp block is overwritten by the last block of bwd1; the ticket
    // counters were left at zero by the forward's reductions
    unsigned* counter = nullptr;
    float* part = work + (long)B * PW + 8 + ((B + 3) & ~3);
    { const int rc_ = tickets(st, B, &counter); if (rc_) return rc_; }
    dim3 grid(vgrid_red_proto(V, 2), B);
    dim3 grid2(vgrid(V), B);
    if (dtype == PASSION_F32) {
        hipLaunchKernelGGL(proto_bwd1_kernel<float>, grid, dim3(256), 0, st, (const float*)fs, (const float*)ft, g, (long)V, B, work, (float*)dfs, Cf, part, counter, Bt);
        LAUNCH_RET();
        hipLaunchKernelGGL(proto_bwd2_kernel<float>, grid2, dim3(256), 0, st, target, (long)V, work, (float*)dfs, Cf, Bt);
    } else if (dtype == PASSION_BF16) {
        hipLaunchKernelGGL(proto_bwd1_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)fs, (const bf16_t*)ft, g, (long)V, B, work, (bf16_t*)dfs, Cf, part, counter, Bt);
        LAUNCH_RET();
        hipLaunchKernelGGL(proto_bwd2_kernel<bf16_t>, grid2, dim3(256), 0, st, target, (long)V, work, (bf16_t*)dfs, Cf, Bt);
    } else
        return -1007;
    LAUNCH_RET();
    return 0;
}


extern "C" int64_t passion_colsum_work_floats(int C) { return (int64_t)((C + 127) / 128 + 3) / 4 * 4 + (int64_t)COLSUM_SLICES * C; }
extern "C" int passion_colsum_batched(const void* g, int dtype, int G, int64_t R, int C, float* out, float* work, void* stream);
extern "C" int passion_colsum(const void* g, int dtype, int64_t R, int C, float* out, float* work, void* stream) {
    return passion_colsum_batched(g, dtype, 1, R, C, out, work, stream);
}
// G matrices [G][R][C] -> out [G][C]; work: G x passion_colsum_work_floats(C) floats
extern "C" int passion_colsum_batched(const void* g, int dtype, int G, int64_t R, int C, float* out, float* work, void* stream) {
    if (C <= 0 || (C & 1) || G < 1) return -1014;
    hipStream_t st = (hipStream_t)stream;
    const int nbx = (C + 127) / 128;
    // enough row slices to put ~2 blocks on every CU, at least 8 rows per wave (two batches of four loads)
    int ny = (int)hmin((long)COLSUM_SLICES, hmax(1L, hmin((long)(512 + nbx - 1) / nbx, R / 32)));
    unsigned* counter = nullptr;                               // [G][nbx] tickets (library scratch); work: [G][ny][C] partial sums
    float* part = work + ((long)G * nbx + 3) / 4 * 4;
    { const int rc_ = tickets(st, (size_t)nbx * G, &counter); if (rc_) return rc_; }
    if (dtype == PASSION_BF16)
        hipLaunchKernelGGL(colsum_kernel<bf16_t>, dim3(nbx, ny, G), dim3(256), 0, st, (const bf16_t*)g, (long)R, C, out, part, counter);
    else if (dtype == PASSION_F32)
        hipLaunchKernelGGL(colsum_kernel<float>, dim3(nbx, ny, G), dim3(256), 0, st, (const float*)g, (long)R, C, out, part, counter);
    else
        return -1007;
    LAUNCH_RET();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// AdamW (amsgrad) over many parameter tensors in one launch per <= ADAM_MAX tensors -- torch.optim.AdamW semantics
// (reference code/train.py:95-96: betas (0.9, 0.999), eps 1e-8, amsgrad=True, decoupled weight decay):
//   p *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  vmax = max(vmax, v);
//   p -= (lr / (1 - b1^t)) * m / (sqrt(vmax) / sqrt(1 - b2^t) + eps)
// The foreach implementation of torch takes ~1.4 ms for the 288 tensors / 35.6 M parameters of mmformer; this streams
// the 36 bytes per parameter once.
namespace {
constexpr int ADAM_MAX = 40;
struct AdamEnt { float* p; const float* g; float* m; float* v; float* vmax; long n; };
struct AdamTable { AdamEnt e[ADAM_MAX]; };

__global__ __launch_bounds__(256) void adamw_multi_kernel(AdamTable t, float lr_wd, float omb1, float b2, float omb2, float step_size,
                                                          float inv_bc2_sqrt, float eps, int amsgrad, const float* hyp) {
    if (hyp) {      // a replayed HIP graph: the step-dependent scalars come from device memory (passion_adamw_step_dev)
        lr_wd = hyp[0]; step_size = hyp[1]; inv_bc2_sqrt = hyp[2];
    }
    const AdamEnt& q = t.e[blockIdx.y];
    const long n4 = q.n / 4;
    const bool vec = ((((size_t)q.p | (size_t)q.g | (size_t)q.m | (size_t)q.v | (size_t)q.vmax) & 15) == 0);
    auto upd = [&](float& p, float g, float& m, float& v, float& vm) {
        p = p * (1.f - lr_wd);
        m = m + (g - m) * omb1;              // lerp, as torch (omb = 1 - beta, rounded from double on the host)
        v = b2 * v + omb2 * g * g;
        float d = v;
        if (amsgrad) { vm = fmaxf(vm, v); d = vm; }
        p = p - step_size * (m / (sqrtf(d) * inv_bc2_sqrt + eps));
    };
    if (vec) {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
            f32x4 p = reinterpret_cast<f32x4*>(q.p)[i], m = reinterpret_cast<f32x4*>(q.m)[i], v = reinterpret_cast<f32x4*>(q.v)[i];
            const f32x4 g = reinterpret_cast<const f32x4*>(q.g)[i];
            f32x4 vm = amsgrad ? reinterpret_cast<f32x4*>(q.vmax)[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float pj = p[j], mj = m[j], vj = v[j], vmj = vm[j];
                upd(pj, g[j], mj, vj, vmj);
                p[j] = pj; m[j] = mj; v[j] = vj; vm[j] = vmj;
            }
            reinterpret_cast<f32x4*>(q.p)[i] = p; reinterpret_cast<f32x4*>(q.m)[i] = m; reinterpret_cast<f32x4*>(q.v)[i] = v;
            if (amsgrad) reinterpret_cast<f32x4*>(q.vmax)[i] = vm;
        }
    }
    for (long i = (vec ? n4 * 4 : 0) + (long)blockIdx.x * 256 + threadIdx.x; i < q.n; i += (long)gridDim.x * 256) {
        float vm = amsgrad ? q.vmax[i] : 0.f;
        upd(q.p[i], q.g[i], q.m[i], q.v[i], vm);
        if (amsgrad) q.vmax[i] = vm;
    }
}
}  // namespace

extern "C" int passion_adamw_hyper(double lr, double beta1d, double beta2d, double weight_decay, int64_t step, float* out3) {
    if (step < 1) return -1031;
    const double bc1 = 1.0 - pow(beta1d, (double)step), bc2 = 1.0 - pow(beta2d, (double)step);
    out3[0] = (float)(lr * weight_decay); out3[1] = (float)(lr / bc1); out3[2] = (float)(1.0 / sqrt(bc2));
    return 0;
}

static int adamw_step_impl(int n, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                           float* const* max_exp_avg_sq, const int64_t* numel, double lr, double beta1d, double beta2d, double eps,
                           double weight_decay, int64_t step, int amsgrad, const float* hyp, void* stream);

extern "C" int passion_adamw_step(int n, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                                  float* const* max_exp_avg_sq, const int64_t* numel, double lr, double beta1d, double beta2d, double eps,
                                  double weight_decay, int64_t step, int amsgrad, void* stream) {
    return adamw_step_impl(n, params, grads, exp_avg, exp_avg_sq, max_exp_avg_sq, numel, lr, beta1d, beta2d, eps, weight_decay, step, amsgrad,
                           nullptr, stream);
}

extern "C" int passion_adamw_step_dev(int n, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                                      float* const* max_exp_avg_sq, const int64_t* numel, double beta1d, double beta2d, double eps,
                                      int amsgrad, const float* hyper_dev, void* stream) {
    if (!hyper_dev) return -1031;
    return adamw_step_impl(n, params, grads, exp_avg, exp_avg_sq, max_exp_avg_sq, numel, 0.0, beta1d, beta2d, eps, 0.0, 1, amsgrad, hyper_dev,
                           stream);
}

static int adamw_step_impl(int n, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                           float* const* max_exp_avg_sq, const int64_t* numel, double lr, double beta1d, double beta2d, double eps,
                           double weight_decay, int64_t step, int amsgrad, const float* hyp, void* stream) {
    if (step < 1) return -1031;
    const float beta2 = (float)beta2d;
    const double bc1 = 1.0 - pow(beta1d, (double)step), bc2 = 1.0 - pow(beta2d, (double)step);
    const float step_size = (float)(lr / bc1), inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    for (int i0 = 0; i0 < n; i0 += ADAM_MAX) {
        AdamTable t;
        const int m = (n - i0 < ADAM_MAX) ? n - i0 : ADAM_MAX;
        long nmax = 0;
        for (int i = 0; i < m; ++i) {
            t.e[i] = AdamEnt{params[i0 + i], grads[i0 + i], exp_avg[i0 + i], exp_avg_sq[i0 + i],
                             amsgrad ? max_exp_avg_sq[i0 + i] : nullptr, (long)numel[i0 + i]};
            nmax = numel[i0 + i] > nmax ? numel[i0 + i] : nmax;
        }
        const int gx = (int)hmax(1, hmin(256, (nmax / 4 + 255) / 256));
        hipLaunchKernelGGL(adamw_multi_kernel, dim3(gx, m), dim3(256), 0, (hipStream_t)stream, t, (float)(lr * weight_decay),
                           (float)(1.0 - (double)beta1d), beta2, (float)(1.0 - (double)beta2d), step_size, inv_bc2_sqrt, (float)eps, amsgrad, hyp);
        LAUNCH_RET();
    }
    return 0;
}
