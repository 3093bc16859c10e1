// Fused (InstanceNorm -> LeakyReLU ->) Conv3d forward / dgrad / wgrad for channels-last volumes.
//
// Replaces general_conv3d_prenorm (reference code/models/blocks.py:300-316), bare nn.Conv3d
// (code/models/mmformer.py:28,90,136-140,337-340,355) and the MaskModal/stack/cat feeding them.
//
// Implicit GEMM on the matrix cores:  D[M = out-channel][N = voxel] += A[M][K] * B[K][N]
// with K = (tap, in-channel).  A (weights) streams from L2 in a packed [tap][K][Mpad] image,
// B is an LDS tile of the input brick (+halo) that the loader has already normalised, activated,
// masked and padded -- each input element is touched once per block instead of 27 times.
// One 256-thread workgroup = 4 wavefronts = a 4x8x8 brick of output voxels; wave w owns depth
// slice w = four 16-voxel MFMA column tiles.  v_mfma_f32_16x16x4_f32 keeps exact f32 numerics
// (a k-ordered fmaf chain), which is what the 1e-3 parity gate is judged on.
//
// dgrad is the same kernel (MODE 1): B = dY brick (zero-stuffed for stride 2), A = the
// [tap][Cout][CinPad] image, plus "mirror taps" that implement the adjoint of reflect padding,
// and an epilogue that goes back through LeakyReLU and accumulates the two InstanceNorm-backward
// sums per (b, channel).
#include <cstdio>
#include <mutex>
#include <vector>
#include "conv_common.h"
#include "prof.h"

using namespace pconv;

int g_passion_force_f32_mfma = 0;   // debugging / A-B switch: route bf16 tensors through the f32-MFMA kernel

namespace {

template <int KS, int S, bool DG> struct Geo {
    static constexpr int HALO = (KS == 3) ? 1 : 0;
    static constexpr int ID = DG ? TD + 2 * HALO : (TD - 1) * S + KS;
    static constexpr int IH = DG ? TH + 2 * HALO : (TH - 1) * S + KS;
    static constexpr int IW = DG ? TW + 2 * HALO : (TW - 1) * S + KS;
    static constexpr int VOL = ID * IH * IW;
    static constexpr int PLANE = ((VOL + 31) / 32) * 32 + 20;
    static constexpr int CK = (KS == 1) ? 16 : ((!DG && S == 2) ? 4 : 8);
    static constexpr int NTAP = KS * KS * KS;
};

// Stage CK channels [c0, c0+ck) of one tensor brick into lds[c][PLANE] as f32.
//  fwd  : source = segment x at input coordinates (reflect / zero padded), normalised+activated.
//  dgrad: source = dY at u = i0 - HALO + z, present only where u is a multiple of S (zero-stuffing).
template <typename TI, int KS, int S, bool DG>
__device__ __forceinline__ void stage_tile(float* lds, const TI* __restrict__ src, long bstride_unused,
                                           int C, int c0, int ck, const float* __restrict__ stats,
                                           float mval, float slope, int pad_mode, int b,
                                           int o0d, int o0h, int o0w,  // tile origin (output coords for fwd, input-of-fwd coords for dgrad)
                                           int Sd, int Sh, int Sw,     // spatial size of the staged tensor
                                           int tid) {
    using G = Geo<KS, S, DG>;
    float mean[G::CK], rstd[G::CK];
    const bool has_norm = (stats != nullptr);
#pragma unroll
    for (int c = 0; c < G::CK; ++c) {
        mean[c] = 0.f; rstd[c] = 1.f;
        if (has_norm && c < ck) {
            mean[c] = stats[((long)b * C + c0 + c) * 2 + 0];
            rstd[c] = stats[((long)b * C + c0 + c) * 2 + 1];
        }
    }
    const bool vec = ((C & 3) == 0) && ((ck & 3) == 0);
    for (int tv = tid; tv < G::VOL; tv += 256) {
        int zw = tv % G::IW, zh = (tv / G::IW) % G::IH, zd = tv / (G::IW * G::IH);
        int sd, sh, sw;
        bool ok = true;
        if (!DG) {
            sd = o0d * S - G::HALO + zd; sh = o0h * S - G::HALO + zh; sw = o0w * S - G::HALO + zw;
            if (pad_mode == PASSION_PAD_REFLECT) {
                sd = reflect_idx(sd, Sd); sh = reflect_idx(sh, Sh); sw = reflect_idx(sw, Sw);
                // bricks that overhang the volume may still point outside: clamp (their outputs are discarded)
                sd = min(max(sd, 0), Sd - 1); sh = min(max(sh, 0), Sh - 1); sw = min(max(sw, 0), Sw - 1);
            } else {
                ok = (sd >= 0 && sd < Sd && sh >= 0 && sh < Sh && sw >= 0 && sw < Sw);
            }
        } else {
            int ud = o0d - G::HALO + zd, uh = o0h - G::HALO + zh, uw = o0w - G::HALO + zw;
            ok = (ud >= 0 && uh >= 0 && uw >= 0 && (ud % S) == 0 && (uh % S) == 0 && (uw % S) == 0);
            sd = ud / S; sh = uh / S; sw = uw / S;
            ok = ok && (sd < Sd && sh < Sh && sw < Sw);
        }
        ok = ok && (mval != 0.f);
        const TI* ptr = src + (((long)sd * Sh + sh) * Sw + sw) * C + c0;
        if (vec) {
#pragma unroll
            for (int c4 = 0; c4 < G::CK / 4; ++c4) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ok && c4 * 4 < ck) v = load4(ptr + c4 * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float t = v[j];
                    if (has_norm) {
                        t = (t - mean[c4 * 4 + j]) * rstd[c4 * 4 + j];
                        t = t > 0.f ? t : t * slope;
                    }
                    if (!ok || c4 * 4 + j >= ck) t = 0.f;
                    lds[(c4 * 4 + j) * G::PLANE + tv] = t;
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < G::CK; ++c) {
                float t = 0.f;
                if (ok && c < ck) {
                    t = to_f32(ptr[c]);
                    if (has_norm) {
                        t = (t - mean[c]) * rstd[c];
                        t = t > 0.f ? t : t * slope;
                    }
                }
                lds[c * G::PLANE + tv] = t;
            }
        }
    }
}

template <typename TI, typename TO, int KS, int S, int MT, int MODE>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvK p) {
    constexpr bool DG = (MODE == 1);
    using G = Geo<KS, S, DG>;
    constexpr int CK = G::CK;
    __shared__ float lds[CK * G::PLANE];
    __shared__ float lsum[DG ? 4 * MT * 16 * 2 : 1];

    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, n = lane & 15, kq = lane >> 4;
    int bid = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tw_ = bid % p.tilesW; bid /= p.tilesW;
    const int th_ = bid % p.tilesH; bid /= p.tilesH;
    const int td_ = bid % p.tilesD; bid /= p.tilesD;
    const int b = bid;
    const int o0d = td_ * TD, o0h = th_ * TH, o0w = tw_ * TW;
    const int mg = blockIdx.y;  // group of MT row tiles

    // spatial domain of this kernel's OUTPUT (fwd: y; dgrad: dx) and of the staged tensor
    const int Od = DG ? p.Di : p.Do, Oh = DG ? p.Hi : p.Ho, Ow = DG ? p.Wi : p.Wo;
    const int Sd = DG ? p.Do : p.Di, Sh = DG ? p.Ho : p.Hi, Sw = DG ? p.Wo : p.Wi;
    const int Ktot = DG ? p.Cout : p.Cin;

    f32x4 acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int lh[4] = {0 + (n >> 3), 2 + (n >> 3), 4 + (n >> 3), 6 + (n >> 3)};
    const int lw = n & 7, ld_ = wv;

    const bool reflect_dg = DG && (KS == 3) && (p.pad_mode == PASSION_PAD_REFLECT);
    const int NOPT = reflect_dg ? KS + 2 : KS;
    // which mirror planes does this brick touch (workgroup-uniform)
    const bool lowD = reflect_dg && (o0d == 0) && (Od >= 2), lowH = reflect_dg && (o0h == 0) && (Oh >= 2),
               lowW = reflect_dg && (o0w == 0) && (Ow >= 2);
    const bool hiD = reflect_dg && p.hi_t[0] >= 0 && (o0d <= Od - 2) && (Od - 2 < o0d + TD);
    const bool hiH = reflect_dg && p.hi_t[1] >= 0 && (o0h <= Oh - 2) && (Oh - 2 < o0h + TH);
    const bool hiW = reflect_dg && p.hi_t[2] >= 0 && (o0w <= Ow - 2) && (Ow - 2 < o0w + TW);

    const int nchunks_seg_max = 0; (void)nchunks_seg_max;
    const int nseg = DG ? 1 : p.nseg;
    for (int sg = 0; sg < nseg; ++sg) {
        const int C = DG ? p.Cout : p.seg[sg].C;
        const int cbase = DG ? 0 : p.seg[sg].cbase;
        const TI* src;
        const float* stats = nullptr;
        float mval = 1.f;
        if (DG) {
            src = reinterpret_cast<const TI*>(p.dy) + (long)b * Sd * Sh * Sw * C;
        } else {
            src = reinterpret_cast<const TI*>(p.seg[sg].x) + (long)p.seg[sg].bsrc(b) * p.seg[sg].bstride;
            stats = p.seg[sg].stats;
            if (p.seg[sg].mask) mval = p.seg[sg].mask[(long)b * p.seg[sg].mask_ld + p.seg[sg].mask_col] ? 1.f : 0.f;
        }
        if (mval == 0.f) continue;  // absent modality contributes exactly zero
        for (int c0 = 0; c0 < C; c0 += CK) {
            const int ck = min(CK, C - c0);
            __syncthreads();
            stage_tile<TI, KS, S, DG>(lds, src, 0, C, c0, ck, stats, mval, p.slope, p.pad_mode, DG ? b : p.seg[sg].bsrc(b), o0d, o0h, o0w,
                                      Sd, Sh, Sw, tid);
            __syncthreads();
            for (int ad = 0; ad < NOPT; ++ad) {
                if (ad == KS && !lowD) continue;
                if (ad == KS + 1 && !hiD) continue;
                int tdp, posd; bool vd;
                axis_tap<KS, G::HALO, DG, S>(ad, ld_, o0d, Od, p.hi_t[0], p.hi_u[0], tdp, posd, vd);
                for (int ah = 0; ah < NOPT; ++ah) {
                    if (ah == KS && !lowH) continue;
                    if (ah == KS + 1 && !hiH) continue;
                    for (int aw = 0; aw < NOPT; ++aw) {
                        if (aw == KS && !lowW) continue;
                        if (aw == KS + 1 && !hiW) continue;
                        int twp, posw; bool vw;
                        axis_tap<KS, G::HALO, DG, S>(aw, lw, o0w, Ow, p.hi_t[2], p.hi_u[2], twp, posw, vw);
                        int thp = 0, posh[4]; bool vh[4];
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
                            axis_tap<KS, G::HALO, DG, S>(ah, lh[nt], o0h, Oh, p.hi_t[1], p.hi_u[1], thp, posh[nt], vh[nt]);
                        const int tap = (tdp * KS + thp) * KS + twp;
                        const float* wrow = reinterpret_cast<const float*>(p.wpack) + ((long)tap * Ktot + cbase + c0) * p.Mpad + (mg * MT) * 16 + n;
                        int pos[4]; bool val[4];
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) {
                            val[nt] = vd && vh[nt] && vw;
                            pos[nt] = val[nt] ? ((posd * G::IH + posh[nt]) * G::IW + posw) : 0;
                        }
#pragma unroll
                        for (int st = 0; st < CK / 4; ++st) {
                            const int c = st * 4 + kq;
                            float a[MT];
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt) a[mt] = (c < ck) ? wrow[(long)c * p.Mpad + mt * 16] : 0.f;
#pragma unroll
                            for (int nt = 0; nt < 4; ++nt) {
                                float bv = lds[c * G::PLANE + pos[nt]];
                                if (reflect_dg) bv = val[nt] ? bv : 0.f;
#pragma unroll
                                for (int mt = 0; mt < MT; ++mt)
                                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], bv, acc[mt][nt], 0, 0, 0);
                            }
                        }
                    }
                }
            }
        }
    }

    // ---------------- epilogue ----------------
    const int oh4[4] = {o0h + lh[0], o0h + lh[1], o0h + lh[2], o0h + lh[3]};
    if (!DG) epilogue_fwd<TO, MT>(p, acc, b, mg, o0d + ld_, o0w + lw, oh4, kq);
    else epilogue_dgrad<TO, MT>(p, acc, lsum, b, mg, o0d + ld_, o0w + lw, oh4, kq, n, tid);
}

// InstanceNorm backward, second half:  dx = rstd * (g - mean(g) - xhat * mean(g * xhat))
// where g = dL/dxhat (already written into dx by the dgrad epilogue) -- in place.  Per channel this is an affine map of
// (g, x): dx = a g + b x + c0 with a = rstd, b = -rstd^2 m2, c0 = rstd^2 m2 mean - rstd m1; the block folds the `rep`
// replicas of the sums and tabulates (a, b, c0) in LDS once, the sweep is then two FMAs per element on VEC-channel
// vectors (16-byte accesses for bf16 when C % 8 == 0).
// coef[c] = (a, b, c0) for the C <= FIN_MAXC channels of this block's sample; folds the rep replicas of the sums
constexpr int FIN_MAXC = 512;
// rep < 0: -rep replicas of 64-bit fixed-point (2^-40) sums (ConvK::sums_fixed)
__device__ __forceinline__ void finish_coefs(float (*coef)[3], const float* stats, const float* sums, int b, int bs, int C, int Cin_tot,
                                             int cbase, int rep, long rep_stride, float invV) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s1 = 0.f, s2 = 0.f;
        if (rep < 0) {
            const long long* fs = reinterpret_cast<const long long*>(sums);
            long long a1 = 0, a2 = 0;
            for (int r = 0; r < -rep; ++r) {
                a1 += fs[(long)r * rep_stride + ((long)b * Cin_tot + cbase + c) * 2 + 0];
                a2 += fs[(long)r * rep_stride + ((long)b * Cin_tot + cbase + c) * 2 + 1];
            }
            s1 = (float)((double)a1 * 9.094947017729282e-13);                   // 2^-40
            s2 = (float)((double)a2 * 9.094947017729282e-13);
        }
        for (int r = 0; r < rep; ++r) {
            s1 += sums[(long)r * rep_stride + ((long)b * Cin_tot + cbase + c) * 2 + 0];
            s2 += sums[(long)r * rep_stride + ((long)b * Cin_tot + cbase + c) * 2 + 1];
        }
        const float mean = stats[((long)bs * C + c) * 2 + 0], rstd = stats[((long)bs * C + c) * 2 + 1];
        const float m1 = s1 * invV, m2 = s2 * invV;
        coef[c][0] = rstd;
        coef[c][1] = -rstd * rstd * m2;
        coef[c][2] = rstd * rstd * m2 * mean - rstd * m1;
    }
}

// The fold of a weight gradient's partial slab (out[i] = sum_r part[r * stride + i], fixed order) riding in a finish launch: the
// finish kernels are the last launch of a conv backward and do not depend on the weight gradient, so the last fold_x blocks of
// every row of their grid do the fold that used to be a launch of its own between the wgrad and the dgrad kernel (73 launches of ~10 us on
// the critical stream of a training step, each with its dependent-launch gap).  G > 1: group g's R slots follow group g - 1's.
struct FoldJob {
    const float* part;
    float* out[PASSION_MAX_GROUPS];
    long n, stride;
    int R, G, main_x;              // main_x == 0: no job, the whole grid is the finish kernel's; else the finish kernel's share of x
    int fold_x;                    // blocks x >= main_x fold (fold blocks FIRST in every row was measured: 3 us slower per finish launch)
};
// Values per block of a fold over R slots: 16 slices x 16 values for the long folds of small weights (1024 slots x 1.7 K values at the
// fine levels), whole rows of values for the short folds of big weights (level 5: 8 slots x 442 K values took 27 K blocks with 16
// loading lanes each -- 65 us for a 14 MB slab).  The order of the adds is a function of R alone.
__host__ __device__ inline int fold_vpb(int R) { return R >= 64 ? 16 : (R >= 16 ? 64 : 256); }
__host__ __device__ inline long fold_nblocks(long n, int R) { const int v = fold_vpb(R); return (n + v - 1) / v; }

// one block's share of the fold: value i = sum over the slots, slices of the slot axis over the thread groups
__device__ __forceinline__ float fold_value(const float* __restrict__ pg, long stride, int R, long n, long blk, long& i_out, bool& owner) {
    __shared__ float red[256 + 16];
    const int vpb = fold_vpb(R), nsl = 256 / vpb;
    const int li = threadIdx.x % vpb, sl = threadIdx.x / vpb;
    const long i = blk * vpb + li;
    float a = 0.f;
    if (i < n) {                                   // four loads in flight per thread; the order of the adds is fixed all the same
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int r = sl;
        for (; r + 3 * nsl < R; r += 4 * nsl) {
            a0 += pg[(long)r * stride + i];
            a1 += pg[(long)(r + nsl) * stride + i];
            a2 += pg[(long)(r + 2 * nsl) * stride + i];
            a3 += pg[(long)(r + 3 * nsl) * stride + i];
        }
        for (; r < R; r += nsl) a0 += pg[(long)r * stride + i];
        a = (a0 + a1) + (a2 + a3);
    }
    i_out = i;
    owner = sl == 0 && i < n;
    if (nsl == 1) return a;                        // block-uniform
    red[sl * (vpb + 1) + li] = a;
    __syncthreads();
    float t = 0.f;
    if (owner)
        for (int k2 = 0; k2 < nsl; ++k2) t += red[k2 * (vpb + 1) + li];
    return t;
}

__device__ __forceinline__ void fold_block(const FoldJob& fj, long id) {
    const long nb = fold_nblocks(fj.n, fj.R);
    if (id >= nb * fj.G) return;                   // block-uniform
    const int g = (int)(id / nb);
    long i;
    bool owner;
    const float t = fold_value(fj.part + (long)g * fj.R * fj.stride, fj.stride, fj.R, fj.n, id % nb, i, owner);
    if (owner) {
        float* o = fj.out[0];
#pragma unroll
        for (int g2 = 1; g2 < PASSION_MAX_GROUPS; ++g2) if (g == g2) o = fj.out[g2];
        o[i] = t;
    }
}

// `passes` > 1 (SegD.bmod: P masked passes over the same bmod samples in one launch): g holds P*bmod per-pass gradients,
// dx the bmod physical samples; the kernel finishes every pass with its own sums / mask and writes their SUM -- what
// autograd would otherwise do with P full-size tensors and a reduction.  passes == 1: in place (g == dx).
constexpr int FIN_TAB = 1024;                     // passes * C entries of (a, b, c0)
template <typename T, int VEC>
__global__ __launch_bounds__(256) void in_bwd_finish_kernel(const T* __restrict__ g_in, T* __restrict__ dx, const T* __restrict__ x,
                                                            const float* __restrict__ stats, const float* __restrict__ sums,
                                                            const uint8_t* mask, int mask_col, int mask_ld, long V, int C, long bstride,
                                                            int Cin_tot, int cbase, int bmod, int passes, int rep, long rep_stride,
                                                            float vscale, long win_lo, long win_hi, FoldJob fj) {
    if (fj.main_x && (int)blockIdx.x >= fj.main_x) { fold_block(fj, (long)(blockIdx.x - fj.main_x) * gridDim.y + blockIdx.y); return; }
    const long gdx = fj.main_x ? fj.main_x : gridDim.x;
    const long bix = blockIdx.x;
    __shared__ float coef[FIN_TAB][3];
    __shared__ int pres[16];
    const int bp = blockIdx.y;                     // physical sample (x, stats, dx when passes > 1) or the plain batch index
    const int bs = (passes == 1 && bmod > 0) ? bp % bmod : bp;
    for (int p = 0; p < passes; ++p) {
        const int b = p * (passes > 1 ? bmod : 0) + bp;        // virtual sample of pass p
        if (stats) finish_coefs(coef + p * C, stats, sums, b, bs, C, Cin_tot, cbase, rep, rep_stride, vscale / (float)V);
        if (threadIdx.x == 0) pres[p] = mask ? (mask[(long)b * mask_ld + mask_col] != 0) : 1;
    }
    __syncthreads();
    const long nvec = V * C / VEC;
    const int cvn = C / VEC;
    T* dxb = dx + (long)bp * bstride;
    const T* xb = x + (long)bs * bstride;
    for (long i = bix * blockDim.x + threadIdx.x; i < nvec; i += gdx * blockDim.x) {
        const int c = (int)(i % cvn) * VEC;
        float acc[VEC], xv[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) { acc[j] = 0.f; xv[j] = 0.f; }
        if (stats) VecIO<T, VEC>::load(xb + i * VEC, xv);
        for (int p = 0; p < passes; ++p) {
            if (!pres[p]) continue;                                   // workgroup-uniform
            const long b = (long)p * (passes > 1 ? bmod : 0) + bp;
            float g[VEC];
            VecIO<T, VEC>::load(g_in + b * bstride + i * VEC, g);
            const float (*cf)[3] = coef + p * C + c;
            // voxels outside [win_lo, win_hi) are another rank's (depth-sharded halo planes): only the rstd * g term is ours
            const bool own = (i * VEC) / C >= win_lo && (i * VEC) / C < win_hi;
#pragma unroll
            for (int j = 0; j < VEC; ++j)
                acc[j] += stats ? (own ? __builtin_fmaf(cf[j][0], g[j], __builtin_fmaf(cf[j][1], xv[j], cf[j][2])) : cf[j][0] * g[j]) : g[j];
        }
        VecIO<T, VEC>::store(dxb + i * VEC, acc);
    }
}

// The same finish for a dgrad that ran over the padded domain (conv_lean.hip MODE 2): dxp is [B][D+2][H+2][W+2][Cin_tot]
// with the gradient of padded position op-1 at coordinate op; the reflect adjoint folds the halo planes back:
// plane 1 also receives plane -1, plane S-2 also receives plane S (per axis, so up to 8 terms at the corners).
// streaming accesses of the finish kernels: every byte is touched once (bit 1 of `fold`: non-temporal loads of the gradient scratch,
// bit 2: non-temporal stores of dx, bit 3: non-temporal loads of x)
typedef unsigned int nt_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int nt_u32x2 __attribute__((ext_vector_type(2)));
template <typename T, int VEC> __device__ __forceinline__ void vload_nt(const T* p, float (&v)[VEC], bool nt) {
    if (nt && sizeof(T) == 2 && VEC == 8) {
        const nt_u32x4 u = __builtin_nontemporal_load(reinterpret_cast<const nt_u32x4*>(p));
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(u[j] << 16); v[2 * j + 1] = __uint_as_float(u[j] & 0xffff0000u); }
    } else {
        VecIO<T, VEC>::load(p, v);
    }
}
template <typename T, int VEC> __device__ __forceinline__ void vstore_nt(T* p, const float (&v)[VEC], bool nt) {
    if (nt && sizeof(T) == 2 && VEC == 8) {
        union { bf16_t h[8]; nt_u32x4 u; } t;
#pragma unroll
        for (int j = 0; j < 8; ++j) t.h[j] = (bf16_t)v[j];
        __builtin_nontemporal_store(t.u, reinterpret_cast<nt_u32x4*>(p));
    } else {
        VecIO<T, VEC>::store(p, v);
    }
}

// The final-form marching input gradient (conv_pair.hip SF / conv_march.hip FIN) wrote dx = rstd (g - m1 - x_hat m2) for every
// voxel from the INTERIOR taps and left the halo outputs of the padded domain (positions -1 / S of an axis, already multiplied by
// LeakyReLU' x mask of the voxel they mirror) in the shell of the padded scratch.  The adjoint of reflect padding adds them onto
// positions 1 / S - 2: dx += rstd * (sum of the <= 7 shell voxels that mirror onto this one).  One thread per (target voxel, four
// channels); targets = voxels with a coordinate in {1, S - 2}, enumerated as three disjoint sets (planes, rows, columns); ~6 S^2
// voxels, a few microseconds instead of a three-tensor pass.
__global__ __launch_bounds__(256) void shell_fold_kernel(bf16_t* __restrict__ dx, const bf16_t* __restrict__ dxp, const float* __restrict__ stats,
                                                         int D, int H, int W, int C, long bstride, int Cin_tot, int cbase, int B) {
    const int Wp = pconv::pad8(W + 2), c4n = C / 4;
    const long nD = 2L * H * W, nH = 2L * (D - 2) * W, nW = 2L * (D - 2) * (H - 2), T = nD + nH + nW;
    const long total = (long)B * T * c4n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % c4n) * 4;
        long r = i / c4n;
        const int b = (int)(r / T);
        r -= (long)b * T;
        int d, h, w;
        if (r < nD) { d = r < (long)H * W ? 1 : D - 2; r %= (long)H * W; h = (int)(r / W); w = (int)(r % W); }
        else if (r < nD + nH) {
            r -= nD;
            const int side = (int)(r / ((long)(D - 2) * W)); r %= (long)(D - 2) * W;
            d = (int)(r / W); d = d == 0 ? 0 : d + 1;            // the D - 2 planes that are not targets themselves: 0, 2 .. D - 3, D - 1
            if (d == D - 2) d = D - 1;
            h = side ? H - 2 : 1; w = (int)(r % W);
        } else {
            r -= nD + nH;
            const int side = (int)(r / ((long)(D - 2) * (H - 2))); r %= (long)(D - 2) * (H - 2);
            d = (int)(r / (H - 2)); d = d == 0 ? 0 : d + 1;
            if (d == D - 2) d = D - 1;
            h = (int)(r % (H - 2)); h = h == 0 ? 0 : h + 1;
            if (h == H - 2) h = H - 1;
            w = side ? W - 2 : 1;
        }
        const int md = d == 1 ? -1 : (d == D - 2 ? D : -2), mh = h == 1 ? -1 : (h == H - 2 ? H : -2), mw = w == 1 ? -1 : (w == W - 2 ? W : -2);
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        const bf16_t* sp = dxp + (long)b * (D + 2) * (H + 2) * Wp * Cin_tot + cbase + c;
        for (int ad = 0; ad <= (md > -2 ? 1 : 0); ++ad)
            for (int ah = 0; ah <= (mh > -2 ? 1 : 0); ++ah)
                for (int aw = 0; aw <= (mw > -2 ? 1 : 0); ++aw) {
                    if (!(ad | ah | aw)) continue;
                    const int qd = ad ? md : d, qh = ah ? mh : h, qw = aw ? mw : w;
                    const f32x4 v = load4(sp + (((long)(qd + 1) * (H + 2) + (qh + 1)) * Wp + (qw + 1)) * Cin_tot);
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[j] += v[j];
                }
        bf16_t* o = dx + (long)b * bstride + (((long)d * H + h) * W + w) * C + c;
        f32x4 cur = load4(o);
#pragma unroll
        for (int j = 0; j < 4; ++j) cur[j] += (stats ? stats[((long)b * C + c + j) * 2 + 1] : 1.f) * a[j];      // (raw input: nothing to scale by)
        store4(o, cur);
    }
}

// One (d, h) row of W voxels per loop trip: the row's halo planes are resolved once (wave-uniform flags, no indexed
// local arrays -- those live in scratch memory), only the two W positions 1 and W-2 gather an extra column.
// The segments of a concatenated input share ONE launch (round 4): the scratch holds the gradient with all Cin channels interleaved per
// voxel, so a launch per segment read every line of it once per segment (the decoders' c1 convs: two 8-channel halves of 32-byte
// voxels -- 222 us per finish against 98 for the same bytes in one segment); here a block walks its rows segment after segment, the
// second visit of a line comes from L2.  Segments of one launch have the same channel count (the launcher groups them).
struct FinSeg { void* dx; const void* x; const float* stats; const uint8_t* mask; long bstride; int C, cbase, mask_col, mask_ld, bmod, coff;
                const void* add; const void* add2; };      // add, add2: SegD.dx_add, dx_add2 (contiguous [B][V][C]), folded into the write of dx
struct FinSegs { FinSeg s[PASSION_MAX_SEG]; int n; };
template <typename T, int VEC>
__global__ __launch_bounds__(256) void in_bwd_finish_pad_kernel(FinSegs fs, const T* __restrict__ dxp, const float* __restrict__ sums, int D, int H, int W,
                                                                int Cin_tot, int rep, long rep_stride, float vscale, int dwin_lo, int dwin_hi, FoldJob fj,
                                                                int fold) {
    if (fj.main_x && (int)blockIdx.x >= fj.main_x) { fold_block(fj, (long)(blockIdx.x - fj.main_x) * gridDim.y + blockIdx.y); return; }
    const int gdx = fj.main_x ? fj.main_x : (int)gridDim.x;
    const int bix = (int)blockIdx.x;
    __shared__ float coef[FIN_MAXC][3];
    const int b = blockIdx.y;
    const long V = (long)D * H * W;
    const bool nt_g = (fold & 2) != 0, nt_dx = (fold & 4) != 0, nt_x = (fold & 8) != 0;
    fold &= 1;
    for (int s = 0; s < fs.n; ++s) {
        const FinSeg& q = fs.s[s];
        if (q.stats) finish_coefs(coef + q.coff, q.stats, sums, b, q.bmod > 0 ? b % q.bmod : b, q.C, Cin_tot, q.cbase, rep, rep_stride, vscale / (float)V);
    }
    __syncthreads();
    const int C = fs.s[0].C;
    const int cvn = C / VEC, rowv = W * cvn;
    const int Wp = pconv::pad8(W + 2);                     // row pitch of the padded scratch (conv_lean.hip MODE 2)
    const int tpr = rowv < 256 ? rowv : 256;              // threads that work on one row; 256 / tpr rows in flight per block
    const int rpb = 256 / tpr, rsub = threadIdx.x / tpr, tin = threadIdx.x % tpr;
    const long prow = (long)Wp * Cin_tot, pplane = (long)(H + 2) * prow;
    for (int r0 = bix * rpb; r0 < D * H; r0 += gdx * rpb) {
        const int r = r0 + rsub;
        if (r >= D * H || rsub >= rpb) continue;
        const int d = r / H, h = r % H;
        // source planes of this row: the own one, plus the halo plane folded onto planes 1 / S-2 (reflect adjoint)
        const int dsrc[3] = {d + 1, 0, D + 1}, hsrc[3] = {h + 1, 0, H + 1};
        const bool dok[3] = {true, fold && d == 1, fold && d == D - 2}, hok[3] = {true, fold && h == 1, fold && h == H - 2};   // fold = 0: zero-padded conv
        for (int s = 0; s < fs.n; ++s) {
            const FinSeg& q = fs.s[s];
            const float* stats = q.stats;
            const bool present = q.mask ? (q.mask[(long)b * q.mask_ld + q.mask_col] != 0) : true;
            T* dxb = reinterpret_cast<T*>(q.dx) + (long)b * q.bstride;
            const T* xb = reinterpret_cast<const T*>(q.x) + (long)(q.bmod > 0 ? b % q.bmod : b) * q.bstride;
            const T* pb = dxp + (long)b * (D + 2) * (H + 2) * Wp * Cin_tot + q.cbase;
            const float (*cf)[3] = coef + q.coff;
            for (int i = tin; i < rowv; i += tpr) {
                const int w = i / cvn, c = (i % cvn) * VEC;
                const long e = ((long)r * W + w) * C + c;
                float g[VEC], xv[VEC];
#pragma unroll
                for (int j = 0; j < VEC; ++j) { g[j] = 0.f; xv[j] = 0.f; }
                if (present) {
                    if (stats) vload_nt<T, VEC>(xb + e, xv, nt_x);
                    const bool wlo = fold && (w == 1), whi = fold && (w == W - 2);
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        if (!dok[a]) continue;
#pragma unroll
                        for (int e2 = 0; e2 < 3; ++e2) {
                            if (!hok[e2]) continue;
                            const T* rowp = pb + dsrc[a] * pplane + hsrc[e2] * prow + c;
                            float t[VEC];
                            vload_nt<T, VEC>(rowp + (long)(w + 1) * Cin_tot, t, nt_g);
#pragma unroll
                            for (int j = 0; j < VEC; ++j) g[j] += t[j];
                            if (wlo) {
                                VecIO<T, VEC>::load(rowp, t);
#pragma unroll
                                for (int j = 0; j < VEC; ++j) g[j] += t[j];
                            }
                            if (whi) {
                                VecIO<T, VEC>::load(rowp + (long)(W + 1) * Cin_tot, t);
#pragma unroll
                                for (int j = 0; j < VEC; ++j) g[j] += t[j];
                            }
                        }
                    }
                    if (stats) {                               // else: input was not normalised in the loader, fold only
                        const bool own = d >= dwin_lo && d < dwin_hi;      // depth-sharded halo planes: only the rstd * g term is ours
#pragma unroll
                        for (int j = 0; j < VEC; ++j)
                            g[j] = own ? __builtin_fmaf(cf[c + j][0], g[j], __builtin_fmaf(cf[c + j][1], xv[j], cf[c + j][2])) : cf[c + j][0] * g[j];
                    }
                }
                if (q.add) {                               // the gradient that reached x by another path (the residual branch, the other readers)
                    float av[VEC];
                    VecIO<T, VEC>::load(reinterpret_cast<const T*>(q.add) + (long)b * V * C + e, av);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) g[j] += av[j];
                    if (q.add2) {
                        VecIO<T, VEC>::load(reinterpret_cast<const T*>(q.add2) + (long)b * V * C + e, av);
#pragma unroll
                        for (int j = 0; j < VEC; ++j) g[j] += av[j];
                    }
                }
                vstore_nt<T, VEC>(dxb + e, g, nt_dx);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// wgrad:  dW[co][ci][tap] += sum_vox dY[vox][co] * act(x)[vox*s + tap][ci]
// GEMM with M = out-channel, N = (ci, tap) (tap fastest: the reference weight layout), K = voxels.
// A workgroup owns (a strided set of bricks) x (one CK-channel chunk of one segment) x (MT row tiles);
// the four waves split that chunk's column tiles.  Partial sums stay in registers across bricks and
// are flushed once with f32 atomics.
constexpr int WG_NTW = 7;  // column tiles per wave (4 waves * 7 * 16 >= 27 taps * 16 channels)

template <typename TX, typename TY, int KS, int S, int MT>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(ConvK p, float* __restrict__ dW, float* __restrict__ dbias,
                                                         int chunks_total, int tiles_per_b) {
    wgrad_clear(p);
    using G = Geo<KS, S, false>;
    constexpr int CK = (KS == 1) ? 16 : ((S == 2) ? 4 : 8);
    static_assert(CK == G::CK, "chunk");
    constexpr int NTAP = G::NTAP;
    constexpr int MC = MT * 16;
    __shared__ float lx[CK * G::PLANE];
    constexpr int DYS = (MC % 32 == 0) ? MC + 16 : MC;  // row stride: lane groups kq,kq+1 land on disjoint banks
    __shared__ float ldy[TD * TH * TW * DYS];

    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, n = lane & 15, kq = lane >> 4;
    // blockIdx.y -> (chunk, row group)
    const int mgroups = p.Mpad / 16 / MT;
    const int mg = blockIdx.y % mgroups;
    int chunk = blockIdx.y / mgroups;
    // locate the chunk: segment + channel offset
    int sg = 0, c0 = 0;
    for (int s2 = 0; s2 < p.nseg; ++s2) {
        const int nch = (p.seg[s2].C + CK - 1) / CK;
        if (chunk < nch) { sg = s2; c0 = chunk * CK; break; }
        chunk -= nch;
    }
    const SegD& sd = p.seg[sg];
    const int ck = min(CK, sd.C - c0);
    const int ncols = ck * NTAP;
    const int ntiles_n = (ncols + 15) / 16;

    // this lane's columns: col = nt*16 + n -> (channel, tap) -> LDS offset
    int coloff[WG_NTW];
    bool colok[WG_NTW];
#pragma unroll
    for (int j = 0; j < WG_NTW; ++j) {
        const int nt = wv + 4 * j;
        const int col = nt * 16 + n;
        colok[j] = (nt < ntiles_n) && (col < ncols);
        int c = 0, tap = 0;
        if (colok[j]) { c = col / NTAP; tap = col % NTAP; }
        const int tw = tap % KS, th = (tap / KS) % KS, td = tap / (KS * KS);
        coloff[j] = c * G::PLANE + (td * G::IH + th) * G::IW + tw;
    }
    f32x4 acc[MT][WG_NTW];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < WG_NTW; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 accb[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) accb[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool bias_block = (dbias != nullptr) && (blockIdx.y / mgroups == 0);  // workgroup-uniform
    const bool do_bias = bias_block && (wv == 0);

    const int total_tiles = p.B * tiles_per_b;
    for (int tile = xcd_swizzle(blockIdx.x, gridDim.x); tile < total_tiles; tile += gridDim.x) {
        int t2 = tile;
        const int tw_ = t2 % p.tilesW; t2 /= p.tilesW;
        const int th_ = t2 % p.tilesH; t2 /= p.tilesH;
        const int td_ = t2 % p.tilesD; t2 /= p.tilesD;
        const int b = t2;
        const int o0d = td_ * TD, o0h = th_ * TH, o0w = tw_ * TW;
        float mval = 1.f;
        if (sd.mask) mval = sd.mask[(long)b * sd.mask_ld + sd.mask_col] ? 1.f : 0.f;
        if (mval == 0.f && !bias_block) continue;  // absent modality: no weight gradient from this brick
        __syncthreads();
        const TX* src = reinterpret_cast<const TX*>(sd.x) + (long)sd.bsrc(b) * sd.bstride;
        stage_tile<TX, KS, S, false>(lx, src, 0, sd.C, c0, ck, sd.stats, mval, p.slope, p.pad_mode, sd.bsrc(b), o0d, o0h, o0w,
                                     p.Di, p.Hi, p.Wi, tid);
        // dY brick [vox][MC] (zero outside the volume / beyond Cout)
        const TY* dy = reinterpret_cast<const TY*>(p.dy) + (long)b * p.Do * p.Ho * p.Wo * p.Cout;
        for (int i = tid; i < TD * TH * TW * (MC / 4); i += 256) {
            const int c4 = i % (MC / 4), v = i / (MC / 4);
            const int w = v % TW, h = (v / TW) % TH, d = v / (TW * TH);
            const int od = o0d + d, oh = o0h + h, ow = o0w + w;
            const int co = mg * MC + c4 * 4;
            f32x4 g = {0.f, 0.f, 0.f, 0.f};
            if (od < p.Do && oh < p.Ho && ow < p.Wo && co < p.Cout)
                g = load4(dy + ((((long)od) * p.Ho + oh) * p.Wo + ow) * p.Cout + co);
#pragma unroll
            for (int j = 0; j < 4; ++j) ldy[v * DYS + c4 * 4 + j] = g[j];
        }
        __syncthreads();
        // K loop: 4 consecutive-w voxels per MFMA step
        for (int ks = 0; ks < TD * TH * TW / 4; ++ks) {
            const int v = ks * 4 + kq;
            const int w = v % TW, h = (v / TW) % TH, d = v / (TW * TH);
            const int vpos = ((d * S) * G::IH + h * S) * G::IW + w * S;
            float a[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[mt] = ldy[v * DYS + mt * 16 + n];
#pragma unroll
            for (int j = 0; j < WG_NTW; ++j) {
                if (wv + 4 * j >= ntiles_n) continue;  // wave-uniform
                float bv = colok[j] ? lx[coloff[j] + vpos] : 0.f;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], bv, acc[mt][j], 0, 0, 0);
            }
            if (do_bias) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    accb[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], 1.0f, accb[mt], 0, 0, 0);
            }
        }
    }
    // flush: lane holds rows co = (mg*MT+mt)*16 + kq*4 + r, column n of each of its tiles.  Every workgroup owns
    // slot blockIdx.x of the partial-sum slab (plain stores, no atomics); reduce_parts_kernel folds the slots.
    if (dW) {
        dW += (long)blockIdx.x * p.wslot;
#pragma unroll
        for (int j = 0; j < WG_NTW; ++j) {
            if (!colok[j]) continue;
            const int col = (wv + 4 * j) * 16 + n;
            const int c = col / NTAP, tap = col % NTAP;
            const long ci = sd.cbase + c0 + c;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = (mg * MT + mt) * 16 + kq * 4 + r;
                    const float v = acc[mt][j][r];
                    if (co < p.Cout) dW[((long)co * p.Cin + ci) * NTAP + tap] = v;
                }
        }
    }
    if (do_bias && n == 0) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = (mg * MT + mt) * 16 + kq * 4 + r;
                if (co < p.Cout) dbias[(long)blockIdx.x * p.wslot + co] = accb[mt][r];
            }
    }
}

// out[i] = sum_{r < R} part[r * stride + i], fixed summation order.  256 threads = 16 outputs x 16 slices of r.
// Optionally also clears zero_n floats at zero_ptr (the replicas the following dgrad launch adds into): saves a memset
// dispatch per conv backward.
__global__ __launch_bounds__(256) void reduce_parts_kernel(const float* __restrict__ part, int R, long n, long stride,
                                                           float* __restrict__ out, float* __restrict__ zero_ptr = nullptr,
                                                           long zero_n = 0) {
    for (long z = (long)blockIdx.x * 256 + threadIdx.x; z < zero_n; z += (long)gridDim.x * 256) zero_ptr[z] = 0.f;
    long i;
    bool owner;
    const float t = fold_value(part, stride, R, n, blockIdx.x, i, owner);      // grid = fold_nblocks(n, R)
    if (owner) out[i] = t;
}

// The same fold for a grouped launch (blockIdx.y = weight set): group g's slots are the Rg consecutive ones from g * Rg on, its
// result goes to outs.p[g].
struct RedOuts { float* p[PASSION_MAX_GROUPS]; };
__global__ __launch_bounds__(256) void reduce_parts_grouped_kernel(const float* __restrict__ part, int Rg, long n, long stride, RedOuts outs,
                                                                   float* __restrict__ zero_ptr = nullptr, long zero_n = 0) {
    if (blockIdx.y == 0)
        for (long z = (long)blockIdx.x * 256 + threadIdx.x; z < zero_n; z += (long)gridDim.x * 256) zero_ptr[z] = 0.f;
    long i;
    bool owner;
    const float t = fold_value(part + (long)blockIdx.y * Rg * stride, stride, Rg, n, blockIdx.x, i, owner);
    if (owner) outs.p[blockIdx.y][i] = t;
}

// weight [Cout][Cin][ntap] -> fwd image [tap][Cin][CoutPad] and dgrad image [tap][Cout][CinPad]
__global__ void pack_weight_kernel(const float* __restrict__ w, int Cin, int Cout, int ntap, int CoutPad, int CinPad,
                                   float* __restrict__ fwd, float* __restrict__ dg) {
    const long nf = (long)ntap * Cin * CoutPad, nd = (long)ntap * Cout * CinPad;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nf + nd; i += (long)gridDim.x * blockDim.x) {
        if (i < nf) {
            const int co = (int)(i % CoutPad);
            const int ci = (int)((i / CoutPad) % Cin);
            const int tap = (int)(i / ((long)CoutPad * Cin));
            fwd[i] = (co < Cout) ? w[((long)co * Cin + ci) * ntap + tap] : 0.f;
        } else {
            const long k = i - nf;
            const int ci = (int)(k % CinPad);
            const int co = (int)((k / CinPad) % Cout);
            const int tap = (int)(k / ((long)CinPad * Cout));
            dg[k] = (ci < Cin) ? w[((long)co * Cin + ci) * ntap + tap] : 0.f;
        }
    }
}

int fill_params(const passion_conv_desc_t* d, ConvK& k) {
    if (!d) return -1001;
    k.dbg = 0;
    k.sums_fixed = 0; k.sums_float_only = 0;
    k.zero_ptr = nullptr; k.zero_n = 0;
    k.fst = FwdStats{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0.f, 0.0};
    k.fst_want = nullptr; k.fst_eps = 0.f;
    k.bgroup = 0; k.wg_vec = 0;
    for (int g = 0; g < PASSION_MAX_GROUPS; ++g) k.biasg[g] = nullptr;
    k.bd_rows = 0; k.bd_k8 = 0;
    k.y_f32 = 0;
    k.spos = nullptr; k.pwm = nullptr; k.pw_done = 0; k.march_k8 = 0;
    if (d->ksize != 1 && d->ksize != 3) return -1002;
    if (d->stride != 1 && d->stride != 2) return -1003;
    if (d->ksize == 1 && d->stride != 1) return -1003;
    if (d->nseg < 1 || d->nseg > PASSION_MAX_SEG) return -1004;
    if (d->Cout % 4) return -1005;
    k.B = d->B; k.Di = d->D; k.Hi = d->H; k.Wi = d->W;
    const int pad = d->ksize / 2;
    k.Do = (d->D + 2 * pad - d->ksize) / d->stride + 1;
    k.Ho = (d->H + 2 * pad - d->ksize) / d->stride + 1;
    k.Wo = (d->W + 2 * pad - d->ksize) / d->stride + 1;
    k.pad_mode = d->pad_mode; k.nseg = d->nseg; k.Cout = d->Cout; k.slope = d->lrelu_slope;
    if (d->ksize == 3 && d->pad_mode == PASSION_PAD_REFLECT && (d->D < 2 || d->H < 2 || d->W < 2)) return -1006;
    int cb = 0;
    for (int s = 0; s < PASSION_MAX_SEG; ++s) {
        SegD& o = k.seg[s];
        if (s < d->nseg) {
            const passion_seg_t& i = d->seg[s];
            o.x = i.x; o.stats = i.stats; o.mask = i.mask; o.dx = i.dx; o.bstride = i.bstride; o.C = i.C;
            o.mask_col = i.mask_col; o.mask_ld = i.mask_ld; o.cbase = cb; o.bmod = i.bmod;
            o.dx_add = i.dx ? i.dx_add : nullptr;
            o.dx_add2 = o.dx_add ? i.dx_add2 : nullptr;
            cb += i.C;
        } else {
            o = SegD{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0};
        }
    }
    k.Cin = cb;
    k.bias = nullptr; k.residual = nullptr; k.y = nullptr; k.dy = nullptr; k.sums = nullptr; k.wpack = nullptr;
    const int sin[3] = {d->D, d->H, d->W}, so[3] = {k.Do, k.Ho, k.Wo};
    for (int a = 0; a < 3; ++a) {
        k.hi_t[a] = -1; k.hi_u[a] = 0;
        for (int t = 0; t < d->ksize; ++t) {
            const int u = sin[a] + 1 - t;
            if (u % d->stride == 0 && u / d->stride < so[a] && u / d->stride >= 0) { k.hi_t[a] = t; k.hi_u[a] = u; }
        }
    }
    return 0;
}

template <typename TI, typename TO, int KS, int S, int MODE>
int launch_conv_mt(ConvK& k, int mtiles, hipStream_t st) {
    const int tiles = k.B * k.tilesD * k.tilesH * k.tilesW;
    int MT = mtiles >= 4 ? 4 : mtiles;
    if (mtiles % MT) MT = (mtiles % 2 == 0) ? 2 : 1;
    while (MT > 1 && (long)tiles * (mtiles / MT) < 512) MT >>= 1;
    dim3 grid(tiles, mtiles / MT);
    if (MT == 4) hipLaunchKernelGGL((conv_mfma_kernel<TI, TO, KS, S, 4, MODE>), grid, dim3(256), 0, st, k);
    else if (MT == 2) hipLaunchKernelGGL((conv_mfma_kernel<TI, TO, KS, S, 2, MODE>), grid, dim3(256), 0, st, k);
    else hipLaunchKernelGGL((conv_mfma_kernel<TI, TO, KS, S, 1, MODE>), grid, dim3(256), 0, st, k);
    LAUNCH_RET();
    return 0;
}

// option "pack_f32" (round 6): 0 = passion_conv_pack_weights writes the bf16 operand images only.  A bf16 model never reads the two f32
// images (they are 2/3 of the bytes a re-pack writes and half of its gather reads: the closing pack of the four encoders' weights
// sits between two steps, 94 us); the f32 kernels refuse to run on images that were not packed instead of reading stale ones.
// (pconv::g_pack_f32: conv_bf16.hip)
template <typename TI, typename TO, int MODE>
int launch_conv_geo(ConvK& k, int ksize, int stride, int mtiles, hipStream_t st) {
    if (!pconv::g_pack_f32) return -1024;            // the f32 operand images are not being packed (option pack_f32 = 0)
    passion_prof_tag("conv_mfma_kernel<%s, %s, KS=%d, S=%d, MODE=%d>", sizeof(TI) == 4 ? "float" : "bf16", sizeof(TO) == 4 ? "float" : "bf16",
                     ksize, stride, MODE);
    if (ksize == 1) return launch_conv_mt<TI, TO, 1, 1, MODE>(k, mtiles, st);
    if (stride == 1) return launch_conv_mt<TI, TO, 3, 1, MODE>(k, mtiles, st);
    return launch_conv_mt<TI, TO, 3, 2, MODE>(k, mtiles, st);
}

struct WgPlan { bool bf16; int MT, chunks, gx, gy; };

// grid of the wgrad launch: y = (8/16-channel chunk) x (row-tile group), x = bricks strided over <= 1024/gy workgroups
// the 4-class logit heads: bf16 activations in, f32 logits out (Cout <= 8), 1x1x1 or 3x3x3 stride 1
static bool lean_f32_head(const ConvK& k, const passion_conv_desc_t* d) {
    if (!pconv::g_conv_lean || g_passion_force_f32_mfma) return false;
    if (d->x_dtype != PASSION_BF16 || d->y_dtype != PASSION_F32 || d->stride != 1 || k.Cout > 8 || k.Cout % 4) return false;
    int K8 = 0;
    for (int s = 0; s < k.nseg; ++s) { if (k.seg[s].C % 8) return false; K8 += k.seg[s].C / 8; }
    return K8 * 8 == k.Cin && K8 <= 32;
}

// voxels per launch from which the finest 3x3x3 layers take the finish-free backward (option conv_marchfinal = 2).  2^22 until the
// weight gradient with the indicator slab and the final-form input gradient got faster in round 6; re-measured then: 2 x 10^6 (one
// 128^3 sample, four 80^3 ones) plain step 9.11 -> 9.03 ms, 80^3 step 5.69 -> 5.58, RFNet +0.8 %, m2ftrans +0.7 %, PASSION +-0;
// 5 x 10^5 and "wherever it applies" the same to the noise
static long marchfinal_minvox() { static const long v = []() { const char* e = getenv("PASSION_MARCHFINAL_MINVOX"); return e && atol(e) > 0 ? atol(e) : 2000000L; }(); return v; }
WgPlan plan_wgrad(const ConvK& k, const passion_conv_desc_t* d) {
    WgPlan w;
    w.bf16 = (d->x_dtype == PASSION_BF16 && d->y_dtype == PASSION_BF16 && bf16_eligible(k, false) && !g_passion_force_f32_mfma) ||
             lean_f32_head(k, d);
    const int CK = w.bf16 ? wgrad_bf16_chunk(k, d->ksize, d->stride) : (d->ksize == 1 ? 16 : (d->stride == 2 ? 4 : 8));
    w.chunks = 0;
    if (w.bf16) {   // chunks run over the 8-channel blocks of the whole virtual concat
        int k8 = 0;
        for (int s = 0; s < k.nseg; ++s) k8 += (k.seg[s].C + 7) / 8;
        w.chunks = (k8 + CK / 8 - 1) / (CK / 8);
    } else {
        for (int s = 0; s < k.nseg; ++s) w.chunks += (k.seg[s].C + CK - 1) / CK;
    }
    const int mtiles = pad16(k.Cout) / 16;
    w.MT = w.bf16 ? ((mtiles % 2 == 0) ? 2 : 1) : ((mtiles % 2 == 0 && d->stride == 1) ? 2 : 1);
    w.gy = w.chunks * (mtiles / w.MT);
    const long total = (long)k.B * cdiv(k.Do, TD) * cdiv(k.Ho, TH) * cdiv(k.Wo, TW);
    // workgroups of a launch = slots of the partial slab x gy.  Small volumes (levels 3..5: <= 512 bricks) take fewer slots: the
    // slab (slots x Cout Cin taps floats: 28 MB at 1024 / gy slots for a 64 -> 64 conv over 4 x 16^3) is written, then read by the fold
    // (measured per layer, weight gradient + the finish launch that folds its slab, 4 samples: 64 -> 64 over 16^3: 56.1 us at 1024 workgroups,
    // 42.7 at 512, 40.0 at 256, 47.3 at 128; 128 -> 128 over 8^3: 39.2 / 39.2 / 34.4; 32 -> 32 over 32^3: 62.8 / 54.6 / 58.3 / 75.9)
    static const long wg_small = []() { const char* e = getenv("PASSION_WGRAD_WGS_SMALL"); return e && atol(e) > 0 ? atol(e) : 0L; }();
    // large volumes: 512 slots.  The marching weight gradients hold two workgroups per CU, so 512 workgroups are ONE resident round, and
    // the slab to fold is half as long (round 6, per launch over 4 x 128^3 / 4 x 64^3 against 1024 slots: indicator kernel 8 -> 8 115.7 ->
    // 105.9 us, 16 -> 8 194.5 -> 183.9 (768: 204 / 295, the (tile, depth segment) units do not divide; 1536: 155 / 318; 2048: 133 / 212);
    // wgrad_s2m_kernel 53.5 -> 45.6; wgrad_march16_kernel 16 -> 16 41.9 -> 34.0, 32 -> 16 62.2 -> 54.4; the streaming pointwise and the
    // brick kernels do not care; PASSION step +0.7 %)
    static const long wg_big = []() { const char* e = getenv("PASSION_WGRAD_WGS"); return e && atol(e) > 0 ? atol(e) : 512L; }();
    const bool finest = d->ksize == 3 && d->stride == 1 && k.Cout == 8 && k.Cin <= 16 && (long)k.B * k.Di * k.Hi * k.Wi >= marchfinal_minvox();
    const long wg_target = total <= 512 ? (wg_small ? wg_small : (total <= 64 ? 256L : 512L)) : wg_big;
    w.gx = (int)hmin(total, hmax(1, wg_target / w.gy));
    // 1x1x1 on the streaming kernel (conv_wgrad_pw.hip): one workgroup handles every input block of its voxels, so the launch has gx
    // workgroups, not gx x gy -- give it the whole target
    if (d->ksize == 1 && d->stride == 1 && w.bf16 && w.MT == mtiles && pconv::wgrad_pw_covers(k, d->y_dtype == PASSION_F32))
    {
        // ... and one or two input blocks leave room for six wavefronts per SIMD (conv_wgrad_pw.hip launches as many workgroups of
        // these slots as are resident at once)
        w.gx = (int)hmin(total, (k.Cin <= 16 && total > 512 ? 2 : 1) * wg_target);
    }
    // a whole number of slots per sample: the finish-free backward folds the slab per sample (march_ident_kernel, pw_ident_kernel); 1024
    // slots over the 5- and 10-sample virtual batches of the PASSION passes kept those launches on the dgrad + finish path
    // (only where that path can run: the weight gradients of every other launch keep their slot count -- and their f32 summation order)
    if (w.gx > k.B && finest) w.gx -= w.gx % k.B;
    return w;
}

template <typename TX, typename TY, int KS, int S>
int launch_wgrad_mt(ConvK& k, const WgPlan& w, float* dW, float* db, hipStream_t st) {
    const int tiles_per_b = k.tilesD * k.tilesH * k.tilesW;
    dim3 grid(w.gx, w.gy);
    passion_prof_tag("conv_wgrad_kernel<%s, %s, %d, %d, %d>", sizeof(TX) == 4 ? "float" : "bf16", sizeof(TY) == 4 ? "float" : "bf16", KS, S, w.MT);
    if (w.MT == 2) {
        if constexpr (S == 1) hipLaunchKernelGGL((conv_wgrad_kernel<TX, TY, KS, S, 2>), grid, dim3(256), 0, st, k, dW, db, w.chunks, tiles_per_b);
    } else hipLaunchKernelGGL((conv_wgrad_kernel<TX, TY, KS, S, 1>), grid, dim3(256), 0, st, k, dW, db, w.chunks, tiles_per_b);
    LAUNCH_RET();
    return 0;
}

template <typename TX, typename TY>
int launch_wgrad_geo(ConvK& k, const WgPlan& w, int ksize, int stride, float* dW, float* db, hipStream_t st) {
    if (ksize == 1) return launch_wgrad_mt<TX, TY, 1, 1>(k, w, dW, db, st);
    if (stride == 1) return launch_wgrad_mt<TX, TY, 3, 1>(k, w, dW, db, st);
    return launch_wgrad_mt<TX, TY, 3, 2>(k, w, dW, db, st);
}

long f32_pack_floats(int Cin, int Cout, int nt) { return (long)nt * Cin * pad16(Cout) + (long)nt * Cout * pad16(Cin); }

double conv_bytes(const ConvK& k, int xs, int ys) {
    const double vin = (double)k.B * k.Di * k.Hi * k.Wi, vout = (double)k.B * k.Do * k.Ho * k.Wo;
    return vin * k.Cin * xs + vout * k.Cout * ys;
}

}  // namespace

// option "conv_marchfinal" / PASSION_CONV_MARCHFINAL: the finish-free backward of the finest 3x3x3 layers (below): 0 off, 1 wherever it
// applies, 2 (default) from 2^22 voxels per launch up.  Exact (tests/test_ops_gpu.py::test_finish_free_marching_backward_equals_dgrad_plus_finish).
// Measured in isolation on 4 x 128^3, 8 -> 8 it trades the 101 us finish pass for +57 us in the weight gradient (the indicator tiles
// double an LDS-bound MFMA phase), +28 us of folds and +11 us in the input gradient: 351 against 355 us per layer -- even, and round 4
// left it off.  In the recorded step (round 5) it is ahead: the finish pass is the one kernel of this backward that runs ALONE on
// the chip (1.15 ms of a 10.3 ms step), the weight gradient it loads instead runs beside the other streams: plain step +1.2 %,
// PASSION +0.8 %, RFNet / m2ftrans +-0.  Small launches lose (150 against 120 us at one 128^3 sample; the 80^3 step -1.4 %): hence
// the threshold.
namespace pconv { int g_conv_padout = 1; int g_conv_marchfinal = 2; }   // PASSION_CONV_PADOUT=0: reflect adjoint through mirror taps everywhere (A/B)

// dgrad over the padded domain (conv_lean.hip MODE 2) + folding finish kernel: 3x3x3 stride-1 reflect-padded bf16 convs
// (the finish kernel folds the halo planes; for inputs normalised in the loader it also completes InstanceNorm backward)
static bool use_padout(const ConvK& k, const passion_conv_desc_t* d) {
    if (!pconv::g_conv_lean || !pconv::g_conv_padout || g_passion_force_f32_mfma) return false;
    if (d->ksize != 3) return false;
    if (pconv::g_conv_padout > 1 && k.Di >= pconv::g_conv_padout) return false;
    if (d->x_dtype != PASSION_BF16 || d->y_dtype != PASSION_BF16) return false;
    if (d->pad_mode != PASSION_PAD_REFLECT) {
        // zero padding: only where the marching kernel runs the padded-domain dgrad (it skips the halo outputs; the lean MODE 2
        // kernel would count them in the InstanceNorm-backward sums) -- the fusion blocks' 3x3x3 convs at the two finest levels
        if (d->pad_mode != PASSION_PAD_ZEROS || !pconv::conv_march_covers(k, d->ksize, d->stride, 2, nullptr)) return false;
    }
    if (k.Di < 2 || k.Hi < 2 || k.Wi < 2 || k.Cin % 4 || !pconv::lean_covers(k, d->stride, true)) return false;
    if (d->stride == 2 && ((k.Di | k.Hi | k.Wi) & 1)) return false;      // even input sizes: Do = Di / 2 exactly
    bool any = false;
    for (int s = 0; s < k.nseg; ++s)
        if (k.seg[s].dx) { any = true; if (k.seg[s].bmod > 0) return false; }    // batched passes: summed by the plain finish kernel
    return any;
}
// Batched passes through a 1x1x1 conv (the fusion blocks' first conv, reference blocks.py:536, run as teacher + four
// students): InstanceNorm backward and the conv transpose are linear in dY and the only per-pass quantity is the 0/1
// modality mask, so  sum_p dX_p[segment i] = Backward_i( sum_p mask[p][i] dY_p ).  The P masked sums (one per segment) are
// laid side by side as a [bmod][V][nseg * Cout] tensor and ONE dgrad runs over the physical samples with block-diagonal
// weights (segment i's rows see only its own Cout-block): dX is written once instead of P times and re-read P times.
namespace pconv { int g_conv_blockdiag = 1; }    // PASSION_CONV_BLOCKDIAG=0: per-pass gradients + summing finish (A/B)
static bool use_blockdiag(const ConvK& k, const passion_conv_desc_t* d) {
    if (!pconv::g_conv_lean || !pconv::g_conv_blockdiag || g_passion_force_f32_mfma) return false;
    if (d->ksize != 1 || d->x_dtype != PASSION_BF16 || d->y_dtype != PASSION_BF16 || k.Cout % 8) return false;
    const int bm = k.seg[0].bmod, cs = k.seg[0].C;
    if (bm <= 0 || k.B % bm || k.B / bm < 2 || cs % 8) return false;
    for (int s = 0; s < k.nseg; ++s)
        if (!k.seg[s].dx || k.seg[s].bmod != bm || k.seg[s].C != cs || k.seg[s].cbase != s * cs) return false;
    return (long)k.nseg * k.Cout / 8 <= 32;
}
static long blockdiag_bytes(const ConvK& k) {     // bf16 [bmod][V][nseg * Cout]
    return ((long)k.seg[0].bmod * k.Di * k.Hi * k.Wi * k.nseg * k.Cout * 2 + 15) / 16 * 16;
}
struct PassSumArgs { const uint8_t* mask[PASSION_MAX_SEG]; int col[PASSION_MAX_SEG], ld[PASSION_MAX_SEG]; };
// one thread per (sample b, voxel v, 8-channel vector c8 of Cout): out[b][v][i * Cout + c8 * 8 ..] = sum_p mask[p*bm+b][i] dy[p*bm+b][v][..]
__global__ __launch_bounds__(256) void pass_sum_kernel(const bf16_t* __restrict__ dy, bf16_t* __restrict__ out, PassSumArgs a, int P, int bm,
                                                       long V, int Cout, int nseg) {
    const int c8n = Cout / 8;
    const long total = (long)bm * V * c8n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c8 = (int)(i % c8n);
        const long bv = i / c8n;
        const int b = (int)(bv / V);
        float acc[PASSION_MAX_SEG][8];
#pragma unroll
        for (int s2 = 0; s2 < PASSION_MAX_SEG; ++s2)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[s2][j] = 0.f;
        for (int p = 0; p < P; ++p) {
            float v[8];
            VecIO<bf16_t, 8>::load(dy + (((long)p * bm) * V + bv) * Cout + c8 * 8, v);
#pragma unroll
            for (int s2 = 0; s2 < PASSION_MAX_SEG; ++s2) {
                if (s2 >= nseg) continue;
                const bool on = !a.mask[s2] || a.mask[s2][((long)p * bm + b) * a.ld[s2] + a.col[s2]] != 0;
                if (on) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[s2][j] += v[j];
                }
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < PASSION_MAX_SEG; ++s2)
            if (s2 < nseg) VecIO<bf16_t, 8>::store(out + (bv * nseg + s2) * Cout + c8 * 8, acc[s2]);
    }
}

// per-pass gradients of the batched-pass segments (SegD.bmod): [B virtual][V][C] each, summed over the passes by the finish
static long pass_scratch_floats(const ConvK& k, const passion_conv_desc_t* d) {
    if (use_blockdiag(k, d)) return blockdiag_bytes(k) / 4;
    const long es = d->x_dtype == PASSION_BF16 ? 2 : 4;
    long bytes = 0;
    for (int s = 0; s < k.nseg; ++s)
        if (k.seg[s].dx && k.seg[s].bmod > 0) bytes += ((long)k.B * k.Di * k.Hi * k.Wi * k.seg[s].C * es + 15) / 16 * 16;
    return bytes / 4;
}
// scratch layout (floats): [InstanceNorm-backward sums | wgrad slab] then, from a 128-byte boundary, [padded-domain gradient]
// [per-pass gradients or the masked pass sums]
// pointwise layers (nt == 1): + the indicator slab [gx][Cout Cin], the per-sample folds [B][2 Cout Cin + Cout] and the means
// [B][Cin][2] of the finish-free backward (conv_pw.hip)
static long pw_extra_floats(const ConvK& k, int gx, int nt) {
    if (nt == 27 && k.Cout == 8 && k.Cin <= 16) {      // 3x3x3 layers of the finest level: indicator slab (slot stride = the weight slab's),
        const long nW = (long)k.Cout * k.Cin * 27;      // per-sample folds [B][2 nW + Cout], means [B][Cin][2] (march_final below)
        return ((long)gx * (nW + k.Cout) + (long)k.B * (2 * nW + k.Cout) + (long)k.B * k.Cin * 2 + 31) / 32 * 32;
    }
    if (nt != 1) return 0;
    const long nW = (long)k.Cout * k.Cin;
    return ((long)gx * nW + (long)k.B * (2 * nW + k.Cout) + (long)k.B * k.Cin * 2 + 31) / 32 * 32;
}
static long slab_end_floats(const ConvK& k, int gx, int nt) {
    const long n = (long)2 * k.B * k.Cin * SUMS_SLOTS + (long)gx * ((long)k.Cout * k.Cin * nt + k.Cout);
    return (n + 31) / 32 * 32 + pw_extra_floats(k, gx, nt);
}
static long padout_floats(const ConvK& k) {     // bf16 [B][D+2][H+2][W+2][Cin]
    return (((long)k.B * (k.Di + 2) * (k.Hi + 2) * pad8(k.Wi + 2) * k.Cin + 1) / 2 + 4 + 31) / 32 * 32;      // rows padded to 8 voxels
}

extern "C" int64_t passion_conv_bwd_scratch_floats(const passion_conv_desc_t* d) {
    ConvK k;
    if (fill_params(d, k)) return -1;
    const WgPlan w = plan_wgrad(k, d);
    const int64_t nt = (int64_t)d->ksize * d->ksize * d->ksize;
    return slab_end_floats(k, w.gx, (int)nt) + (use_padout(k, d) ? padout_floats(k) : 0) + pass_scratch_floats(k, d);
}

extern "C" int64_t passion_conv_pack_floats(int Cin, int Cout, int ksize) {
    const int64_t nt = (int64_t)ksize * ksize * ksize;
    // [f32 fwd image | f32 dgrad image | bf16 fwd + dgrad images]
    return nt * Cin * pad16(Cout) + nt * Cout * pad16(Cin) + (bf16_pack_elems(Cin, Cout, (int)nt) + 1) / 2;
}

extern "C" int passion_conv_pack_weight(const float* weight, int Cin, int Cout, int ksize, float* packed, void* stream) {
    const float* ws[1] = {weight};
    float* outs[1] = {packed};
    return passion_conv_pack_weights(1, ws, &Cin, &Cout, &ksize, outs, stream);
}

// Every conv weight is re-packed once per optimizer step: ~110 tensors, a few microseconds of work each -- as
// separate launches that is 220 dispatches per step, so up to PACK_MAX weights share one launch (blockIdx.y = weight).
constexpr int PACK_MAX = 48;
struct PackEnt { const float* w; float* out; int Cin, Cout, ntap, pad_; };
struct PackTable { PackEnt e[PACK_MAX]; };

__global__ __launch_bounds__(256) void pack_multi_kernel(PackTable t) {
    // 32-bit index arithmetic throughout (an image has < 2^31 elements: checked by the launcher): the 64-bit divisions by run-time
    // values of the first version were the kernel -- 105 us for the 24 M elements of the four encoders' images behind the closing
    // optimizer pass, on the critical stream
    const PackEnt& q = t.e[blockIdx.y];
    const bool want_f32 = q.pad_ != 0;             // (PackEnt.pad_: 1 = all four images, 0 = the bf16 ones only)
    const unsigned Cin = q.Cin, Cout = q.Cout, ntap = q.ntap;
    const unsigned CoutPad = pad16(Cout), CinPad = pad16(Cin);
    const unsigned nf = ntap * Cin * CoutPad, nd = ntap * Cout * CinPad;        // f32 images
    const unsigned ci8 = pad8(Cin), co8 = pad8(Cout);
    const unsigned bf = ntap * ci8 * CoutPad, bd = ntap * co8 * CinPad;         // bf16 images (follow the f32 ones)
    float* fwd = q.out;
    float* dg = q.out + nf;
    bf16_t* bfwd = reinterpret_cast<bf16_t*>(q.out + nf + nd);
    bf16_t* bdg = bfwd + bf;
    const float* w = q.w;
    const unsigned total = nf + nd + bf + bd;
    for (unsigned i = (want_f32 ? 0u : nf + nd) + blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        if (i < nf) {                       // [tap][Cin][CoutPad]
            const unsigned co = i % CoutPad, r = i / CoutPad;
            const unsigned ci = r % Cin, tap = r / Cin;
            fwd[i] = (co < Cout) ? w[(co * Cin + ci) * ntap + tap] : 0.f;
        } else if (i < nf + nd) {           // [tap][Cout][CinPad]
            const unsigned k = i - nf;
            const unsigned ci = k % CinPad, r = k / CinPad;
            const unsigned co = r % Cout, tap = r / Cout;
            dg[k] = (ci < Cin) ? w[(co * Cin + ci) * ntap + tap] : 0.f;
        } else if (i < nf + nd + bf) {      // bf16 [tap][CinPad8/8][CoutPad16][8]
            const unsigned k = i - nf - nd;
            const unsigned e = k & 7u, r = k >> 3;
            const unsigned co = r % CoutPad, r2 = r / CoutPad;
            const unsigned nb = ci8 >> 3;
            const unsigned cb = r2 % nb, tap = r2 / nb;
            const unsigned ci = cb * 8 + e;
            bfwd[k] = (bf16_t)((co < Cout && ci < Cin) ? w[(co * Cin + ci) * ntap + tap] : 0.f);
        } else {                            // bf16 [tap][CoutPad8/8][CinPad16][8]
            const unsigned k = i - nf - nd - bf;
            const unsigned e = k & 7u, r = k >> 3;
            const unsigned ci = r % CinPad, r2 = r / CinPad;
            const unsigned nb = co8 >> 3;
            const unsigned cb = r2 % nb, tap = r2 / nb;
            const unsigned co = cb * 8 + e;
            bdg[k] = (bf16_t)((co < Cout && ci < Cin) ? w[(co * Cin + ci) * ntap + tap] : 0.f);
        }
    }
}

extern "C" int passion_conv_pack_weights(int n, const float* const* weights, const int* Cin, const int* Cout, const int* ksize,
                                         float* const* packed, void* stream) {
    for (int i0 = 0; i0 < n; i0 += PACK_MAX) {
        PackTable t;
        const int m = (n - i0 < PACK_MAX) ? n - i0 : PACK_MAX;
        for (int i = 0; i < m; ++i) {
            const int ks = ksize[i0 + i];
            if (ks != 1 && ks != 3) return -1002;
            if (4L * ks * ks * ks * pad16(Cin[i0 + i]) * pad16(Cout[i0 + i]) >= (1L << 31)) return -1002;      // 32-bit indices in the kernel
            t.e[i] = PackEnt{weights[i0 + i], packed[i0 + i], Cin[i0 + i], Cout[i0 + i], ks * ks * ks, pconv::g_pack_f32 ? 1 : 0};
        }
        hipLaunchKernelGGL(pack_multi_kernel, dim3(192, m), dim3(256), 0, (hipStream_t)stream, t);
        LAUNCH_RET();
    }
    return 0;
}

// A grouped launch (include/passion_hip.h, passion_conv3d_fwd_grouped): G weight sets, sample b uses set b / (B / G).  `stride`
// = floats between the packed workspaces of consecutive sets; bias[g] / dW[g] = that set's bias / [dW | dbias] output.
struct GroupArgs { int G; long stride; const float* const* bias; float* const* dW; };

static void set_groups(ConvK& k, const GroupArgs* ga) {
    if (!ga) return;
    k.bgroup = k.B / ga->G;
    k.wg_vec = ga->stride / 4;                     // floats -> 16-byte vectors of the bf16 images
    for (int g = 0; g < ga->G; ++g) k.biasg[g] = ga->bias ? ga->bias[g] : nullptr;
}

static int conv3d_fwd_impl(const passion_conv_desc_t* d, const float* packed, const float* bias, const void* residual, void* y,
                           float* stats_out, float eps, void* stream, const GroupArgs* ga = nullptr) {
    ConvK k;
    int rc = fill_params(d, k);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    set_groups(k, ga);
    k.wpack = packed; k.Mpad = pad16(k.Cout); k.bias = bias; k.residual = residual; k.y = y;
    k.fst_want = stats_out; k.fst_eps = eps;
    k.tilesD = cdiv(k.Do, TD); k.tilesH = cdiv(k.Ho, TH); k.tilesW = cdiv(k.Wo, TW);
    const int mtiles = k.Mpad / 16;
    const int xs = d->x_dtype == PASSION_BF16 ? 2 : 4, ys = d->y_dtype == PASSION_BF16 ? 2 : 4;
    const int nt = d->ksize * d->ksize * d->ksize;
    {
        ProfScope prof(d->ksize == 3 ? 0 : 1, conv_bytes(k, xs, ys) + (double)nt * k.Cin * k.Cout * 4,
                       2.0 * k.B * k.Do * k.Ho * k.Wo * (double)k.Cout * k.Cin * nt, st);
        if (d->x_dtype == PASSION_F32 && d->y_dtype == PASSION_F32)
            rc = launch_conv_geo<float, float, 0>(k, d->ksize, d->stride, mtiles, st);
        else if (d->x_dtype == PASSION_BF16 && d->y_dtype == PASSION_BF16) {
            if (bf16_eligible(k, false) && !g_passion_force_f32_mfma)
                rc = launch_conv_bf16(k, d->ksize, d->stride, false, packed + f32_pack_floats(k.Cin, k.Cout, nt), st);
            else
                rc = launch_conv_geo<bf16_t, bf16_t, 0>(k, d->ksize, d->stride, mtiles, st);
        } else if (d->x_dtype == PASSION_BF16 && d->y_dtype == PASSION_F32) {
            rc = 1;
            if (lean_f32_head(k, d) && !residual) {
                k.y_f32 = 1;
                rc = launch_conv_lean(k, d->ksize, d->stride, 0, packed + f32_pack_floats(k.Cin, k.Cout, nt), st);
                if (rc > 0) k.y_f32 = 0;
            }
            if (rc > 0) rc = launch_conv_geo<bf16_t, float, 0>(k, d->ksize, d->stride, mtiles, st);
        }
        else
            rc = -1007;
    }
    if (rc || !stats_out || k.fst.stats) return rc;      // k.fst.stats: the conv kernel itself left the statistics
    // kernels without the fused epilogue (f32 storage, stride 2, ragged bricks, the general bf16 kernel): the statistics
    // kernels of norm.hip over y, scratch from the stream's region
    const int64_t V = (int64_t)k.Do * k.Ho * k.Wo;
    void* plain = nullptr;
    rc = stream_scratch(st, 0, (size_t)passion_instnorm_scratch_doubles(k.B, V, k.Cout) * sizeof(double), nullptr, &plain);
    if (rc) return rc;
    return passion_instnorm_stats(y, d->y_dtype, k.B, V, k.Cout, V * k.Cout, eps, static_cast<double*>(plain), stats_out, stream);
}

extern "C" int passion_conv3d_fwd(const passion_conv_desc_t* d, const float* packed, const float* bias,
                                  const void* residual, void* y, void* stream) {
    return conv3d_fwd_impl(d, packed, bias, residual, y, nullptr, 0.f, stream);
}

extern "C" int passion_conv3d_fwd_stats(const passion_conv_desc_t* d, const float* packed, const float* bias,
                                        const void* residual, void* y, float* stats, float eps, void* stream) {
    if (!stats) return -1015;
    return conv3d_fwd_impl(d, packed, bias, residual, y, stats, eps, stream);
}

// wgrad and dgrad of one conv only share their inputs (x, dY): the weight-gradient kernels + fold go to a companion stream
// of the caller's stream and run under the dgrad + InstanceNorm-finish kernels; the caller's stream waits for them before
// the call returns, so nothing changes for the caller (ordering, buffer lifetimes).  Experimental: PASSION_WGRAD_FORK=1 /
// option "wgrad_fork".
// Round 4, under graph replay (where the fork would cost the host nothing): a conv whose backward runs on the capture's ORIGIN stream
// forks and replays correctly (tools/fork_capture_probe.py conv1 / conv3 / the whole model with its side streams off), the same conv on
// a torch side stream crashes inside torch's capture_end (hipStreamEndCapture / instantiate) -- with the companions joined to the
// origin beforehand too, and with the forked section left empty; a stand-alone HIP program with the same stream / event pattern
// (tools/capture_nest_probe.hip: nested forks, re-used events, memset nodes) does not.  Unresolved; off.
namespace pconv { int g_wgrad_fork = 0; }   // measured: no gain on the 128^3 step (42.6 vs 42.4 vol/s) -> off by default
namespace {
struct Companion { hipStream_t caller, side; hipEvent_t fork, done; };
std::vector<Companion> g_comp;
std::mutex g_comp_mu;
// the dependency set of a capturing stream (the nodes its next launch would depend on); false = the stream is not capturing
bool capture_tail(hipStream_t st, std::vector<hipGraphNode_t>& out) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    hipGraph_t graph = nullptr;
    const hipGraphNode_t* deps = nullptr;
    size_t n = 0;
    if (hipStreamGetCaptureInfo_v2(st, &cs, &id, &graph, &deps, &n) != hipSuccess || cs != hipStreamCaptureStatusActive) return false;
    out.assign(deps, deps + n);
    return true;
}
Companion* companion(hipStream_t caller) {
    std::lock_guard<std::mutex> lk(g_comp_mu);
    for (auto& c : g_comp) if (c.caller == caller) return &c;
    if (g_comp.size() >= 16) return nullptr;
    Companion c{caller, nullptr, nullptr, nullptr};
    if (hipStreamCreateWithFlags(&c.side, hipStreamNonBlocking) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&c.fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c.done, hipEventDisableTiming) != hipSuccess) return nullptr;
    g_comp.reserve(16);
    g_comp.push_back(c);
    return &g_comp.back();
}
}  // namespace

// Grouped stride-2 3x3x3 layers (bf16, more than one input channel): their weight gradient has no lean kernel (stride 1 only); the
// general bf16 kernel takes the stacked batch when every weight set can have the same whole number of slab slots (option
// "wgrad_group_s2" / PASSION_WGRAD_GROUP_S2=0: one launch + fold per set as before round 6).
namespace pconv { int g_wgrad_group_s2 = 1; }
static bool grouped_general_wgrad(const ConvK& k, const passion_conv_desc_t* d, const WgPlan& w, int G) {
    return pconv::g_wgrad_group_s2 && G > 1 && d->ksize == 3 && d->stride == 2 && w.bf16 && k.Cin > 1 && !g_passion_force_f32_mfma &&
           d->x_dtype == PASSION_BF16 && d->y_dtype == PASSION_BF16 && wgrad_bf16_chunk(k, d->ksize, d->stride) == 8 && w.gx >= G;
}

// phase 0: the whole backward.  Depth-sharded volumes (one volume split along D over the ranks) need the InstanceNorm-backward
// sums of ALL ranks before the gradient can be finished: phase 1 stops after the dgrad with the sums folded into
// sums[0 .. 2 B Cin) (the caller all-reduces them), phase 2 runs only the finish with 1 / V scaled by vscale = V_local / V_global
// and the mean-subtraction terms restricted to the depth planes [dwin_lo, dwin_hi) the rank owns.
// ga / parts: grouped launches (passion_conv3d_bwd_grouped).  parts bit 0: weight (+ bias) gradient, bit 1: input gradient.
static int conv3d_bwd_core(const passion_conv_desc_t* d, const float* packed, const void* dy, float* dweight,
                           float* dbias, float* sums, void* stream, int phase, float vscale, int dwin_lo, int dwin_hi,
                           const GroupArgs* ga, int parts, unsigned* add_fused);

// dx += add over [B][V][C] (dx with its batch stride, add contiguous): what is left of passion_seg_t.dx_add when the path a conv
// backward took has no kernel that folds it into the write of dx
template <typename T, int VEC>
__global__ __launch_bounds__(256) void dx_add_kernel(T* __restrict__ dx, const T* __restrict__ add, long per, long bstride) {
    const long b = blockIdx.y;
    T* o = dx + b * bstride;
    const T* a = add + b * per;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * VEC; i < per; i += (long)gridDim.x * 256 * VEC) {
        if (i + VEC <= per) {
            float u[VEC], v[VEC];
            VecIO<T, VEC>::load(o + i, u);
            VecIO<T, VEC>::load(a + i, v);
#pragma unroll
            for (int j = 0; j < VEC; ++j) u[j] += v[j];
            VecIO<T, VEC>::store(o + i, u);
        } else {
            for (long j = i; j < per; ++j) o[j] = (T)((float)o[j] + (float)a[j]);
        }
    }
}

static int conv3d_bwd_impl(const passion_conv_desc_t* d, const float* packed, const void* dy, float* dweight,
                           float* dbias, float* sums, void* stream, int phase, float vscale, int dwin_lo, int dwin_hi,
                           const GroupArgs* ga = nullptr, int parts = 3) {
    unsigned fused = 0u;
    const int rc = conv3d_bwd_core(d, packed, dy, dweight, dbias, sums, stream, phase, vscale, dwin_lo, dwin_hi, ga, parts, &fused);
    if (rc || !d || phase == 1 || !(parts & 2)) return rc;
    for (int s = 0; s < d->nseg; ++s) {
        const passion_seg_t& sg = d->seg[s];
        if (!sg.dx || !sg.dx_add) continue;
        if (sg.bmod > 0) return -1021;
        const long per = (long)d->D * d->H * d->W * sg.C;
        const dim3 grid((unsigned)hmin((per / 8 + 255) / 256 + 1, 2048), d->B);
        for (int which = 0; which < 2; ++which) {          // bit s: dx_add of segment s was folded into the write of dx; bit 8 + s: dx_add2
            const void* add = which ? sg.dx_add2 : sg.dx_add;
            if (!add || ((fused >> (which * 8 + s)) & 1u)) continue;
            if (d->x_dtype == PASSION_BF16)
                hipLaunchKernelGGL((dx_add_kernel<bf16_t, 8>), grid, dim3(256), 0, (hipStream_t)stream, (bf16_t*)sg.dx, (const bf16_t*)add, per, (long)sg.bstride);
            else
                hipLaunchKernelGGL((dx_add_kernel<float, 4>), grid, dim3(256), 0, (hipStream_t)stream, (float*)sg.dx, (const float*)add, per, (long)sg.bstride);
            LAUNCH_RET();
        }
    }
    return 0;
}

static int conv3d_bwd_core(const passion_conv_desc_t* d, const float* packed, const void* dy, float* dweight,
                           float* dbias, float* sums, void* stream, int phase, float vscale, int dwin_lo, int dwin_hi,
                           const GroupArgs* ga, int parts, unsigned* add_fused) {
    ConvK k;
    int rc = fill_params(d, k);
    if (rc) return rc;
    set_groups(k, ga);
    if (ga) { dweight = ga->dW ? ga->dW[0] : nullptr; dbias = dweight ? dweight + (long)k.Cout * k.Cin * d->ksize * d->ksize * d->ksize : nullptr; }
    if (phase != 0) {
        for (int s = 0; s < k.nseg; ++s) if (k.seg[s].bmod > 0) return -1019;       // batched passes are not sharded
    }
    hipStream_t st = (hipStream_t)stream;
    const int nt = d->ksize * d->ksize * d->ksize;
    const int xs = d->x_dtype == PASSION_BF16 ? 2 : 4, ys = d->y_dtype == PASSION_BF16 ? 2 : 4;
    k.dy = dy;
    k.tilesD = cdiv(k.Do, TD); k.tilesH = cdiv(k.Ho, TH); k.tilesW = cdiv(k.Wo, TW);
    bool need_dx = false, need_norm = false, sums_cleared = false;
    for (int s = 0; s < k.nseg; ++s) {
        if (k.seg[s].dx) { need_dx = true; if (k.seg[s].stats) need_norm = true; }
    }
    // Pointwise layers, whole backward in one call, every input gradient normalised: the weight gradient runs FIRST and also
    // leaves what the InstanceNorm-backward means are made of (conv_pw.hip); the input-gradient kernel then writes dx
    // itself -- no sums, no finish pass.
    bool pw_final = false;
    WgPlan wpl{};
    const bool pw_bd = d->ksize == 1 && use_blockdiag(k, d);       // batched passes: one dgrad over the masked sum of the passes
    const bool pw_f32 = d->y_dtype == PASSION_F32 && d->x_dtype == PASSION_BF16 && d->ksize == 1 && lean_f32_head(k, d);     // logit heads: f32 dY
    if (phase == 0 && !ga && parts == 3 && d->ksize == 1 && d->stride == 1 && pconv::g_conv_pwfinal && pconv::g_conv_lean && !g_passion_force_f32_mfma &&
        d->x_dtype == PASSION_BF16 && (d->y_dtype == PASSION_BF16 || pw_f32) && need_dx && need_norm && dweight && dbias && sums) {
        k.Mpad = pad16(k.Cout);
        wpl = plan_wgrad(k, d);
        bool all_norm = true, any_bmod = false;
        for (int s = 0; s < k.nseg; ++s) {
            if (k.seg[s].dx && !k.seg[s].stats) all_norm = false;
            if (k.seg[s].bmod > 0) any_bmod = true;
        }
        const int bphys = pw_bd ? k.seg[0].bmod : k.B;
        const int kblocks = pw_bd ? k.nseg * k.Cout / 8 : (k.Cout + 7) / 8;
        k.y_f32 = pw_f32 ? 1 : 0;
        pw_final = all_norm && (pw_bd || !any_bmod) && wpl.bf16 && wgrad_bf16_chunk(k, 1, 1) != 16 &&
                   wgrad_lean_covers(k, 1, 1, wpl.MT, wpl.gx, wpl.gy) && conv_pw_dgrad_covers(k, kblocks, bphys);
        k.y_f32 = 0;
        static const bool dbg = getenv("PASSION_PWFINAL_DBG") != nullptr;
        if (dbg) fprintf(stderr, "[pwfinal] B=%d V=%ld Cin=%d Cout=%d nseg=%d: final=%d all_norm=%d bd=%d any_bmod=%d bf16=%d chunk=%d leancov=%d gxmod=%d pwcov=%d\n",
                         k.B, (long)k.Di * k.Hi * k.Wi, k.Cin, k.Cout, k.nseg, (int)pw_final, (int)all_norm, (int)pw_bd, (int)any_bmod, (int)wpl.bf16,
                         wgrad_bf16_chunk(k, 1, 1), (int)wgrad_lean_covers(k, 1, 1, wpl.MT, wpl.gx, wpl.gy), wpl.gx % k.B, (int)conv_pw_dgrad_covers(k, kblocks, bphys));
    } else if (d->ksize == 1 && []() { static const bool dbg2 = getenv("PASSION_PWFINAL_DBG") != nullptr; return dbg2; }()) {
        fprintf(stderr, "[pwfinal] B=%d V=%ld Cin=%d Cout=%d: not planned (phase %d ga %d parts %d need_dx %d need_norm %d dW %d db %d dt %d/%d)\n", k.B,
                (long)k.Di * k.Hi * k.Wi, k.Cin, k.Cout, phase, ga != nullptr, parts, (int)need_dx, (int)need_norm, dweight != nullptr, dbias != nullptr, d->x_dtype, d->y_dtype);
    }
    // 3x3x3 stride-1 reflect layers of the finest level (8 output channels, <= 16 input channels: e1_c2 / e1_c3 grouped, d1_c1, d1_c2 of
    // both decoders), every input gradient normalised: the same order -- weight gradient first, with the indicator slab
    // (conv_wgrad_march.hip), per-sample folds + means (march_ident_kernel), then the FOLD-FREE marching input gradient that
    // writes dx itself (conv_march.hip FIN).  No padded scratch volume, no sums, no in_bwd_finish_pad launch.
    if (phase == 0 && parts == 3 && d->ksize == 3 && d->stride == 1 && d->pad_mode == PASSION_PAD_REFLECT &&
        (pconv::g_conv_marchfinal == 1 || (pconv::g_conv_marchfinal == 2 && (long)k.B * k.Di * k.Hi * k.Wi >= marchfinal_minvox())) &&
        pconv::g_conv_lean && !g_passion_force_f32_mfma && d->x_dtype == PASSION_BF16 && d->y_dtype == PASSION_BF16 && need_dx && need_norm &&
        dweight && sums && k.Cout == 8 && k.Cin <= 16 && !pconv::g_wgrad_fork) {
        bool ok = true;
        for (int s = 0; s < k.nseg; ++s) if (!k.seg[s].dx || !k.seg[s].stats || k.seg[s].bmod > 0) ok = false;
        k.Mpad = pad16(k.Cout);
        const WgPlan w = plan_wgrad(k, d);
        ConvK kd = k;                                    // would the marching input gradient take it?
        kd.Mpad = pad16(k.Cin);
        ok = ok && w.bf16 && w.MT == 1 && k.Hi % 8 == 0 && k.Wi % 16 == 0 && k.Di >= 8 && k.Hi >= 16 && k.Wi >= 32 && wgrad_lean_covers(k, 3, 1, w.MT, w.gx, w.gy) &&
             pconv::g_wgrad_march && pconv::conv_march_covers(kd, 3, 1, 2, nullptr) && kd.Mpad == 16 && w.gx % k.B == 0;
        if (ok) {
            const long nW = (long)k.Cout * k.Cin * nt;
            k.wslot = nW + k.Cout;
            float* wslab = sums + (long)2 * k.B * k.Cin * SUMS_SLOTS;
            float* sslab = wslab + (long)w.gx * k.wslot;
            float* dwb = sslab + (long)w.gx * k.wslot;
            float* pb = dwb + (long)k.B * (nW + k.Cout);
            float* mb = pb + (long)k.B * nW;
            k.spos = sslab;
            {
                ProfScope prof(3, conv_bytes(k, xs, ys) + (double)nt * k.Cin * k.Cout * 4, 2.0 * 2.0 * k.B * k.Do * k.Ho * k.Wo * (double)k.Cout * k.Cin * nt, st);
                rc = launch_wgrad_march(k, 3, 1, w.MT, w.gx, w.gy, wslab, wslab + nW, st);
            }
            k.spos = nullptr;
            if (rc > 0) rc = -1023;                      // planned for the marching kernels: must be covered
            if (rc) return rc;
            const int tiles_per_b = k.tilesD * k.tilesH * k.tilesW;
            const int wgs_per_b = (tiles_per_b + k.per - 1) / k.per;
            const int G = ga ? ga->G : 1;
            float* dw_out[PASSION_MAX_GROUPS] = {nullptr, nullptr, nullptr, nullptr};
            float* db_out[PASSION_MAX_GROUPS] = {nullptr, nullptr, nullptr, nullptr};
            for (int g = 0; g < G; ++g) {
                dw_out[g] = ga ? ga->dW[g] : dweight;
                db_out[g] = ga ? (ga->dW[g] ? ga->dW[g] + nW : nullptr) : dbias;
            }
            const bf16_t* wdg = reinterpret_cast<const bf16_t*>(packed + f32_pack_floats(k.Cin, k.Cout, nt)) + (long)nt * pad8(k.Cin) * pad16(k.Cout);
            {
                // what stands between the weight gradient and the input gradient: counted with the input gradient (kind 6)
                ProfScope prof(6, conv_bytes(k, xs, ys) + (double)nt * k.Cin * k.Cout * 4, 0.0, st);
                passion_prof_tag("finish of conv_march_kernel<1, 16, 2, 0, 1, 1>");
                rc = launch_march_ident(k, wslab, k.wslot, sslab, wgs_per_b, wdg, ga ? ga->stride * 2 : 0, dwb, pb, mb, dw_out, db_out, G, st);
            }
            if (rc) return rc;
            k.pwm = mb;
            k.sums = nullptr; k.sums_rep = 1; k.sums_fixed = 0; k.sums_float_only = 0;
            k.tilesD = cdiv(k.Di, TD); k.tilesH = cdiv(k.Hi, TH); k.tilesW = cdiv(k.Wi, TW);
            bf16_t* shell = use_padout(k, d) ? reinterpret_cast<bf16_t*>(sums + slab_end_floats(k, w.gx, nt)) : nullptr;      // the padded scratch: only its shell is used
            k.y = shell;
            {
                ProfScope prof(2, conv_bytes(k, xs, ys) + (double)nt * k.Cin * k.Cout * 4, 2.0 * k.B * k.Do * k.Ho * k.Wo * (double)k.Cout * k.Cin * nt, st);
                rc = launch_conv_march(k, 3, 1, 2, packed + f32_pack_floats(k.Cin, k.Cout, nt), st);
            }
            if (rc > 0 || (rc == 0 && !k.pw_done)) rc = -1023;
            if (rc == 0 && k.pw_done == 2 && k.nseg == 1 && k.seg[0].dx_add && add_fused) *add_fused |= 1u;      // (conv_pair.hip added it in its epilogue)
            if (rc == 0 && k.pw_done == 2) {             // interior written in final form, halo outputs in the shell: fold them
                char dg_tag[64];
                snprintf(dg_tag, sizeof(dg_tag), "%s", passion_prof_last_tag());
                ProfScope prof(6, 0.0, 0.0, st);
                passion_prof_tag("finish of %s", dg_tag);
                for (int s = 0; s < k.nseg; ++s) {
                    const SegD& sg = k.seg[s];
                    const long T = 2L * k.Hi * k.Wi + 2L * (k.Di - 2) * k.Wi + 2L * (k.Di - 2) * (k.Hi - 2);
                    const long items = (long)k.B * T * (sg.C / 4);
                    hipLaunchKernelGGL(shell_fold_kernel, dim3((unsigned)hmin((items + 255) / 256, 4096)), dim3(256), 0, st, (bf16_t*)sg.dx, shell, sg.stats,
                                       k.Di, k.Hi, k.Wi, sg.C, sg.bstride, k.Cin, sg.cbase, k.B);
                    LAUNCH_RET();
                }
            }
            return rc;
        }
    }
    Companion* comp = (phase == 0 && pconv::g_wgrad_fork == 1 && need_dx && (dweight || dbias) && !pw_final) ? companion(st) : nullptr;
    // option "wgrad_fork" = 2 (round 5): the same fork WITHOUT a second stream, for a step that is being recorded as a HIP graph: the
    // capturing stream's dependency set is read before the weight-gradient launches (hipStreamGetCaptureInfo_v2), put back behind
    // them (hipStreamUpdateCaptureDependencies: the input-gradient + finish launches then depend on what the weight gradient depended
    // on, not on it) and the weight gradient's tail is added to the set again at the end of the call -- two branches of the graph,
    // no stream, no event node.  Outside a capture the launches stay in stream order.
    std::vector<hipGraphNode_t> cap_d0, cap_dw;
    bool capfork = false;
    if (!comp && phase == 0 && (pconv::g_wgrad_fork == 2 || (pconv::g_wgrad_fork == 3 && (long)k.Di * k.Hi * k.Wi <= 40000)) && need_dx &&
        (dweight || dbias) && !pw_final)             // 3: the deep levels only (launches that do not fill the chip)
        capfork = capture_tail(st, cap_d0);
    // levels 3..5 (round 5): the 3x3x3 layers whose input gradient runs on conv_deep.hip get their weight gradient in the SAME launch
    // (deep_pair_kernel: two workgroup ranges of one grid).  The weight-gradient launcher parks its launch (wgrad_lean_defer), the
    // input-gradient launcher picks it up; whatever is still parked after the input gradient's launch goes out on its own.
    pending_wgrad().on = false;
    const bool pair_ok = !comp && !capfork && phase == 0 && parts == 3 && !pw_final && need_dx && (dweight || dbias) && d->ksize == 3 && d->stride == 1 &&
                         d->x_dtype == PASSION_BF16 && d->y_dtype == PASSION_BF16 && pconv::g_conv_lean && pconv::g_conv_deep && pconv::g_deep_pair &&
                         !g_passion_force_f32_mfma && (long)k.Di * k.Hi * k.Wi <= 40000 && k.Cout % 32 == 0 && k.Cin % 16 == 0;
    const bool forked = comp || capfork;              // the weight gradient's fold cannot ride in the finish launch (another stream / branch)
    const bool no_clear = forked || pair_ok;          // ... nor can the weight-gradient kernel clear the sums the input gradient adds into (pair: they run side by side;
                                                      // its fold still rides in the finish launch, which follows the pair launch)
    hipStream_t caller_st = st;
    if (comp) {                                      // fork: the weight-gradient work below goes to the companion stream
        HIP_RET(hipEventRecord(comp->fork, caller_st));
        HIP_RET(hipStreamWaitEvent(comp->side, comp->fork, 0));
        st = comp->side;
    }
    // cleared by the weight-gradient fold when that runs on the same stream as the dgrad
    float* zsums = (need_norm && sums && !no_clear) ? sums : nullptr;
    bool defer_fold = false;
    FoldJob fj{nullptr, {nullptr, nullptr, nullptr, nullptr}, 0, 0, 0, 0, 0, 0};
    const long zn = zsums ? (long)2 * k.B * k.Cin * SUMS_REP : 0;
    // ---- wgrad (+ bias grad): per-workgroup partial slabs, then a fixed-order fold (no atomics, deterministic) ----
    if (phase != 2 && (dweight || dbias) && (parts & 1)) {
        k.Mpad = pad16(k.Cout);
        WgPlan w = plan_wgrad(k, d);
        // grouped stride-2 layers (e2_c1 .. e5_c1 of the four modality encoders): no lean kernel; the general bf16 weight gradient takes the
        // whole stacked batch with its slots dealt to the weight sets in equal runs (conv_bf16.hip) -- one launch + one fold (riding in
        // the finish launch) instead of one of each per modality
        const bool grouped_general = ga && grouped_general_wgrad(k, d, w, ga->G);
        if (grouped_general) w.gx -= w.gx % ga->G;
        // the first layer (Cin = 1) on the matrix cores: slots in sample order, so a grouped launch is one launch + one fold too
        const bool c1m = w.bf16 && d->x_dtype == PASSION_BF16 && d->y_dtype == PASSION_BF16 && !g_passion_force_f32_mfma && pconv::g_conv_lean &&
                         dweight && pconv::wgrad_c1m_covers(k, d->ksize, d->stride, w.gx);
        // the first stride-2 layer (8 -> 16) on its marching kernel (conv_s2_wgrad.hip): slots in sample order as well
        const bool s2m = !c1m && w.bf16 && d->x_dtype == PASSION_BF16 && d->y_dtype == PASSION_BF16 && !g_passion_force_f32_mfma && pconv::g_conv_lean &&
                         dweight && (!ga || grouped_general) && pconv::wgrad_s2m_covers(k, d->ksize, d->stride, w.gx);
        const bool direct_slots = grouped_general || c1m || s2m;        // slab slots dealt to the weight sets in equal runs
        const long nW = (long)k.Cout * k.Cin * nt;
        k.wslot = nW + k.Cout;                       // one slot = [dW | dbias] of one workgroup
        float* wslab = sums + (long)2 * k.B * k.Cin * SUMS_SLOTS;
        float* bslab = wslab + nW;
        // the fold of the slab rides in the finish launch at the end of this call when one follows (FoldJob); the wgrad kernel then
        // clears the InstanceNorm-backward sums itself
        defer_fold = !forked && phase == 0 && (parts & 2) && need_dx && (ga || (dweight && dbias == dweight + nW)) && !pw_final;
        if (pw_final) k.spos = wslab + (long)w.gx * k.wslot;
        if (defer_fold && zsums) { k.zero_ptr = zsums; k.zero_n = zn; }
        {
            ProfScope prof(3, conv_bytes(k, xs, ys) + (double)nt * k.Cin * k.Cout * 4,
                           2.0 * k.B * k.Do * k.Ho * k.Wo * (double)k.Cout * k.Cin * nt, st);
            float* dWs = dweight ? wslab : nullptr;
            float* dbs = dbias ? bslab : nullptr;
            defer_next_wgrad() = pair_ok;
            if (c1m) {
                rc = launch_wgrad_c1m(k, d->ksize, d->stride, w.gx, dWs, dbs, st);
                if (rc > 0) rc = -1014;
            } else if (s2m) {
                rc = launch_wgrad_s2m(k, d->ksize, d->stride, w.gx, dWs, dbs, st);
                if (rc > 0) rc = -1014;
            } else if (ga && grouped_general) {      // grouped stride-2 layers: the general bf16 kernel, slots dealt to the sets in equal runs
                rc = launch_wgrad_bf16(k, d->ksize, d->stride, w.MT, w.gx, w.gy, dWs, dbs, st);
            } else if (ga) {                         // grouped: the lean kernel (slots in sample order), checked by the caller
                rc = launch_wgrad_lean(k, d->ksize, d->stride, w.MT, w.gx, w.gy, dWs, dbs, st);
                if (rc > 0) rc = -1014;
            } else if (w.bf16 && d->y_dtype == PASSION_F32) {
                k.y_f32 = 1;
                rc = launch_wgrad_lean(k, d->ksize, d->stride, w.MT, w.gx, w.gy, dWs, dbs, st);
                k.y_f32 = 0;
                if (rc > 0) rc = -1014;              // planned for the lean kernel: must be covered
            } else if (w.bf16) {
                rc = (pconv::g_conv_lean && d->x_dtype == PASSION_BF16 && d->y_dtype == PASSION_BF16)
                         ? launch_wgrad_c1(k, d->ksize, d->stride, w.gx, w.gy, dWs, dbs, st) : 1;
                if (rc > 0) rc = launch_wgrad_bf16(k, d->ksize, d->stride, w.MT, w.gx, w.gy, dWs, dbs, st);
            }
            else if (d->x_dtype == PASSION_F32 && d->y_dtype == PASSION_F32)
                rc = launch_wgrad_geo<float, float>(k, w, d->ksize, d->stride, dWs, dbs, st);
            else if (d->x_dtype == PASSION_BF16 && d->y_dtype == PASSION_BF16)
                rc = launch_wgrad_geo<bf16_t, bf16_t>(k, w, d->ksize, d->stride, dWs, dbs, st);
            else if (d->x_dtype == PASSION_BF16 && d->y_dtype == PASSION_F32)
                rc = launch_wgrad_geo<bf16_t, float>(k, w, d->ksize, d->stride, dWs, dbs, st);
            else
                rc = -1007;
            defer_next_wgrad() = false;
            if (rc) return rc;
        }
        k.zero_ptr = nullptr; k.zero_n = 0;
        if (pw_final) {
            // folded per sample below (launch_pw_ident), which also writes dweight / dbias
        } else
        if (defer_fold) {
            fj.part = wslab; fj.n = k.wslot; fj.stride = k.wslot; fj.R = w.gx; fj.G = 1;
            fj.out[0] = dweight;
            if (ga) {                                // one fold per weight set: R = its samples x workgroups per sample
                const int tiles_per_b = k.tilesD * k.tilesH * k.tilesW;
                fj.R = direct_slots ? w.gx / ga->G : k.bgroup * ((tiles_per_b + k.per - 1) / k.per);
                fj.G = ga->G;
                for (int g = 0; g < ga->G; ++g) fj.out[g] = ga->dW[g];
            }
            sums_cleared = zsums != nullptr;
        } else
        if (ga) {                                    // one fold per weight set: Rg = its samples x workgroups per sample
            const int tiles_per_b = k.tilesD * k.tilesH * k.tilesW;
            const int wgs_per_b = direct_slots ? 0 : (tiles_per_b + k.per - 1) / k.per;
            const int Rg = direct_slots ? w.gx / ga->G : k.bgroup * wgs_per_b;
            RedOuts ro;
            for (int g = 0; g < PASSION_MAX_GROUPS; ++g) ro.p[g] = g < ga->G ? ga->dW[g] : nullptr;
            hipLaunchKernelGGL(reduce_parts_grouped_kernel, dim3((unsigned)fold_nblocks(k.wslot, Rg), ga->G), dim3(256), 0, st, wslab,
                               Rg, k.wslot, k.wslot, ro, zsums, zn);
            LAUNCH_RET();
            sums_cleared = zsums != nullptr;
        } else
        if (dweight && dbias == dweight + nW) {      // caller keeps [dW | dbias] in one buffer: one fold for both
            hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)fold_nblocks(k.wslot, w.gx)), dim3(256), 0, st, wslab, w.gx, k.wslot,
                               k.wslot, dweight, zsums, zn);
            LAUNCH_RET();
            sums_cleared = zsums != nullptr;
        } else {
            if (dweight) {
                hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)fold_nblocks(nW, w.gx)), dim3(256), 0, st, wslab, w.gx, nW, k.wslot, dweight,
                                   zsums, zn);
                LAUNCH_RET();
                sums_cleared = zsums != nullptr;
            }
            if (dbias) {
                hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)fold_nblocks(k.Cout, w.gx)), dim3(256), 0, st, bslab, w.gx, (long)k.Cout, k.wslot, dbias);
                LAUNCH_RET();
            }
        }
    }
    if (comp) {
        HIP_RET(hipEventRecord(comp->done, comp->side));
        st = caller_st;
    }
    if (capfork) {                                   // the launches below hang off what the weight gradient hung off
        if (!capture_tail(st, cap_dw)) return -1024;
        HIP_RET(hipStreamUpdateCaptureDependencies(st, cap_d0.data(), cap_d0.size(), hipStreamSetCaptureDependencies));
    }
    if (pw_final) {
        // per-sample folds + contraction -> (m1, m2) / V (over the passes of a batched launch), then the input gradient with its
        // final epilogue; no finish pass follows
        const long nW = (long)k.Cout * k.Cin;
        const float* wslab = sums + (long)2 * k.B * k.Cin * SUMS_SLOTS;
        float* sslab = const_cast<float*>(wslab) + (long)wpl.gx * k.wslot;
        float* dwb = sslab + (long)wpl.gx * nW;
        float* pb = dwb + (long)k.B * (nW + k.Cout);
        float* mb = pb + (long)k.B * nW;
        const bf16_t* wdg = reinterpret_cast<const bf16_t*>(packed + f32_pack_floats(k.Cin, k.Cout, 1)) + (long)pad8(k.Cin) * pad16(k.Cout);
        k.spos = nullptr;
        // slots of sample b: the wgs_per_b consecutive ones from b * wgs_per_b on (wgrad_lean_kernel; spare slots past the last sample are zero)
        const int wg_tiles = cdiv(k.Do, TD) * cdiv(k.Ho, TH) * cdiv(k.Wo, TW);
        const int wgs_per_b = (wg_tiles + k.per - 1) / k.per;
        rc = launch_pw_ident(k, wslab, k.wslot, sslab, wgs_per_b, wdg, dwb, pb, mb, dweight, dbias, pw_bd ? k.seg[0].bmod : k.B, st);
        if (rc) return rc;
        k.pwm = mb;
        if (!pw_bd) {
            k.sums = nullptr; k.sums_rep = 1; k.sums_fixed = 0; k.sums_float_only = 0;
            k.tilesD = cdiv(k.Di, TD); k.tilesH = cdiv(k.Hi, TH); k.tilesW = cdiv(k.Wi, TW);
            ProfScope prof(2, conv_bytes(k, xs, ys) + (double)k.Cin * k.Cout * 4, 2.0 * k.B * k.Do * k.Ho * k.Wo * (double)k.Cout * k.Cin, st);
            k.y_f32 = pw_f32 ? 1 : 0;
            rc = launch_conv_lean(k, 1, 1, 1, packed + f32_pack_floats(k.Cin, k.Cout, 1), st);
            k.y_f32 = 0;
            if (rc > 0 || (rc == 0 && !k.pw_done)) rc = -1020;      // planned for the pointwise kernels: must be covered
            return rc;
        }
        // batched passes: the common path below sums the passes' dY and runs the block-diagonal dgrad (with k.pwm: its final form)
    }
    // ---- dgrad ----
    if (!need_dx || !(parts & 2)) return 0;
    if (k.Cin % 4) return -1008;
    for (int s = 0; s < k.nseg; ++s) if (k.seg[s].C % 4) return -1008;
    if (need_norm) {
        if (!sums) return -1009;
        if (phase != 2 && !sums_cleared) HIP_RET(hipMemsetAsync(sums, 0, sizeof(float) * 2 * (size_t)k.B * k.Cin * SUMS_REP, st));
    }
    k.sums = sums;
    k.sums_rep = SUMS_REP;                           // the lean launcher lowers it
    k.sums_fixed = 0;
    k.sums_float_only = phase != 0;
    k.Mpad = pad16(k.Cin);
    k.wpack = packed + (long)nt * k.Cin * pad16(k.Cout);
    k.tilesD = cdiv(k.Di, TD); k.tilesH = cdiv(k.Hi, TH); k.tilesW = cdiv(k.Wi, TW);
    const int mtiles = k.Mpad / 16;
    bool padout = use_padout(k, d);
    void* dxp = nullptr;
    // batched-pass segments: the dgrad writes the per-pass gradients into scratch, the finish kernel sums them into dx
    void* real_dx[PASSION_MAX_SEG] = {nullptr, nullptr, nullptr, nullptr};
    const bool blockdiag = use_blockdiag(k, d);
    if (phase == 2) {
        if (padout) dxp = sums + slab_end_floats(k, plan_wgrad(k, d).gx, nt);
    } else
    if (blockdiag) {         // passes summed first (per segment, masked), one dgrad over the physical samples
        const WgPlan w0 = plan_wgrad(k, d);
        bf16_t* dycat = reinterpret_cast<bf16_t*>(sums + slab_end_floats(k, w0.gx, nt) + (padout ? padout_floats(k) : 0));
        const int bm = k.seg[0].bmod, P = k.B / bm;
        const long V = (long)k.Di * k.Hi * k.Wi;
        PassSumArgs pa;
        for (int s = 0; s < PASSION_MAX_SEG; ++s) {
            pa.mask[s] = s < k.nseg ? k.seg[s].mask : nullptr;
            pa.col[s] = s < k.nseg ? k.seg[s].mask_col : 0;
            pa.ld[s] = s < k.nseg ? k.seg[s].mask_ld : 0;
        }
        const long items = (long)bm * V * (k.Cout / 8);
        hipLaunchKernelGGL(pass_sum_kernel, dim3((unsigned)hmin((items + 255) / 256, 8192)), dim3(256), 0, st,
                           reinterpret_cast<const bf16_t*>(dy), dycat, pa, P, bm, V, k.Cout, k.nseg);
        LAUNCH_RET();
        k.bd_rows = k.seg[0].C;
        k.bd_k8 = k.Cout / 8;
        k.Cout = k.nseg * k.Cout;                    // K axis of the dgrad GEMM: one Cout-block per segment
        k.B = bm;
        k.dy = dycat;
        for (int s = 0; s < k.nseg; ++s) { k.seg[s].mask = nullptr; k.seg[s].bmod = 0; }
        k.tilesD = cdiv(k.Di, TD); k.tilesH = cdiv(k.Hi, TH); k.tilesW = cdiv(k.Wi, TW);
    } else {
        const WgPlan w0 = plan_wgrad(k, d);
        char* gp = reinterpret_cast<char*>(sums + slab_end_floats(k, w0.gx, nt) + (padout ? padout_floats(k) : 0));
        const long es = d->x_dtype == PASSION_BF16 ? 2 : 4;
        for (int s = 0; s < k.nseg; ++s) {
            if (k.seg[s].dx && k.seg[s].bmod > 0) {
                if ((long)(k.B / k.seg[s].bmod) * k.seg[s].C > FIN_TAB || k.B / k.seg[s].bmod > 16) return -1015;
                real_dx[s] = k.seg[s].dx;
                k.seg[s].dx = gp;
                gp += ((long)k.B * k.Di * k.Hi * k.Wi * k.seg[s].C * es + 15) / 16 * 16;
            }
        }
    }
    // RAW inputs of the finest 3x3x3 layers (nothing normalised in the loader: RFNet's post-norm general_conv3d, reference
    // blocks.py:354-370 -- the activation in front of the conv is its own op there): no InstanceNorm backward follows the input gradient, so
    // the finish pass over the padded scratch only folded the halo -- a copy of the whole tensor (in_bwd_finish_pad_kernel: 3.5 ms of a
    // 47 ms RFNet step, alone on the chip).  The final-form kernels of the finish-free backward (conv_march.hip FIN, conv_pair.hip SF) write
    // dx themselves; for a raw input their epilogue has no means to apply (k.pwm only has to be readable).
    bool raw_done = false;
    if (phase == 0 && padout && !blockdiag && need_dx && !need_norm && d->ksize == 3 && d->stride == 1 && d->pad_mode == PASSION_PAD_REFLECT &&
        (pconv::g_conv_marchfinal == 1 || (pconv::g_conv_marchfinal == 2 && (long)k.B * k.Di * k.Hi * k.Wi >= marchfinal_minvox())) &&
        d->x_dtype == PASSION_BF16 && d->y_dtype == PASSION_BF16 && k.Cout == 8 && k.Cin <= 16 && k.Hi % 8 == 0 && k.Wi % 16 == 0 && k.Di >= 8 &&
        k.Hi >= 16 && k.Wi >= 32) {
        bool ok = true;
        for (int s = 0; s < k.nseg; ++s) if (!k.seg[s].dx || k.seg[s].stats || k.seg[s].bmod > 0) ok = false;
        if (ok) {
            ConvK kf = k;
            kf.pwm = sums;                               // read, never applied (no statistics on any segment)
            kf.sums = nullptr; kf.sums_rep = 1; kf.sums_fixed = 0; kf.sums_float_only = 0;
            kf.y = sums + slab_end_floats(k, plan_wgrad(k, d).gx, nt);      // the padded scratch: only its shell is used
            kf.pw_done = 0;
            {
                ProfScope prof(2, conv_bytes(k, xs, ys) + (double)nt * k.Cin * k.Cout * 4, 2.0 * k.B * k.Do * k.Ho * k.Wo * (double)k.Cout * k.Cin * nt, st);
                rc = launch_conv_march(kf, 3, 1, 2, packed + f32_pack_floats(k.Cin, k.Cout, nt), st);
            }
            if (rc < 0) return rc;
            if (rc == 0 && kf.pw_done) {
                raw_done = true;
                if (kf.pw_done == 2 && kf.nseg == 1 && kf.seg[0].dx_add && add_fused) *add_fused |= 1u;
                if (kf.pw_done == 2) {                   // interior written in final form, halo outputs in the shell: fold them
                    char dg_tag[64];
                    snprintf(dg_tag, sizeof(dg_tag), "%s", passion_prof_last_tag());
                    ProfScope prof(6, 0.0, 0.0, st);
                    passion_prof_tag("finish of %s", dg_tag);
                    for (int s = 0; s < k.nseg; ++s) {
                        const SegD& sg = k.seg[s];
                        const long T = 2L * k.Hi * k.Wi + 2L * (k.Di - 2) * k.Wi + 2L * (k.Di - 2) * (k.Hi - 2);
                        const long items = (long)k.B * T * (sg.C / 4);
                        hipLaunchKernelGGL(shell_fold_kernel, dim3((unsigned)hmin((items + 255) / 256, 4096)), dim3(256), 0, st, (bf16_t*)sg.dx,
                                           reinterpret_cast<const bf16_t*>(kf.y), (const float*)nullptr, k.Di, k.Hi, k.Wi, sg.C, sg.bstride, k.Cin, sg.cbase, k.B);
                        LAUNCH_RET();
                    }
                }
            }
            rc = 0;
        }
    }
    if (phase != 2 && !raw_done) {
        ProfScope prof(2, conv_bytes(k, xs, ys) + (double)nt * k.Cin * k.Cout * 4,
                       2.0 * k.B * k.Do * k.Ho * k.Wo * (double)k.Cout * k.Cin * nt, st);
        if (padout) {
            const WgPlan w = plan_wgrad(k, d);
            dxp = sums + slab_end_floats(k, w.gx, nt);
            k.tilesD = cdiv(k.Di + 2, TD); k.tilesH = cdiv(k.Hi + 2, TH); k.tilesW = cdiv(k.Wi + 2, TW);
            k.y = dxp;
            rc = d->pad_mode == PASSION_PAD_REFLECT
                     ? launch_conv_lean(k, d->ksize, d->stride, 2, packed + f32_pack_floats(k.Cin, k.Cout, nt), st)
                     : launch_conv_march(k, d->ksize, d->stride, 2, packed + f32_pack_floats(k.Cin, k.Cout, nt), st);    // zero padding: marching kernel or not at all
            if (rc < 0) return rc;
            if (rc > 0) {
                padout = false;
                k.tilesD = cdiv(k.Di, TD); k.tilesH = cdiv(k.Hi, TH); k.tilesW = cdiv(k.Wi, TW);
            }
        }
        // TI = dy dtype, TO = x/dx dtype
        // the one-brick-per-workgroup kernels below (f32 parity mode, ragged / stride-2 bf16 shapes): fixed-point sums over a few
        // replicas, so that these gradients repeat bit for bit too (the lean launchers set their own)
        if (!padout && phase == 0 && !k.sums_float_only) { k.sums_rep = SUMS_REP_LEAN; k.sums_fixed = 1; }
        if (padout) rc = 0;
        else if (d->x_dtype == PASSION_F32 && d->y_dtype == PASSION_F32)
            rc = launch_conv_geo<float, float, 1>(k, d->ksize, d->stride, mtiles, st);
        else if (blockdiag) {
            rc = launch_conv_lean(k, d->ksize, d->stride, 1, packed + f32_pack_floats(k.Cin, k.bd_k8 * 8, nt), st);
            if (rc > 0) rc = -1018;              // planned for the lean kernel: must be covered
            if (pw_final) return (rc == 0 && !k.pw_done) ? -1020 : rc;      // the pointwise kernel wrote dx itself (k.pwm): nothing to finish
        }
        else if (d->x_dtype == PASSION_BF16 && d->y_dtype == PASSION_BF16) {
            if (bf16_eligible(k, true) && !g_passion_force_f32_mfma)
                rc = launch_conv_bf16(k, d->ksize, d->stride, true, packed + f32_pack_floats(k.Cin, k.Cout, nt), st);
            else
                rc = launch_conv_geo<bf16_t, bf16_t, 1>(k, d->ksize, d->stride, mtiles, st);
        }
        else if (d->x_dtype == PASSION_BF16 && d->y_dtype == PASSION_F32) {
            rc = 1;
            if (lean_f32_head(k, d) && d->ksize == 1) {
                k.y_f32 = 1;
                rc = launch_conv_lean(k, d->ksize, d->stride, 1, packed + f32_pack_floats(k.Cin, k.Cout, nt), st);
                k.y_f32 = 0;
            }
            if (rc > 0) rc = launch_conv_geo<float, bf16_t, 1>(k, d->ksize, d->stride, mtiles, st);
        }
        else
            rc = -1007;
        if (rc) return rc;
    }
    if ((rc = wgrad_lean_flush(st))) return rc;          // a parked weight gradient the input-gradient launcher did not take
    // kind 6: what completes the input gradient after the dgrad kernel -- the fold of the InstanceNorm-backward sums and the
    // finish kernels (halo fold + InstanceNorm backward): reported by bench.py beside the dgrad kernel they belong to
    char dg_tag[64];
    snprintf(dg_tag, sizeof(dg_tag), "%s", passion_prof_last_tag());
    ProfScope prof_fin(6, conv_bytes(k, xs, ys) + (double)nt * k.Cin * k.Cout * 4, 0.0, st);
    passion_prof_tag("finish of %s", dg_tag);
    int rep = phase == 2 ? 1 : k.sums_rep;
    if (k.sums_fixed && phase == 0) rep = -rep;          // fixed-point replicas: folded (exactly) by the finish kernels
    if (phase != 2 && need_norm && rep > 0 && (rep > SUMS_REP_LEAN || phase == 1)) {          // many replicas: fold them first (in place, replica 0)
        const int nw = 2 * k.B * k.Cin;
        hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)fold_nblocks(nw, k.sums_rep)), dim3(256), 0, st, sums, k.sums_rep, (long)nw, (long)nw, sums);
        LAUNCH_RET();
        rep = 1;
    }
    if (phase == 1) return 0;
    const long hw = (long)k.Hi * k.Wi;
    const long win_lo = (long)dwin_lo * hw, win_hi = (long)dwin_hi * hw;
    // ---- InstanceNorm backward finish (in place on dx) ----
    const long V = (long)k.Di * k.Hi * k.Wi;
    const long rs = (long)2 * k.B * k.Cin;
    if (padout && !raw_done) {
        // one launch per group of segments with the same channel count (FinSegs): the scratch is read from HBM once
        bool done[PASSION_MAX_SEG] = {false, false, false, false};
        static const int fin_nt = []() { const char* e = getenv("PASSION_FINISH_NT"); return e ? atoi(e) : 0; }();
        // blocks per sample: every block first builds its coefficient table (dependent loads of the statistics and the sums: ~2 us of
        // latency) before it streams its rows -- 4096 blocks of four rows each paid that too often (4 x 128^3, 8 channels: 117 us;
        // 2048: 107; 1024: 105; 512: 124)
        static const long fin_blocks = []() { const char* e = getenv("PASSION_FINISH_BLOCKS"); return e && atol(e) > 0 ? atol(e) : 1536L; }();
        static const bool fin_merge = []() { const char* e = getenv("PASSION_FINISH_MERGE"); return !(e && e[0] == '0'); }();
        for (int s = 0; s < k.nseg; ++s) {
            const SegD& sg = k.seg[s];
            if (done[s] || !sg.dx) continue;
            if (sg.C > FIN_MAXC) return -1013;            // coefficient table of the finish kernels
            const bool v8 = d->x_dtype == PASSION_BF16 && sg.C % 8 == 0 && k.Cin % 8 == 0 && sg.cbase % 8 == 0;
            FinSegs fs;
            fs.n = 0;
            int coff = 0;
            for (int s2 = s; s2 < k.nseg; ++s2) {
                const SegD& q = k.seg[s2];
                const bool v8q = d->x_dtype == PASSION_BF16 && q.C % 8 == 0 && k.Cin % 8 == 0 && q.cbase % 8 == 0;
                if (done[s2] || !q.dx || q.C != sg.C || v8q != v8 || coff + q.C > FIN_MAXC || (fs.n > 0 && !fin_merge)) continue;
                fs.s[fs.n++] = FinSeg{q.dx, q.x, q.stats, q.mask, q.bstride, q.C, q.cbase, q.mask_col, q.mask_ld, q.bmod, coff, q.dx_add, q.dx_add2};
                if (q.dx_add && add_fused) *add_fused |= (1u << s2) | (q.dx_add2 ? 1u << (8 + s2) : 0u);
                coff += q.C;
                done[s2] = true;
            }
            FoldJob fjl = fj;                             // the pending fold goes with the first finish launch
            fjl.main_x = 0; fjl.fold_x = 0;
            const long fold_blocks = defer_fold ? fold_nblocks(fj.n, fj.R) * fj.G : 0;
            const int rowv = k.Wi * sg.C / (v8 ? 8 : 4);
            const int rpb = 256 / (rowv < 256 ? rowv : 256);
            const int foldarg = (d->pad_mode == PASSION_PAD_REFLECT ? 1 : 0) | (fin_nt << 1);
            dim3 pg((unsigned)hmin(((long)k.Di * k.Hi + rpb - 1) / rpb, fin_blocks), k.B);
            if (defer_fold) { fjl.main_x = (int)pg.x; fjl.fold_x = (int)((fold_blocks + pg.y - 1) / pg.y); pg.x += (unsigned)fjl.fold_x; }
            if (v8)
                hipLaunchKernelGGL((in_bwd_finish_pad_kernel<bf16_t, 8>), pg, dim3(256), 0, st, fs, (const bf16_t*)dxp, sums, k.Di, k.Hi, k.Wi, k.Cin, rep, rs,
                                   vscale, dwin_lo, dwin_hi, fjl, foldarg);
            else
                hipLaunchKernelGGL((in_bwd_finish_pad_kernel<bf16_t, 4>), pg, dim3(256), 0, st, fs, (const bf16_t*)dxp, sums, k.Di, k.Hi, k.Wi, k.Cin, rep, rs,
                                   vscale, dwin_lo, dwin_hi, fjl, foldarg);
            LAUNCH_RET();
            defer_fold = false;
        }
    }
    for (int s = 0; s < k.nseg && !padout && !raw_done; ++s) {
        const SegD& sg = k.seg[s];
        const bool sum_passes = real_dx[s] != nullptr;
        if (!sg.dx || (!sg.stats && !sum_passes)) continue;
        if (sg.C > FIN_MAXC) return -1013;            // coefficient table of the finish kernels
        const bool v8 = d->x_dtype == PASSION_BF16 && sg.C % 8 == 0 && k.Cin % 8 == 0 && sg.cbase % 8 == 0;
        const long nvec = V * sg.C / (v8 ? 8 : 4);
        const int passes = sum_passes ? k.B / sg.bmod : 1;
        dim3 grid((unsigned)hmin((nvec + 255) / 256, 2048), sum_passes ? sg.bmod : k.B);
        void* out = sum_passes ? real_dx[s] : sg.dx;
        FoldJob fjl = fj;                             // the pending fold goes with the first finish launch
        fjl.main_x = 0; fjl.fold_x = 0;
        const long fold_blocks = defer_fold ? fold_nblocks(fj.n, fj.R) * fj.G : 0;
        if (defer_fold) { fjl.main_x = (int)grid.x; fjl.fold_x = (int)((fold_blocks + grid.y - 1) / grid.y); grid.x += (unsigned)fjl.fold_x; }
        if (d->x_dtype == PASSION_F32)
            hipLaunchKernelGGL((in_bwd_finish_kernel<float, 4>), grid, dim3(256), 0, st, (const float*)sg.dx, (float*)out, (const float*)sg.x,
                               sg.stats, sums, sg.mask, sg.mask_col, sg.mask_ld, V, sg.C, sg.bstride, k.Cin, sg.cbase, sg.bmod, passes, rep, rs, vscale, win_lo, win_hi, fjl);
        else if (v8)
            hipLaunchKernelGGL((in_bwd_finish_kernel<bf16_t, 8>), grid, dim3(256), 0, st, (const bf16_t*)sg.dx, (bf16_t*)out, (const bf16_t*)sg.x,
                               sg.stats, sums, sg.mask, sg.mask_col, sg.mask_ld, V, sg.C, sg.bstride, k.Cin, sg.cbase, sg.bmod, passes, rep, rs, vscale, win_lo, win_hi, fjl);
        else
            hipLaunchKernelGGL((in_bwd_finish_kernel<bf16_t, 4>), grid, dim3(256), 0, st, (const bf16_t*)sg.dx, (bf16_t*)out, (const bf16_t*)sg.x,
                               sg.stats, sums, sg.mask, sg.mask_col, sg.mask_ld, V, sg.C, sg.bstride, k.Cin, sg.cbase, sg.bmod, passes, rep, rs, vscale, win_lo, win_hi, fjl);
        LAUNCH_RET();
        defer_fold = false;
    }
    if (defer_fold) {                                 // no finish kernel ran (raw input, no padded domain): the fold on its own
        RedOuts ro;
        for (int g = 0; g < PASSION_MAX_GROUPS; ++g) ro.p[g] = fj.out[g];
        hipLaunchKernelGGL(reduce_parts_grouped_kernel, dim3((unsigned)fold_nblocks(fj.n, fj.R), fj.G), dim3(256), 0, st, fj.part, fj.R, fj.n, fj.stride, ro,
                           (float*)nullptr, 0L);
        LAUNCH_RET();
    }
    if (comp) HIP_RET(hipStreamWaitEvent(caller_st, comp->done, 0));     // join
    if (capfork && !cap_dw.empty()) HIP_RET(hipStreamUpdateCaptureDependencies(st, cap_dw.data(), cap_dw.size(), hipStreamAddCaptureDependencies));
    return 0;
}


// ---- grouped launches: G weight sets over one batch (samples b / (B / G) -> set) ------------------------------------------
namespace {
// the descriptor of group g's samples alone (the per-group fallback of the kernels that take one weight set per launch)
passion_conv_desc_t sub_desc(const passion_conv_desc_t* d, int g, int bg) {
    passion_conv_desc_t s = *d;
    s.B = bg;
    const long es = d->x_dtype == PASSION_BF16 ? 2 : 4;
    for (int i = 0; i < d->nseg; ++i) {
        passion_seg_t& sg = s.seg[i];
        const long off = (long)g * bg;
        sg.x = static_cast<const char*>(sg.x) + off * sg.bstride * es;
        if (sg.stats) sg.stats += off * sg.C * 2;
        if (sg.mask) sg.mask += off * sg.mask_ld;
        if (sg.dx) sg.dx = static_cast<char*>(sg.dx) + off * sg.bstride * es;
        if (sg.dx_add) sg.dx_add = static_cast<const char*>(sg.dx_add) + off * (long)d->D * d->H * d->W * sg.C * es;      // (contiguous)
        if (sg.dx_add2) sg.dx_add2 = static_cast<const char*>(sg.dx_add2) + off * (long)d->D * d->H * d->W * sg.C * es;
    }
    return s;
}
int check_groups(const passion_conv_desc_t* d, int G) {
    if (!d || G < 1 || G > PASSION_MAX_GROUPS || d->B % G) return -1016;
    for (int i = 0; i < d->nseg; ++i) if (d->seg[i].bmod > 0) return -1016;      // batched passes are not grouped
    return 0;
}
bool uniform_stride(const float* const* packed, int G, long& stride) {
    stride = G > 1 ? packed[1] - packed[0] : 0;
    for (int g = 1; g < G; ++g) if (packed[g] - packed[g - 1] != stride) return false;
    return G == 1 || (stride > 0 && stride % 4 == 0);
}
}  // namespace

extern "C" int passion_conv3d_fwd_grouped(const passion_conv_desc_t* d, int G, const float* const* packed, const float* const* bias,
                                          const void* residual, void* y, float* stats, float eps, void* stream) {
    int rc = check_groups(d, G);
    if (rc) return rc;
    if (G == 1) return conv3d_fwd_impl(d, packed[0], bias ? bias[0] : nullptr, residual, y, stats, eps, stream);
    ConvK k;
    rc = fill_params(d, k);
    if (rc) return rc;
    long stride = 0;
    const bool native = d->x_dtype == PASSION_BF16 && d->y_dtype == PASSION_BF16 && bf16_eligible(k, false) && !g_passion_force_f32_mfma &&
                        uniform_stride(packed, G, stride);
    if (native) {
        GroupArgs ga{G, stride, bias, nullptr};
        return conv3d_fwd_impl(d, packed[0], bias ? bias[0] : nullptr, residual, y, stats, eps, stream, &ga);
    }
    const int bg = d->B / G;
    const long ys = d->y_dtype == PASSION_BF16 ? 2 : 4;
    const long vout = (long)k.Do * k.Ho * k.Wo * k.Cout;
    for (int g = 0; g < G; ++g) {
        const passion_conv_desc_t s = sub_desc(d, g, bg);
        const long off = (long)g * bg * vout;
        rc = conv3d_fwd_impl(&s, packed[g], bias ? bias[g] : nullptr, residual ? static_cast<const char*>(residual) + off * ys : nullptr,
                             static_cast<char*>(y) + off * ys, stats ? stats + (long)g * bg * k.Cout * 2 : nullptr, eps, stream);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int passion_conv3d_bwd_grouped(const passion_conv_desc_t* d, int G, const float* const* packed, const void* dy,
                                          float* const* dW, float* sums, void* stream) {
    int rc = check_groups(d, G);
    if (rc) return rc;
    ConvK k;
    rc = fill_params(d, k);
    if (rc) return rc;
    const int nt = d->ksize * d->ksize * d->ksize;
    const long nW = (long)k.Cout * k.Cin * nt;
    if (G == 1) return conv3d_bwd_impl(d, packed[0], dy, dW ? dW[0] : nullptr, dW ? dW[0] + nW : nullptr, sums, stream, 0, 1.0f, 0, d->D);
    long stride = 0;
    const bool bf16 = d->x_dtype == PASSION_BF16 && d->y_dtype == PASSION_BF16 && !g_passion_force_f32_mfma && uniform_stride(packed, G, stride);
    const bool native_dgrad = bf16 && bf16_eligible(k, true);
    k.tilesD = cdiv(k.Do, TD); k.tilesH = cdiv(k.Ho, TH); k.tilesW = cdiv(k.Wo, TW);
    const WgPlan w = plan_wgrad(k, d);
    const bool c1m = bf16 && dW && w.bf16 && pconv::g_conv_lean && pconv::wgrad_c1m_covers(k, d->ksize, d->stride, w.gx);      // first layer, matrix cores
    const bool native_wgrad = bf16 && dW && w.bf16 && pconv::g_conv_lean && wgrad_bf16_chunk(k, d->ksize, d->stride) != 16 &&
                              ((!(k.Cin == 1) && (wgrad_lean_covers(k, d->ksize, d->stride, w.MT, w.gx, w.gy) || grouped_general_wgrad(k, d, w, G))) || c1m);
    GroupArgs ga{G, stride, nullptr, dW};
    bool no_dx = true;                                // a raw input (the first layer's): there is no input gradient to run
    for (int i = 0; i < d->nseg; ++i) if (d->seg[i].dx) no_dx = false;
    if (no_dx && native_wgrad) return conv3d_bwd_impl(d, packed[0], dy, nullptr, nullptr, sums, stream, 0, 1.0f, 0, d->D, &ga, 1);
    const int bg = d->B / G;
    const long ys = d->y_dtype == PASSION_BF16 ? 2 : 4;
    const long vout = (long)k.Do * k.Ho * k.Wo * k.Cout;
    if (native_dgrad) {
        if (native_wgrad || !dW) return conv3d_bwd_impl(d, packed[0], dy, nullptr, nullptr, sums, stream, 0, 1.0f, 0, d->D, &ga, dW ? 3 : 2);
        for (int g = 0; g < G; ++g) {                 // weight gradients set by set (general kernels), then ONE input-gradient pass
            passion_conv_desc_t s = sub_desc(d, g, bg);
            for (int i = 0; i < s.nseg; ++i) s.seg[i].dx = nullptr;
            rc = conv3d_bwd_impl(&s, packed[g], static_cast<const char*>(dy) + (long)g * bg * vout * ys, dW[g], dW[g] + nW, sums, stream, 0, 1.0f,
                                 0, d->D, nullptr, 1);
            if (rc) return rc;
        }
        GroupArgs gd{G, stride, nullptr, nullptr};
        return conv3d_bwd_impl(d, packed[0], dy, nullptr, nullptr, sums, stream, 0, 1.0f, 0, d->D, &gd, 2);
    }
    for (int g = 0; g < G; ++g) {
        const passion_conv_desc_t s = sub_desc(d, g, bg);
        rc = conv3d_bwd_impl(&s, packed[g], static_cast<const char*>(dy) + (long)g * bg * vout * ys, dW ? dW[g] : nullptr,
                             dW ? dW[g] + nW : nullptr, sums, stream, 0, 1.0f, 0, d->D);
        if (rc) return rc;
    }
    return 0;
}

// the two-phase backward (below) of a grouped launch: the four modality encoders of a depth-sharded volume as one launch per layer
// and phase.  Only where both gradients have a grouped kernel (bf16, 8-channel-aligned): -1021 otherwise, the caller goes set by set.
extern "C" int passion_conv3d_bwd_grouped_phase(const passion_conv_desc_t* d, int G, const float* const* packed, const void* dy,
                                                float* const* dW, float* sums, int phase, float vscale, int dwin_lo, int dwin_hi,
                                                void* stream) {
    if (phase != 1 && phase != 2) return -1003;
    int rc = check_groups(d, G);
    if (rc) return rc;
    ConvK k;
    rc = fill_params(d, k);
    if (rc) return rc;
    const int nt = d->ksize * d->ksize * d->ksize;
    const long nW = (long)k.Cout * k.Cin * nt;
    if (G == 1) return conv3d_bwd_impl(d, packed[0], dy, dW ? dW[0] : nullptr, dW ? dW[0] + nW : nullptr, sums, stream, phase, vscale, dwin_lo, dwin_hi);
    long stride = 0;
    const bool bf16 = d->x_dtype == PASSION_BF16 && d->y_dtype == PASSION_BF16 && !g_passion_force_f32_mfma && uniform_stride(packed, G, stride);
    k.tilesD = cdiv(k.Do, TD); k.tilesH = cdiv(k.Ho, TH); k.tilesW = cdiv(k.Wo, TW);
    const WgPlan w = plan_wgrad(k, d);
    const bool native_wgrad = bf16 && dW && w.bf16 && pconv::g_conv_lean && wgrad_bf16_chunk(k, d->ksize, d->stride) != 16 &&
                              !(k.Cin == 1) && wgrad_lean_covers(k, d->ksize, d->stride, w.MT, w.gx, w.gy);
    if (!(bf16 && bf16_eligible(k, true))) return -1021;
    if (native_wgrad || !dW || phase == 2) {
        GroupArgs ga{G, stride, nullptr, phase == 2 ? nullptr : dW};
        return conv3d_bwd_impl(d, packed[0], dy, nullptr, nullptr, sums, stream, phase, vscale, dwin_lo, dwin_hi, &ga, (dW && phase == 1 && native_wgrad) ? 3 : 2);
    }
    // stride-2 layers: weight gradients set by set (general kernels), then ONE input-gradient pass -- as passion_conv3d_bwd_grouped
    const int bg = d->B / G;
    const long vout = (long)k.Do * k.Ho * k.Wo * k.Cout;
    for (int g = 0; g < G; ++g) {
        passion_conv_desc_t s = sub_desc(d, g, bg);
        for (int i = 0; i < s.nseg; ++i) s.seg[i].dx = nullptr;
        rc = conv3d_bwd_impl(&s, packed[g], static_cast<const char*>(dy) + (long)g * bg * vout * 2, dW[g], dW[g] + nW, sums, stream, 0, 1.0f, 0, d->D,
                             nullptr, 1);
        if (rc) return rc;
    }
    GroupArgs gd{G, stride, nullptr, nullptr};
    return conv3d_bwd_impl(d, packed[0], dy, nullptr, nullptr, sums, stream, 1, vscale, dwin_lo, dwin_hi, &gd, 2);
}

extern "C" int passion_conv3d_bwd(const passion_conv_desc_t* d, const float* packed, const void* dy, float* dweight,
                                  float* dbias, float* sums, void* stream) {
    return conv3d_bwd_impl(d, packed, dy, dweight, dbias, sums, stream, 0, 1.0f, 0, d ? d->D : 0);
}

extern "C" int passion_conv3d_bwd_phase(const passion_conv_desc_t* d, const float* packed, const void* dy, float* dweight,
                                        float* dbias, float* sums, int phase, float vscale, int dwin_lo, int dwin_hi, void* stream) {
    if (phase != 1 && phase != 2) return -1003;
    return conv3d_bwd_impl(d, packed, dy, dweight, dbias, sums, stream, phase, vscale, dwin_lo, dwin_hi);
}
