// Gather-form input gradient of the STRIDE-2 3x3x3 reflect-padded pre-norm convs (the first conv of encoder levels 2..5,
// reference code/models/mmformer.py:33,38,43,48 = general_conv3d_prenorm(..., stride=2), blocks.py:300-316), bf16, over the padded
// domain (same contract as conv_lean.hip MODE 2: the gradient of padded position q - 1 lands at coordinate q of the
// [B][D+2][H+2][pad8(W+2)][Cin] scratch, the finish kernel folds the halo planes; InstanceNorm-backward sums as 2^-40 fixed-point
// integer atomics).
//
// Why a kernel of its own: the lean kernel runs all 27 taps over a ZERO-STUFFED dY brick -- 7 of every 8 staged values and MFMA
// steps multiply zeros (the grouped e2_c1 launch: 398 us for 168 MB = 5 % of 8 TB/s, the slowest launch of a step).  The forward
// reads padded coordinate 2 o + t - 1, so an input position with scratch coordinate q receives, per axis,
//     q even:  t = 0 from o = q / 2   and   t = 2 from o = q / 2 - 1;        q odd:  t = 1 from o = (q - 1) / 2,
// i.e. 1, 2, 4 or 8 taps depending on the PARITY CLASS (q_d, q_h, q_w) mod 2 -- 27 taps over the eight classes together instead of
// 27 each.  A workgroup owns a 4 x 8 x 8 brick of q, stages the 3 x 5 x 5 dY voxels it needs ONCE (no stuffing), and every
// 16-voxel MFMA column tile holds voxels of one class, so only that class's taps are issued:
//     wave 0: (e,e,e) 8 taps      wave 1: (e,e,o), (e,o,e) 4 + 4      wave 2: (o,e,e) 4, (e,o,o) 2, (o,e,o) 2      wave 3: (o,o,e) 2, (o,o,o) 1
// rows = input channels (MT row tiles of 16), K = (tap, 8-channel block of Cout) pairs, four per v_mfma_f32_16x16x32_bf16.
#include "conv_common.h"
#include "prof.h"

namespace pconv { int g_conv_s2 = 1; extern int g_conv_target; }      // PASSION_CONV_S2=0 / option "conv_s2": back to the zero-stuffed lean kernel (A/B)
using namespace pconv;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int BD = 4, BH = 8, BW = 8;                 // brick of scratch coordinates q (even origin)
constexpr int OD = BD / 2 + 1, OH = BH / 2 + 1, OW = BW / 2 + 1;      // dY voxels per axis: o = q0 / 2 - 1 .. q0 / 2 + B / 2 - 1
constexpr int NVOX = OD * OH * OW;                    // 75

template <typename T> __device__ __forceinline__ T gload(const void* ptr) {
    return *reinterpret_cast<const T __attribute__((address_space(1)))*>(reinterpret_cast<unsigned long long>(ptr));
}
__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (__bf16)0.f;
    return z;
}

// classes of a wave: up to three (pd, ph, pw) parities, bit-packed 3 bits each, count in the top bits
__device__ __forceinline__ int wave_classes(int wv, int& n) {
    // parity bit = 1: ODD coordinate (one tap), 0: even (two taps); order (pd, ph, pw) = bits 2, 1, 0
    if (wv == 0) { n = 1; return 0b000; }
    if (wv == 1) { n = 2; return 0b001 | (0b010 << 3); }
    if (wv == 2) { n = 3; return 0b100 | (0b011 << 3) | (0b101 << 6); }
    n = 2; return 0b110 | (0b111 << 3);
}

template <int MT>
__global__ __launch_bounds__(256) void conv_s2_dgrad_kernel(ConvK p, int K8) {
    extern __shared__ __attribute__((aligned(16))) char lds[];      // [NVOX][K8 * 16 bytes + 16 pad]
    __shared__ float lsum[4 * MT * 16 * 2];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Sd = p.Di, Sh = p.Hi, Sw = p.Wi;                      // input (gradient) volume
    const int Qd = p.Do, Qh = p.Ho, Qw = p.Wo;                      // dY volume
    const int RS = K8 * 16 + 16;                                    // LDS bytes per staged voxel (pad: rows 16 bytes apart mod 128)

    int bx = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tiles_b = p.tilesD * p.tilesH * p.tilesW;
    const int b = bx / tiles_b;
    int r = bx % tiles_b;
    const int tw_ = r % p.tilesW; r /= p.tilesW;
    const int th_ = r % p.tilesH;
    const int td_ = r / p.tilesH;
    const int q0d = td_ * BD, q0h = th_ * BH, q0w = tw_ * BW;
    const int o0d = q0d / 2 - 1, o0h = q0h / 2 - 1, o0w = q0w / 2 - 1;

    // ---- stage the dY block (zeros outside the volume): 16-byte chunks, voxel-major
    const bf16_t* dy = reinterpret_cast<const bf16_t*>(p.dy) + (long)b * Qd * Qh * Qw * p.Cout;
    const int k8s = 31 - __builtin_clz(K8);                         // K8 is a power of two (launcher): index splits are shifts, not divisions
    for (int i = tid; i < NVOX * K8; i += 256) {
        const int v = i >> k8s, blk = i & (K8 - 1);
        const int zw = v % OW, zh = (v / OW) % OH, zd = v / (OW * OH);
        const int od = o0d + zd, oh = o0h + zh, ow = o0w + zw;
        u32x4 val = {0u, 0u, 0u, 0u};
        if (od >= 0 && od < Qd && oh >= 0 && oh < Qh && ow >= 0 && ow < Qw)
            val = gload<u32x4>(dy + (((long)od * Qh + oh) * Qw + ow) * p.Cout + blk * 8);
        *reinterpret_cast<u32x4*>(lds + v * RS + blk * 16) = val;
    }

    // ---- epilogue constants of this lane's four rows per row tile (input channels mt * 16 + 4 g .. + 3): one input segment
    const SegD& sd = p.seg[0];
    const int bs = sd.bsrc(b);
    float mval = 1.f;
    if (sd.mask) mval = sd.mask[(long)b * sd.mask_ld + sd.mask_col] ? 1.f : 0.f;
    const bool hs = sd.stats != nullptr;
    f32x4 mean[MT], rstd[MT];
    bool rowok[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int c = mt * 16 + g * 4;
        rowok[mt] = c < p.Cin;
        mean[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        rstd[mt] = (f32x4){1.f, 1.f, 1.f, 1.f};
        if (hs && rowok[mt]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                mean[mt][j] = sd.stats[((long)bs * sd.C + c + j) * 2 + 0];
                rstd[mt][j] = sd.stats[((long)bs * sd.C + c + j) * 2 + 1];
            }
        }
    }
    const float f_pos = mval, f_neg = hs ? mval * p.slope : mval;
    const bf16_t* xb = reinterpret_cast<const bf16_t*>(sd.x) + (long)bs * sd.bstride;
    const int Wp = pad8(Sw + 2);
    bf16_t* outb = reinterpret_cast<bf16_t*>(p.y) + (long)b * (Sd + 2) * (Sh + 2) * Wp * p.Cin;
    const bf16x8* wbase = reinterpret_cast<const bf16x8*>(p.wpack) + wvec_of(p, b);

    float s1[MT][4], s2[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) { s1[mt][j] = 0.f; s2[mt][j] = 0.f; }
    __syncthreads();

    int ncls;
    const int clist = wave_classes(wv, ncls);
    const int ih = n >> 2, iw = n & 3;                               // the lane's voxel inside a class tile: (id = tile, ih, iw)
    for (int ci = 0; ci < ncls; ++ci) {
        const int cls = (clist >> (3 * ci)) & 7;
        const int pd = (cls >> 2) & 1, ph = (cls >> 1) & 1, pw = cls & 1;         // 1 = odd coordinate
        const int nd = 2 - pd, nh = 2 - ph, nw = 2 - pw;
        const int npair = nd * nh * nw * K8;
        // One memory latency per class instead of one per K step and one more in the epilogue: the x operands of the epilogue and
        // the first PDW steps' weight fragments are requested together, before anything waits (the first version loaded each
        // step's weights right in front of its MFMAs and x after them: the grouped e2_c1 launch spent 270 us on 168 MB).
        constexpr int PDW = 4;
        bool okx[2];
        long ovox[2];
        u32x2 xr[2][MT];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int qd = q0d + pd + 2 * t, qh = q0h + ph + 2 * ih, qw = q0w + pw + 2 * iw;
            okx[t] = qd <= Sd + 1 && qh <= Sh + 1 && qw <= Sw + 1;
            const int xd = min(max(reflect_idx(qd - 1, Sd), 0), Sd - 1), xh_ = min(max(reflect_idx(qh - 1, Sh), 0), Sh - 1),
                      xw = min(max(reflect_idx(qw - 1, Sw), 0), Sw - 1);
            const long xvox = ((long)xd * Sh + xh_) * Sw + xw;
            ovox[t] = ((long)qd * (Sh + 2) + qh) * Wp + qw;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)       // (unconditional: a clamped voxel of a valid tensor; rows past Cin read channel 0)
                xr[t][mt] = gload<u32x2>(xb + xvox * sd.C + (rowok[mt] ? mt * 16 + g * 4 : 0));
        }
        const int nstep = (npair + 3) / 4;
        bf16x8 aw[PDW][MT];
        int vsel[PDW], bsel[PDW];                 // per step: the lane's dY voxel offset (t = 0) and K block, -1 = no pair
        auto wload = [&](int s, int u) {
            const int pr = s * 4 + g;
            const bool pv = pr < npair;
            const int ent = pv ? pr >> k8s : 0, blk = pv ? pr & (K8 - 1) : 0;
            // tap of the class: index a per axis -> (t, local dY offset): odd: t = 1, +1; even: a = 0: t = 0, +1; a = 1: t = 2, +0
            // (nw, nh in {1, 2}: log2 = n - 1)
            const int aw_ = ent & (nw - 1), ah = (ent >> (nw - 1)) & (nh - 1), ad = ent >> ((nw - 1) + (nh - 1));
            const int tdp = pd ? 1 : 2 * ad, thp = ph ? 1 : 2 * ah, twp = pw ? 1 : 2 * aw_;
            const int ld = pd ? 1 : 1 - ad, lh = ph ? 1 : 1 - ah, lw = pw ? 1 : 1 - aw_;
            const int tap = (tdp * 3 + thp) * 3 + twp;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) aw[u][mt] = gload<bf16x8>(wbase + ((long)tap * K8 + blk) * p.Mpad + mt * 16 + n);
            vsel[u] = pv ? ((ld * OH + (ih + lh)) * OW + (iw + lw)) * RS + blk * 16 : -1;
            bsel[u] = blk;
        };
#pragma unroll
        for (int u = 0; u < PDW; ++u) if (u < nstep) wload(u, u);
        f32x4 acc[2][MT];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[t][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int s0 = 0; s0 < nstep; s0 += PDW) {
#pragma unroll
            for (int u = 0; u < PDW; ++u) {
                const int s = s0 + u;
                if (s >= nstep) break;
                const bool pv = vsel[u] >= 0;
                bf16x8 a[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a[mt] = pv ? aw[u][mt] : zero8();
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    // voxel (id = t, ih, iw) of the class reads dY local ((t + ld), (ih + lh), (iw + lw))
                    const bf16x8 bv = pv ? *reinterpret_cast<const bf16x8*>(lds + vsel[u] + t * OH * OW * RS) : zero8();
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], bv, acc[t][mt], 0, 0, 0);
                }
                if (s + PDW < nstep) wload(s + PDW, u);
            }
        }
        // ---- epilogue of the class: back through LeakyReLU / mask, sums, store into the padded scratch
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int c = mt * 16 + g * 4;
                const bool ok = okx[t] && rowok[mt];
                const u32x2 xq = xr[t][mt];
                const float xv[4] = {__uint_as_float(xq.x << 16), __uint_as_float(xq.x & 0xffff0000u),
                                     __uint_as_float(xq.y << 16), __uint_as_float(xq.y & 0xffff0000u)};
                const float vf = (ok && hs) ? 1.f : 0.f;
                f32x4 gv = acc[t][mt];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xh = hs ? (xv[j] - mean[mt][j]) * rstd[mt][j] : 1.f;
                    const float gg = gv[j] * (xh > 0.f ? f_pos : f_neg);
                    gv[j] = gg;
                    const float gs = gg * vf;
                    s1[mt][j] += gs;
                    s2[mt][j] = __builtin_fmaf(gs, xh, s2[mt][j]);
                }
                if (ok) store4(outb + ovox[t] * p.Cin + c, gv);
            }
        }
    }

    // ---- InstanceNorm-backward sums: 16 voxel lanes (shuffles) -> 4 waves (LDS, fixed order) -> one fixed-point atomic per value
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a1 = group16_sum(s1[mt][j]), a2 = group16_sum(s2[mt][j]);
            if (n == 0) {
                lsum[((wv * MT + mt) * 16 + g * 4 + j) * 2 + 0] = hs ? a1 : 0.f;
                lsum[((wv * MT + mt) * 16 + g * 4 + j) * 2 + 1] = hs ? a2 : 0.f;
            }
        }
    __syncthreads();
    for (int i = tid; i < MT * 16 * 2; i += 256) {
        const int row = i >> 1, k2 = i & 1;
        const float t = ((lsum[(0 * MT * 16 + row) * 2 + k2] + lsum[(1 * MT * 16 + row) * 2 + k2]) + lsum[(2 * MT * 16 + row) * 2 + k2]) +
                        lsum[(3 * MT * 16 + row) * 2 + k2];
        if (row < p.Cin && t != 0.f) {
            const long idx = (long)(blockIdx.x % p.sums_rep) * p.B * p.Cin * 2 + ((long)b * p.Cin + row) * 2 + k2;
            if (p.sums_fixed) {
                double v = (double)t * 1099511627776.0;                          // 2^40
                v = fmin(fmax(v, -9.0e18), 9.0e18);
                atomicAdd(reinterpret_cast<unsigned long long*>(p.sums) + idx, (unsigned long long)(long long)llrint(v));
            } else {
                atomicAdd(&p.sums[idx], t);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same gradient for the first stride-2 layer of the encoders (8 -> 16 channels from 128^3 to 64^3, e2_c1: the slowest launch of
// a training step on the brick kernel above -- 38 K workgroups of 256 outputs each, 260 us for 168 MB) as a MARCHING kernel.
// Written in m = q >> 1 (q = 2 m + p, p = the parity class per axis) the gather is a stride-1 correlation over the dY grid:
//     dA[2 m + p][ci] = sum_{a in {0,1}^3, co} W'[(p, ci)][(a, co)] dY[m - a][co],
//     per axis: p even: a = 0 -> tap 0, a = 1 -> tap 2;  p odd: a = 0 -> tap 1, a = 1 -> no tap,
// i.e. ONE set of B fragments (the 2 x 2 x 2 neighbourhood of dY[m], 8 taps x 16 channels = four K steps of 32) serves all eight
// classes, and the eight classes x 8 input channels are the 64 rows of four MFMA row tiles -- no idle rows, no zero stuffing.  A
// workgroup owns an 8 x 16 tile of the (m_h, m_w) plane and marches along m_d with two dY planes live in a four-slot LDS ring
// (planes fetched four steps ahead by unconditional loads, x for the epilogue two steps ahead); lane (n, g) of row tile mt holds
// class (mt >> 1, mt & 1, g >> 1), channels 4 (g & 1) .. + 3 of m_w = w0 + n: its store is 8 bytes of a 512-byte contiguous run
// of the padded scratch.  Sums: one slot per workgroup, folded by the last workgroup of a sample (as conv_march.hip).
constexpr int SM_TH = 8, SM_TW = 16;                  // tile of m positions
constexpr int SM_EH = SM_TH + 1, SM_EW = SM_TW + 1;   // staged dY rows / columns: m - 1 .. m + T - 1
constexpr int SM_VS = 32;                             // bytes per staged voxel (16 channels)
constexpr int SM_RSB = SM_EW * SM_VS;
constexpr int SM_PLANE = 8192, SM_NSLOT = 4, SM_RING = SM_NSLOT * SM_PLANE;
constexpr int SM_SV = SM_EH * SM_EW, SM_CH = SM_SV * 2, SM_NIT = (SM_CH + 255) / 256;
static_assert(SM_EH * SM_RSB <= SM_PLANE, "slot size");

struct S2MP { int nseg_d, planes_per_seg, tilesH, tilesW, nwg_b; unsigned* tickets; };

template <typename T> __device__ __forceinline__ T gload_at(const void* base, unsigned off) {
    return *reinterpret_cast<const T __attribute__((address_space(1)))*>(reinterpret_cast<unsigned long long>(base) + off);
}
template <typename T> __device__ __forceinline__ void gstore_at(void* base, unsigned off, T v) {
    *reinterpret_cast<T __attribute__((address_space(1)))*>(reinterpret_cast<unsigned long long>(base) + off) = v;
}

__global__ __launch_bounds__(256) void conv_s2_march_kernel(ConvK p, S2MP mp) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    __shared__ float lsum[4 * 4 * 4 * 2 + 16];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = (lane >> 4) & 3;
    const int half = g >> 1, cgl = (g & 1) * 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Sd = p.Di, Sh = p.Hi, Sw = p.Wi;                      // input (gradient) volume
    const int Qd = p.Do, Qh = p.Ho, Qw = p.Wo;                      // dY volume; m runs over 0 .. Q per axis

    const int bx = xcd_swizzle(blockIdx.x, gridDim.x);
    const int b = bx / mp.nwg_b;
    int r = bx % mp.nwg_b;
    const int tw_i = r % mp.tilesW; r /= mp.tilesW;
    const int th_i = r % mp.tilesH; r /= mp.tilesH;
    const int h0 = th_i * SM_TH, w0 = tw_i * SM_TW;
    const int mlo = r * mp.planes_per_seg, mhi = min(mlo + mp.planes_per_seg, Qd + 1), L = mhi - mlo;
    if (L <= 0) return;

    // ---- staging: chunk c = (voxel, 8-channel block) of the 9 x 17 dY voxels of a plane
    const char* src = reinterpret_cast<const char*>(reinterpret_cast<const bf16_t*>(p.dy) + (long)b * Qd * Qh * Qw * 16);
    unsigned srcel[SM_NIT];
    int dstoff[SM_NIT];
    unsigned keepbits = 0u;
#pragma unroll
    for (int u = 0; u < SM_NIT; ++u) {
        const int c = min(u * 256 + tid, SM_CH - 1), sv = c >> 1, blk = c & 1;
        const int zh = sv / SM_EW, zw = sv % SM_EW;
        const int oh = h0 - 1 + zh, ow = w0 - 1 + zw;
        const bool ok = u * 256 + tid < SM_CH && oh >= 0 && oh < Qh && ow >= 0 && ow < Qw;
        srcel[u] = ((unsigned)(min(max(oh, 0), Qh - 1) * Qw + min(max(ow, 0), Qw - 1)) * 16u + (unsigned)blk * 8u) * 2u;
        dstoff[u] = zh * SM_RSB + zw * SM_VS + blk * 16;
        if (ok) keepbits |= 1u << u;
    }
    const unsigned plane_b = (unsigned)(Qh * Qw) * 32u;
    // virtual dY planes of this segment: o = mlo - 1 + j, j = 0 .. L; output plane i (m_d = mlo + i) reads j = i + 1 and j = i
    auto plane_of = [&](int j, bool& live) {
        const int o = mlo - 1 + j;
        live = j <= L && o >= 0 && o < Qd;
        return min(max(o, 0), Qd - 1);
    };
    u32x4 raw[4][SM_NIT];
    auto fetch = [&](int j, u32x4 (&dst)[SM_NIT]) {
        bool live;
        const unsigned pb = (unsigned)plane_of(j, live) * plane_b;
#pragma unroll
        for (int u = 0; u < SM_NIT; ++u) dst[u] = gload_at<u32x4>(src, pb + srcel[u]);
    };
#pragma unroll
    for (int j = 0; j < 4; ++j) fetch(j, raw[j]);
    auto commit = [&](int j, const u32x4 (&rv)[SM_NIT]) {
        char* slot = lds + (j & (SM_NSLOT - 1)) * SM_PLANE;
        bool live;
        plane_of(j, live);
        const unsigned keep = live ? keepbits : 0u;
#pragma unroll
        for (int u = 0; u < SM_NIT; ++u) {
            u32x4 o = rv[u];
            if (!((keep >> u) & 1u)) o = (u32x4){0u, 0u, 0u, 0u};
            if (u * 256 + tid < SM_CH) *reinterpret_cast<u32x4*>(slot + dstoff[u]) = o;
        }
    };

    // ---- weights: A fragment of (row tile mt, K step s); row 16 mt + n = (class 2 mt + (n >> 3), input channel n & 7), K slice
    // 4 s + g = (tap a = (4 s + g) >> 1 = (ad, ah, aw), channel block (4 s + g) & 1)
    const bf16x8* wbase = reinterpret_cast<const bf16x8*>(p.wpack) + wvec_of(p, b);
    bf16x8 wfr[4][4];
    int pre[4];                                                     // B fragment of step s, tile row 0: LDS offset for ring slot 0
    const int r0 = 2 * wv;
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
        const int pr = 4 * s2 + g, a = pr >> 1, blk = pr & 1, ad = a >> 2, ah = (a >> 1) & 1, aw = a & 1;
        pre[s2] = ((SM_NSLOT - ad) & (SM_NSLOT - 1)) * SM_PLANE + (r0 + 1 - ah) * SM_RSB + (n + 1 - aw) * SM_VS + blk * 16;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int cls = 2 * mt + (n >> 3), pd = cls >> 2, ph = (cls >> 1) & 1, pw = cls & 1;
            const bool ok = !(pd && ad) && !(ph && ah) && !(pw && aw) && (n & 7) < p.Cin;
            const int td = pd ? 1 : 2 * ad, th = ph ? 1 : 2 * ah, tw = pw ? 1 : 2 * aw;
            wfr[mt][s2] = gload<bf16x8>(wbase + ((long)((td * 3 + th) * 3 + tw) * 2 + blk) * p.Mpad + (n & 7));
            if (!ok) wfr[mt][s2] = zero8();
        }
    }
    // ---- epilogue constants: the lane's four input channels cgl .. of the one input segment
    const SegD& sd = p.seg[0];
    const int bs = sd.bsrc(b);
    const bool hs = sd.stats != nullptr, row_ok = cgl < p.Cin;
    float mval = 1.f;
    if (sd.mask) mval = gload<uint8_t>(sd.mask + (long)b * sd.mask_ld + sd.mask_col) ? 1.f : 0.f;
    f32x4 ersd = {0.f, 0.f, 0.f, 0.f}, eshf = {1.f, 1.f, 1.f, 1.f};
    if (hs && row_ok) {
        const float* esp = sd.stats + ((long)bs * sd.C + cgl) * 2;
        const f32x4 a = gload<f32x4>(esp), c2 = gload<f32x4>(esp + 4);
        ersd = (f32x4){a[1], a[3], c2[1], c2[3]};
        eshf = (f32x4){-a[0] * a[1], -a[2] * a[3], -c2[0] * c2[1], -c2[2] * c2[3]};
    }
    const float f_pos = row_ok ? mval : 0.f, f_neg = hs ? f_pos * p.slope : f_pos;
    for (int i = tid; i < SM_RING / 16; i += 256) reinterpret_cast<u32x4*>(lds)[i] = (u32x4){0u, 0u, 0u, 0u};

    const int Wp = pad8(Sw + 2);
    const char* xbase = reinterpret_cast<const char*>(reinterpret_cast<const bf16_t*>(sd.x) + (long)bs * sd.bstride);
    char* ybase = reinterpret_cast<char*>(reinterpret_cast<bf16_t*>(p.y) + (long)b * (Sd + 2) * (Sh + 2) * Wp * p.Cin);
    const unsigned eCb = hs ? (unsigned)sd.C * 2u : 0u, elane = (row_ok && hs) ? (unsigned)cgl * 2u : 0u;
    // in-plane offsets of the lane's output per (row parity ph, tile row t): q_h = 2 (h0 + r0 + t) + ph, q_w = 2 (w0 + n) + half
    unsigned oel[2][2], xel[2][2], vbits = 0u;
    const int qw = 2 * (w0 + n) + half;
#pragma unroll
    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int qh = 2 * (h0 + r0 + t) + ph;
            if (qh <= Sh + 1 && qw <= Sw + 1) vbits |= 1u << (ph * 2 + t);
            oel[ph][t] = ((unsigned)(qh * Wp + qw) * (unsigned)p.Cin + (unsigned)cgl) * 2u;
            const int xh = min(max(reflect_idx(qh - 1, Sh), 0), Sh - 1), xw = min(max(reflect_idx(qw - 1, Sw), 0), Sw - 1);
            xel[ph][t] = (unsigned)(xh * Sw + xw) * eCb + elane;
        }
    const unsigned oplane_b = (unsigned)((Sh + 2) * Wp) * (unsigned)p.Cin * 2u, xplane_b = (unsigned)(Sh * Sw) * eCb;
    u32x2 eop[2][8];
    auto eload = [&](int i, u32x2 (&dst)[8]) {
#pragma unroll
        for (int pd = 0; pd < 2; ++pd) {
            const int qd = 2 * (mlo + i) + pd;
            const unsigned xo = __umul24((unsigned)min(max(reflect_idx(qd - 1, Sd), 0), Sd - 1), xplane_b);
#pragma unroll
            for (int ph = 0; ph < 2; ++ph)
#pragma unroll
                for (int t = 0; t < 2; ++t) dst[(pd * 2 + ph) * 2 + t] = gload_at<u32x2>(xbase, xo + xel[ph][t]);
        }
    };
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2v[4] = {0.f, 0.f, 0.f, 0.f};

    // output plane i (m_d = mlo + i: q_d = 2 m_d, 2 m_d + 1) from ring planes j = i + 1 (dY[m_d]) and j = i (dY[m_d - 1])
    auto compute = [&](int i, const u32x2 (&eo)[8]) {
        const unsigned sbv = (unsigned)((i + 1) & (SM_NSLOT - 1)) * (unsigned)SM_PLANE;
        f32x4 acc[4][2];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        bf16x8 cur[2], nxt[2];
        auto bload = [&](int s2, bf16x8 (&dst)[2]) {
            const char* base = lds + (((unsigned)pre[s2] + sbv) & (unsigned)(SM_RING - 1));
            dst[0] = *reinterpret_cast<const bf16x8*>(base);
            dst[1] = *reinterpret_cast<const bf16x8*>(base + SM_RSB);
        };
        bload(0, cur);
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            if (s2 + 1 < 4) bload(s2 + 1, nxt);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[mt][s2], cur[t], acc[mt][t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            cur[0] = nxt[0]; cur[1] = nxt[1];
        }
        if (i >= L) return;                                          // (planes past the segment: nothing stored or counted)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int pd = mt >> 1, ph = mt & 1, qd = 2 * (mlo + i) + pd;
            const bool dval = qd <= Sd + 1;
            const unsigned ob = (unsigned)qd * oplane_b;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const bool valid = dval && ((vbits >> (ph * 2 + t)) & 1u);
                const float fp = valid ? f_pos : 0.f, fn = valid ? f_neg : 0.f;
                const u32x2 xr = eo[mt * 2 + t];
                const float xv[4] = {__uint_as_float(xr.x << 16), __uint_as_float(xr.x & 0xffff0000u),
                                     __uint_as_float(xr.y << 16), __uint_as_float(xr.y & 0xffff0000u)};
                union { bf16_t h[4]; u32x2 u; } pk;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xh = __builtin_fmaf(xv[j], ersd[j], eshf[j]);
                    const float gg = acc[mt][t][j] * (xh > 0.f ? fp : fn);
                    pk.h[j] = (bf16_t)gg;
                    s1[j] += gg;
                    s2v[j] = __builtin_fmaf(gg, xh, s2v[j]);
                }
                if (valid && row_ok) gstore_at<u32x2>(ybase, ob + oel[ph][t], pk.u);
            }
        }
    };

    // ---- pipeline.  Iteration j: barrier; commit plane j; fetch plane j + 4; compute output plane j - 2 (ring planes j - 1, j - 2,
    // committed before this barrier); fetch the epilogue operands of output plane j
    eload(0, eop[0]);
    eload(1, eop[1]);
    __syncthreads();                                                 // ring zeroed before the first commit
    commit(0, raw[0]); fetch(4, raw[0]);
    __syncthreads();
    commit(1, raw[1]); fetch(5, raw[1]);
    for (int j0 = 2; j0 < L + 2; j0 += 4) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = j0 + jj;                                   // plane j sits in raw[(2 + jj) & 3], output j - 2 in operand slot jj & 1
            __syncthreads();
            commit(j, raw[(2 + jj) & 3]);
            fetch(j + 4, raw[(2 + jj) & 3]);
            compute(j - 2, eop[jj & 1]);
            eload(j, eop[jj & 1]);
        }
    }

    // ---- InstanceNorm-backward sums: 16 voxel lanes (shuffles) -> 4 waves x 2 lane groups (LDS, fixed order) -> this workgroup's slot
    const float keep = hs ? 1.f : 0.f;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a1 = group16_sum(s1[j] * keep), a2 = group16_sum(s2v[j] * keep);
        if (n == 0) {
            lsum[((wv * 4 + g) * 4 + j) * 2 + 0] = a1;
            lsum[((wv * 4 + g) * 4 + j) * 2 + 1] = a2;
        }
    }
    __syncthreads();
    if (tid < 16) {
        const int c = tid >> 1, k2 = tid & 1, gq = c >> 2, j = c & 3;
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) t += lsum[((w * 4 + gq) * 4 + j) * 2 + k2] + lsum[((w * 4 + gq + 2) * 4 + j) * 2 + k2];
        if (c < p.Cin) {
            float* rs = p.sums + (long)(bx % mp.nwg_b) * p.B * p.Cin * 2;
            st_agent(&rs[((long)b * p.Cin + c) * 2 + k2], t);
        }
    }
    {
        __shared__ int last_wg;
        __shared__ double fpart[256];
        stores_done();
        __syncthreads();
        if (tid == 0) last_wg = atomicAdd(&mp.tickets[b], 1u) == (unsigned)(mp.nwg_b - 1);
        __syncthreads();
        if (last_wg) {
            acquire_agent();
            fold_sums_slots(p.sums, b, p.B, p.Cin, mp.nwg_b, tid, fpart);
            if (tid == 0) __hip_atomic_store(&mp.tickets[b], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// 0 = launched, 1 = not this shape
int launch_s2_march(ConvK& k, hipStream_t st) {
    static const bool on = []() { const char* e = getenv("PASSION_S2_MARCH"); return !(e && e[0] == '0'); }();
    if (!on || k.Cin != 8 || k.Cout != 16 || k.Mpad != 16) return 1;
    const long pvox = (long)(k.Di + 2) * (k.Hi + 2) * pad8(k.Wi + 2);
    if ((long)k.Do * k.Ho * k.Wo * 32 >= (1L << 32) || pvox * k.Cin * 2 >= (1L << 32) || (long)k.Di * k.Hi * k.Wi * k.Cin * 2 >= (1L << 32)) return 1;
    S2MP mp;
    mp.tilesH = (k.Ho + 1 + SM_TH - 1) / SM_TH; mp.tilesW = (k.Wo + 1 + SM_TW - 1) / SM_TW;
    const int tiles = mp.tilesH * mp.tilesW, dcount = k.Do + 1;
    // D segments: one workgroup per CU is resident (272 registers), so the launch runs in rounds of 256 workgroups and a segment pays two
    // halo planes: the split that minimises rounds x (planes per segment + 2) among those the target allows (round 6, 4 x 128^3: six
    // segments = 1080 workgroups = 4.2 rounds 111 us, four = 720 = 2.8 rounds 101; five 109, seven 110, eight 124)
    static const int s2_nseg = []() { const char* e = getenv("PASSION_S2_NSEG"); return e ? atoi(e) : 0; }();
    int nseg_max = (int)hmax(1, (g_conv_target + (long)k.B * tiles - 1) / ((long)k.B * tiles));
    nseg_max = (int)hmin(nseg_max, hmax(1, dcount / 8));
    nseg_max = (int)hmin(nseg_max, hmax(1, SUMS_SLOTS / tiles));
    int nseg = 1;
    long best = -1;
    for (int n = 1; n <= nseg_max; ++n) {
        const long rounds = ((long)k.B * tiles * n + 255) / 256, cost = rounds * ((dcount + n - 1) / n + 2);
        if (best < 0 || cost < best) { best = cost; nseg = n; }
    }
    if (s2_nseg > 0) nseg = (int)hmin(s2_nseg, nseg_max);
    mp.planes_per_seg = (dcount + nseg - 1) / nseg;
    mp.nseg_d = (dcount + mp.planes_per_seg - 1) / mp.planes_per_seg;
    mp.nwg_b = mp.nseg_d * tiles;
    if (mp.nwg_b > SUMS_SLOTS) return 1;
    void* zero = nullptr;
    const int rc = stream_scratch(st, (size_t)k.B * sizeof(unsigned), 0, &zero, nullptr);
    if (rc) return rc;
    mp.tickets = static_cast<unsigned*>(zero);
    k.sums_rep = 1;
    k.sums_fixed = 0;
    static bool granted = false;
    if (!granted) {
        HIP_RET(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_s2_march_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SM_RING));
        granted = true;
    }
    passion_prof_tag("conv_s2_march_kernel");
    hipLaunchKernelGGL(conv_s2_march_kernel, dim3(k.B * mp.nwg_b), dim3(256), SM_RING, st, k, mp);
    LAUNCH_RET();
    return 0;
}

template <int MT>
int launch_s2(ConvK& k, int K8, hipStream_t st) {
    const int bytes = NVOX * (K8 * 16 + 16);
    const long grid = (long)k.B * k.tilesD * k.tilesH * k.tilesW;
    passion_prof_tag("conv_s2_dgrad_kernel<%d>", MT);
    hipLaunchKernelGGL((conv_s2_dgrad_kernel<MT>), dim3((unsigned)grid), dim3(256), bytes, st, k, K8);
    LAUNCH_RET();
    return 0;
}

}  // namespace

namespace pconv {

// 0 = launched, < 0 = HIP error, 1 = not covered (the caller goes on to the zero-stuffed lean kernel).  k.tiles* cover the padded
// domain (S + 2)^3 in 4 x 8 x 8 bricks and k.y = the padded scratch, as for conv_lean.hip MODE 2.
int launch_conv_s2_dgrad(ConvK& k, const void* packed_bf16, hipStream_t st) {
    if (!g_conv_s2) return 1;
    if (k.nseg != 1 || k.seg[0].C != k.Cin || k.seg[0].bmod > 0 || !k.seg[0].dx) return 1;
    if (k.pad_mode != PASSION_PAD_REFLECT || k.y_f32 || k.bd_rows) return 1;
    if (k.Cin % 4 || k.Cin > 64 || k.Cout % 8 || k.Cout > 256) return 1;
    if ((k.Di | k.Hi | k.Wi) & 1) return 1;
    if (k.Do * 2 != k.Di || k.Ho * 2 != k.Hi || k.Wo * 2 != k.Wi) return 1;
    const int K8 = k.Cout / 8;
    if (K8 & (K8 - 1)) return 1;                                    // shifts instead of divisions in the kernel
    k.Mpad = pad16(k.Cin);
    k.wpack = reinterpret_cast<const bf16_t*>(packed_bf16) + (long)27 * pad8(k.Cin) * pad16(k.Cout);      // the dgrad image follows the forward one
    k.sums_rep = SUMS_REP_LEAN;
    k.sums_fixed = k.sums_float_only ? 0 : 1;
    {
        const int rc = launch_s2_march(k, st);                       // 8 -> 16 channels: the marching form
        if (rc <= 0) return rc;
    }
    k.sums_rep = SUMS_REP_LEAN;
    k.sums_fixed = k.sums_float_only ? 0 : 1;
    const int mt = k.Mpad / 16;
    if (mt == 1) return launch_s2<1>(k, K8, st);
    if (mt == 2) return launch_s2<2>(k, K8, st);
    if (mt == 4) return launch_s2<4>(k, K8, st);
    return 1;
}

}  // namespace pconv
