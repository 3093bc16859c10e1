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

namespace pconv { int g_conv_s2 = 1; }      // PASSION_CONV_S2=0 / option "conv_s2": back to the zero-stuffed lean kernel (A/B)
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
    for (int i = tid; i < NVOX * K8; i += 256) {
        const int v = i / K8, blk = i % K8;
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
        f32x4 acc[2][MT];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[t][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s * 4 < npair; ++s) {
            const int pr = s * 4 + g;
            const bool pv = pr < npair;
            const int ent = pv ? pr / K8 : 0, blk = pv ? pr % K8 : 0;
            // tap of the class: index a per axis -> (t, local dY offset): odd: t = 1, +1; even: a = 0: t = 0, +1; a = 1: t = 2, +0
            const int aw = ent % nw, ah = (ent / nw) % nh, ad = ent / (nw * nh);
            const int tdp = pd ? 1 : 2 * ad, thp = ph ? 1 : 2 * ah, twp = pw ? 1 : 2 * aw;
            const int ld = pd ? 1 : 1 - ad, lh = ph ? 1 : 1 - ah, lw = pw ? 1 : 1 - aw;
            const int tap = (tdp * 3 + thp) * 3 + twp;
            bf16x8 a[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[mt] = pv ? gload<bf16x8>(wbase + ((long)tap * K8 + blk) * p.Mpad + mt * 16 + n) : zero8();
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                // voxel (id = t, ih, iw) of the class reads dY local ((t + ld), (ih + lh), (iw + lw))
                const int v = ((t + ld) * OH + (ih + lh)) * OW + (iw + lw);
                const bf16x8 bv = pv ? *reinterpret_cast<const bf16x8*>(lds + v * RS + blk * 16) : zero8();
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], bv, acc[t][mt], 0, 0, 0);
            }
        }
        // ---- epilogue of the class: back through LeakyReLU / mask, sums, store into the padded scratch
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int qd = q0d + pd + 2 * t, qh = q0h + ph + 2 * ih, qw = q0w + pw + 2 * iw;
            const bool valid = qd <= Sd + 1 && qh <= Sh + 1 && qw <= Sw + 1;
            const int xd = min(max(reflect_idx(qd - 1, Sd), 0), Sd - 1), xh_ = min(max(reflect_idx(qh - 1, Sh), 0), Sh - 1),
                      xw = min(max(reflect_idx(qw - 1, Sw), 0), Sw - 1);
            const long xvox = ((long)xd * Sh + xh_) * Sw + xw;
            const long ovox = ((long)qd * (Sh + 2) + qh) * Wp + qw;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int c = mt * 16 + g * 4;
                const bool ok = valid && rowok[mt];
                u32x2 xr = {0u, 0u};
                if (ok && hs) xr = gload<u32x2>(xb + xvox * sd.C + c);
                const float xv[4] = {__uint_as_float(xr.x << 16), __uint_as_float(xr.x & 0xffff0000u),
                                     __uint_as_float(xr.y << 16), __uint_as_float(xr.y & 0xffff0000u)};
                const float vf = (ok && hs) ? 1.f : 0.f;
                f32x4 gv = acc[t][mt];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xh = (xv[j] - mean[mt][j]) * rstd[mt][j];
                    const float gg = gv[j] * (xh > 0.f ? f_pos : f_neg);
                    gv[j] = gg;
                    const float gs = gg * vf;
                    s1[mt][j] += gs;
                    s2[mt][j] = __builtin_fmaf(gs, xh, s2[mt][j]);
                }
                if (ok) store4(outb + ovox * p.Cin + c, gv);
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
    k.Mpad = pad16(k.Cin);
    k.wpack = reinterpret_cast<const bf16_t*>(packed_bf16) + (long)27 * pad8(k.Cin) * pad16(k.Cout);      // the dgrad image follows the forward one
    k.sums_rep = SUMS_REP_LEAN;
    k.sums_fixed = k.sums_float_only ? 0 : 1;
    const int mt = k.Mpad / 16;
    if (mt == 1) return launch_s2<1>(k, K8, st);
    if (mt == 2) return launch_s2<2>(k, K8, st);
    if (mt == 4) return launch_s2<4>(k, K8, st);
    return 1;
}

}  // namespace pconv
