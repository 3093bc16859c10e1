// Depth-PAIR marching conv: forward (MODE 0) and padded-domain input gradient (MODE 2) of the 3x3x3 stride-1 convs with an
// 8-channel K axis and <= 8 output rows -- the 8 -> 8 layers of the 128^3 level (reference code/models/mmformer.py:28-38 encoder level
// 1, :150-189 decoder level 1; blocks.py:300-316 general_conv3d_prenorm), the launches `roofline` names.
//
// conv_march.hip runs these on v_mfma_f32_16x16x32_bf16 with HALF of the 16 MFMA rows empty (8 output channels), and the counters /
// the instruction stream say what that costs: per plane and wave 21 MFMAs against ~210 VALU instructions (epilogue arithmetic, LDS
// and global addresses, accumulator moves) -- the kernel is bound by VALU issue, and in the epilogue only the lanes of rows 0..7
// (32 of 64) hold anything.  Here the idle rows carry the NEXT output plane (the trick conv_wgrad_march.hip plays with its second
// depth tap):
//   D[8 h + co][voxel] = sum_{pd = 0..3} sum_{th, tw, ci} W'[8 h + co][(pd, th, tw, ci)] * X[plane j + pd][voxel + (th, tw)][ci],
//   W'[co][pd] = W[co][kd = pd] (pd <= 2), W'[8 + co][pd] = W[co][kd = pd - 1] (pd >= 1), zero otherwise:
// K = 4 x 9 x 8 = 288 = NINE MFMA steps for TWO output planes instead of 2 x 7, every accumulator lane is a real output (lane group g:
// plane g >> 1, channels 4 (g & 1) ..), so the epilogue's VALU work per output halves, and the barrier / staging / address
// bookkeeping is paid once per pair.  Everything else follows conv_march.hip: a workgroup owns an (8 x TW) tile of the (H, W)
// plane and marches along D, planes fetched ahead into registers by unconditional loads, normalised (InstanceNorm + LeakyReLU) on
// their way into an LDS ring (8 slots of a power-of-two size: the slot of plane j + pd is an add and an and), weights of the whole
// K axis in registers, MODE 2 with the halo outputs as extra column tiles through the zero guard ring and per-workgroup sums slots
// folded by the last workgroup of a sample.  Same operand images, scratch / sums contract and finish kernels.
#include <cstdlib>
#include <type_traits>
#include "conv_common.h"
#include "prof.h"

namespace pconv { extern int g_conv_target, g_conv_dbg; int g_conv_pair = 1; }
using namespace pconv;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int MTH = 8;                   // tile rows (two per wave)
constexpr int NSLOT = 8;                 // LDS ring slots (planes)

// NB: 8-channel blocks of the K axis (1; 2 = a 16-channel input, forward only: the decoders' 16 -> 8 convs over a concat of two
// 8-channel tensors); K steps of 32 = 4 (tap, block) pairs, 36 NB pairs
template <int TW, int NB = 1> struct GeoP {
    static constexpr int EH = MTH + 4, EW = TW + 4;           // staged plane: tile + halo 1 + zero guard ring 1
    static constexpr int VS = 16 * NB, RSB = EW * VS;
    static constexpr int PLANE = (TW == 16 ? 4096 : 8192) * NB;      // slot size (>= EH * RSB = 3840 / 6912 per block), a power of two
    static constexpr int SV = (MTH + 2) * (TW + 2);           // voxels fetched per plane
    static constexpr int CH = SV * NB;                        // 16-byte chunks fetched per plane
    static constexpr int NIT = (CH + 255) / 256;
    static constexpr int NTW = TW / 16;
    static constexpr int RING = NSLOT * PLANE;
    static constexpr int NSTEP = 9 * NB;
    static_assert(EH * RSB <= PLANE, "slot size");
};

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T> __device__ __forceinline__ T gload(const void* ptr) {
    return *reinterpret_cast<const T __attribute__((address_space(1)))*>(reinterpret_cast<unsigned long long>(ptr));
}
// uniform base + 32-bit byte offset per lane (one sample of any tensor here is < 4 GB: launcher)
template <typename T> __device__ __forceinline__ T gload_at(const void* base, unsigned off) {
    return *reinterpret_cast<const T __attribute__((address_space(1)))*>(reinterpret_cast<unsigned long long>(base) + off);
}
template <typename T> __device__ __forceinline__ void gstore_at(void* base, unsigned off, T v) {
    *reinterpret_cast<T __attribute__((address_space(1)))*>(reinterpret_cast<unsigned long long>(base) + off) = v;
}
__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (__bf16)0.f;
    return z;
}

struct PairP {
    int nseg_d, planes_per_seg;    // D segments per tile and output planes per segment (even)
    int tilesH, tilesW;
    int nwg_b;                     // workgroups per sample
    int edge_first;                // input gradients: face tiles first in dispatch order (needs tilesH, tilesW >= 3)
    unsigned* tickets;             // MODE 2: one zeroed counter per sample (left at zero): the workgroup that arrives last folds the sample's sums slots
};

// SF (MODE 2): the FINAL form of the input gradient.  The InstanceNorm-backward means are known beforehand (p.pwm: m1 = mean g,
// m2 = mean g x_hat per (sample, channel), from the weight gradient -- conv_wgrad_march.hip IND / march_ident_kernel), so the
// epilogue of an INTERIOR position writes dx = rstd (g - m1 - x_hat m2) straight into the dx tensor; only the halo outputs (the
// shell of the padded domain: positions -1 / S of an axis) still go to the padded scratch, and a small kernel folds that shell onto
// the positions 1 / S - 2 it mirrors (shell_fold_kernel, conv.hip).  No sums, no second pass over the volume.
template <int TW, int MODE, bool HAS_RES, bool SF = false, int NB = 1>
__global__ __launch_bounds__(256)
void conv_pair_kernel(ConvK p, PairP mp) {
    static_assert(!SF || MODE == 2, "final form: input gradient");
    static_assert(NB == 1 || MODE == 0, "16-channel K axis: forward only");
    using G = GeoP<TW, NB>;
    constexpr bool DG = (MODE == 2);
    constexpr int NIT = G::NIT, NTW = G::NTW, VS = G::VS, RSB = G::RSB, PLANE = G::PLANE, NSTEP = G::NSTEP;
    constexpr int NT = 2 * NTW, TT = DG ? NT + 2 : NT;        // main tiles per wave; MODE 2: + up to two halo tiles (edge workgroups)
    constexpr bool EOP = DG || HAS_RES;
    extern __shared__ __attribute__((aligned(16))) char lds[];            // [NSLOT planes]
    __shared__ float lsum[4 * 4 * 4 * 2 + 16];                // [wave][g][j][2] partial sums, then the workgroup's 8 x 2 totals

    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = (lane >> 4) & 3;
    const int half = g >> 1, cgl = (g & 1) * 4;               // this lane's accumulator: output plane `half` of the pair, channels cgl .. cgl + 3
    const int wv = rfl(tid >> 6);
    const int Sd = p.Di, Sh = p.Hi, Sw = p.Wi;
    const int rows = DG ? p.Cin : p.Cout;

    // ---- which tile / D segment
    const int bx = xcd_swizzle(blockIdx.x, gridDim.x);
    const int b = bx / mp.nwg_b;
    int r = bx % mp.nwg_b;
    int tw_i = r % mp.tilesW; r /= mp.tilesW;
    int th_i = r % mp.tilesH; r /= mp.tilesH;
    const int dseg = r;
    if (DG && mp.edge_first) face_tiles_first(mp.tilesH, mp.tilesW, th_i, tw_i);      // (conv_common.h)
    const int slot = (dseg * mp.tilesH + th_i) * mp.tilesW + tw_i;                    // sums / statistics slot: by tile, whatever the dispatch order
    const int h0 = th_i * MTH, w0 = tw_i * TW;
    // zero-padded convs (the fusion blocks' 3x3x3 conv): the adjoint has no halo outputs -- positions -1 and S of every axis are
    // neither computed nor counted; the finish kernel then reads the interior of the padded scratch only
    const bool nohalo = DG && p.pad_mode != PASSION_PAD_REFLECT;
    // output planes [dlo, dhi) in "position" coordinates: MODE 0: 0 .. Sd-1; MODE 2: -1 .. Sd (the padded domain); an even number of them
    const int dfirst = (DG && !nohalo) ? -1 : 0, dcount = (DG && !nohalo) ? Sd + 2 : Sd;
    const int dlo = dfirst + dseg * mp.planes_per_seg;
    const int dhi = min(dlo + mp.planes_per_seg, dfirst + dcount);
    const int L = dhi - dlo, NP = L >> 1;
    if (L <= 0) return;

    // ---- prologue: every staging thread resolves its source from the launch descriptor (scalar reads) and issues, back to back,
    // the first four planes, the weights, the mask byte and the InstanceNorm statistics
    const SegD& sd0 = p.seg[0];
    const int bs0 = sd0.bsrc(b);
    const int jb = tid % NB;                                   // 256 % NB == 0: a thread always stages the same 8-channel block
    const bf16_t* src;
    unsigned Cst = 8;                                          // channels per voxel of the staged block's tensor
    const uint8_t* mptr = nullptr;
    const float* sptr = nullptr;
    if (DG) {
        src = reinterpret_cast<const bf16_t*>(p.dy) + (long)b * Sd * Sh * Sw * 8;
    } else if (NB == 1) {
        src = reinterpret_cast<const bf16_t*>(sd0.x) + (long)bs0 * sd0.bstride;
        if (sd0.mask) mptr = sd0.mask + (long)b * sd0.mask_ld + sd0.mask_col;
        if (sd0.stats) sptr = sd0.stats + (long)bs0 * 8 * 2;
    } else {                                                   // block jb of the virtual concat: one 16-channel segment or two of 8
        src = reinterpret_cast<const bf16_t*>(sd0.x);
        int rr = jb;
        bool found = false;
#pragma unroll
        for (int s2 = 0; s2 < PASSION_MAX_SEG; ++s2) {
            if (s2 < p.nseg && !found) {
                const SegD& sd = p.seg[s2];
                const int nb = sd.C / 8;
                if (rr < nb) {
                    found = true;
                    const int bs = sd.bsrc(b), c0 = rr * 8;
                    src = reinterpret_cast<const bf16_t*>(sd.x) + (long)bs * sd.bstride + c0;
                    Cst = (unsigned)sd.C;
                    if (sd.mask) mptr = sd.mask + (long)b * sd.mask_ld + sd.mask_col;
                    if (sd.stats) sptr = sd.stats + ((long)bs * sd.C + c0) * 2;
                }
                rr -= nb;
            }
        }
    }
    const bool reflect = !DG && p.pad_mode == PASSION_PAD_REFLECT;
    unsigned srcel[NIT];                                       // in-plane source byte offset (clamped)
    int dstoff[NIT];                                           // LDS byte offset inside a plane slot
    unsigned keepbits = 0u;                                    // bit u: chunk u carries data (else it stays zero)
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
        const int c = u * 256 + tid, sv = min(c, G::CH - 1) / NB;
        const int zh = sv / (TW + 2), zw = sv % (TW + 2);
        int ih = h0 - 1 + zh, iw = w0 - 1 + zw;
        bool ok = c < G::CH;
        if (reflect) { ih = reflect_idx(ih, Sh); iw = reflect_idx(iw, Sw); }
        else ok = ok && ih >= 0 && ih < Sh && iw >= 0 && iw < Sw;
        ih = min(max(ih, 0), Sh - 1); iw = min(max(iw, 0), Sw - 1);
        srcel[u] = (unsigned)(ih * Sw + iw) * Cst * 2u;
        dstoff[u] = (zh + 1) * RSB + (zw + 1) * VS + jb * 16;
        if (ok) keepbits |= 1u << u;
    }
    const unsigned plane_b = (unsigned)(Sh * Sw) * Cst * 2u;
    // virtual input planes of this segment: v = dlo - 1 + j, j = 0 .. L + 1 (the pair loop also touches L + 2, L + 3: dead)
    auto plane_of = [&](int j, bool& live) {                  // source plane of virtual plane j; live = it holds data
        int v = dlo - 1 + j;
        live = j <= L + 1;
        if (reflect) v = reflect_idx(v, Sd);
        else live = live && v >= 0 && v < Sd;
        return min(max(v, 0), Sd - 1);
    };
    // every global load of the plane loop is UNCONDITIONAL (clamped address, data discarded by a select) and requested four planes
    // before its use, in a fixed program order (vmcnt counts in order)
    u32x4 raw[4][NIT];
    auto fetch = [&](int j, u32x4 (&dst)[NIT]) {
        bool live;
        const unsigned pbase = (unsigned)plane_of(j, live) * plane_b;
#pragma unroll
        for (int u = 0; u < NIT; ++u) dst[u] = gload_at<u32x4>(src, pbase + srcel[u]);
    };
#pragma unroll
    for (int j = 0; j < 4; ++j) fetch(j, raw[j]);

    // weights: the nine A fragments in registers.  Lane (row n, group g) of step s holds K slice q = 4 s + g = (pd, th, tw): rows 0..7
    // = output channel n of the pair's first plane (depth tap pd), rows 8..15 = channel n - 8 of the second (depth tap pd - 1).
    // MODE 2 is the gather form: offset (kd, th, tw) of the staged dY multiplies the packed tap (2 - kd, 2 - th, 2 - tw).
    bf16x8 wfr[NSTEP];
    int pre[NSTEP];                                            // LDS byte offset of the lane's B fragment of step s for tile 0, ring slot 0
    const int r0 = 2 * wv;
    const int lanebase = (r0 + 2) * RSB + (n + 2) * VS;        // main tile t: + (t / NTW) * RSB + (t % NTW) * 16 * VS (an immediate)
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        const int q = 4 * s + g, tq = q / NB, blk = q % NB, pd = tq / 9, r9 = tq % 9, th = r9 / 3, tw = r9 % 3;
        const int kd = pd - (n >> 3);
        const bool wok = kd >= 0 && kd <= 2 && (n & 7) < rows;
        const int kdc = min(max(kd, 0), 2);
        const int tap = DG ? (2 - kdc) * 9 + (2 - th) * 3 + (2 - tw) : kdc * 9 + th * 3 + tw;
        wfr[s] = gload<bf16x8>(reinterpret_cast<const bf16x8*>(p.wpack) + wvec_of(p, b) + ((long)tap * NB + blk) * p.Mpad + (n & 7));
        if (!wok) wfr[s] = zero8();
        pre[s] = pd * PLANE + (th - 1) * RSB + (tw - 1) * VS + blk * 16 + lanebase;
    }
    // mask byte + statistics of the staged block
    unsigned mbyte = 1u;
    if (mptr) mbyte = gload<uint8_t>(mptr);
    f32x4 st4[4];
    const bool has_norm = sptr != nullptr;
    if (has_norm) {
#pragma unroll
        for (int c = 0; c < 4; ++c) st4[c] = gload<f32x4>(sptr + 4 * c);       // (mean, rstd) of channels 2c, 2c + 1
    }
    // MODE 2 epilogue: back through LeakyReLU / the modality mask for the lane's four rows (= forward input channels cgl ..)
    const bool row_ok = cgl < rows;
    const bool e_st = DG && sd0.stats != nullptr;
    float emval = 1.f;
    f32x4 ersd = {0.f, 0.f, 0.f, 0.f}, eshf = {1.f, 1.f, 1.f, 1.f};      // x_hat = x * ersd + eshf (no statistics: 1, i.e. the positive branch)
    if (DG) {
        if (sd0.mask) emval = gload<uint8_t>(sd0.mask + (long)b * sd0.mask_ld + sd0.mask_col) ? 1.f : 0.f;
        if (e_st && row_ok) {
            const float* esp = sd0.stats + ((long)bs0 * sd0.C + cgl) * 2;
            const f32x4 a = gload<f32x4>(esp), c2 = gload<f32x4>(esp + 4);
            ersd = (f32x4){a[1], a[3], c2[1], c2[3]};
            eshf = (f32x4){-a[0] * a[1], -a[2] * a[3], -c2[0] * c2[1], -c2[2] * c2[3]};
        }
    }
    const float f_pos = row_ok ? emval : 0.f, f_neg = e_st ? f_pos * p.slope : f_pos;      // LeakyReLU' x modality mask
    f32x4 fm1 = {0.f, 0.f, 0.f, 0.f}, fm2 = {0.f, 0.f, 0.f, 0.f};             // SF: the means (m1, m2) of the lane's four channels
    if (SF && row_ok) {
        const float* mp2 = p.pwm + ((long)b * p.Cin + cgl) * 2;
        const f32x4 a = gload<f32x4>(mp2), c2 = gload<f32x4>(mp2 + 4);
        fm1 = (f32x4){a[0], a[2], c2[0], c2[2]};
        fm2 = (f32x4){a[1], a[3], c2[1], c2[3]};
    }
    // zero the whole ring once: the guard ring and the out-of-volume halo of zero-padded / gradient planes stay zero
    for (int i = tid; i < G::RING / 16; i += 256) reinterpret_cast<u32x4*>(lds)[i] = (u32x4){0u, 0u, 0u, 0u};

    // normalisation constants of the staged block: x_hat = x * rstd + shift
    f32x2 sc[4], sh[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        sc[c] = (f32x2){1.f, 1.f}; sh[c] = (f32x2){0.f, 0.f};
        if (has_norm) {
            sc[c] = (f32x2){st4[c][1], st4[c][3]};
            sh[c] = (f32x2){-st4[c][0] * st4[c][1], -st4[c][2] * st4[c][3]};
        }
    }
    if (mbyte == 0u) keepbits = 0u;                            // absent modality: the block stays zero
    const f32x2 slope2 = {p.slope, p.slope};
    auto commit = [&](int j, const u32x4 (&rv)[NIT]) {
        char* slot = lds + (j & (NSLOT - 1)) * PLANE;
        bool live;
        plane_of(j, live);
        const unsigned keep = live ? keepbits : 0u;            // zero padding is applied AFTER norm + activation: dead planes stay zero
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            u32x4 o = rv[u];
            if (has_norm) {                                    // workgroup-uniform
                bf16x8 ob;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f32x2 xv = {__uint_as_float(rv[u][c] << 16), __uint_as_float(rv[u][c] & 0xffff0000u)};
                    f32x2 t = __builtin_elementwise_fma(xv, sc[c], sh[c]);
                    const f32x2 ts = t * slope2;
                    t.x = fmaxf(t.x, ts.x);
                    t.y = fmaxf(t.y, ts.y);
                    ob[2 * c] = (__bf16)t.x; ob[2 * c + 1] = (__bf16)t.y;
                }
                o = __builtin_bit_cast(u32x4, ob);
            }
            if (!((keep >> u) & 1u)) o = (u32x4){0u, 0u, 0u, 0u};
            if (u * 256 + tid < G::CH) *reinterpret_cast<u32x4*>(slot + dstoff[u]) = o;
        }
    };

    // ---- output side: the column tiles of this lane.  t < NT: main tiles (row r0 + t / NTW, columns 16 (t % NTW) + n); MODE 2 adds up
    // to two halo tiles per wave (edge workgroups): the halo rows (position -1 / Sh) over the tile's columns and the halo columns
    // (position -1 / Sw) over rows h0 - 1 .. h0 + MTH, dealt over the four waves
    int tph[TT], tpw[TT];                                      // the lane's (h, w) position per tile
    int delta[2] = {0, 0};                                     // halo tiles: LDS offset of the lane's voxel relative to main tile 0's
    unsigned vmask = 0u;                                       // bit t: the lane's voxel of tile t exists
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        tph[t] = h0 + r0 + t / NTW; tpw[t] = w0 + (t % NTW) * 16 + n;
        vmask |= 1u << t;
    }
    int nx = 0;                                                // halo tiles of this wave (wave-uniform)
    if (DG) {
        const bool top = h0 == 0, bot = h0 + MTH == Sh, left = w0 == 0, right = w0 + TW == Sw;
#pragma unroll
        for (int xslot = 0; xslot < 2; ++xslot) {
            const int want = wv + 4 * xslot;                   // this wave's extra number `want` of the workgroup's list
            int e = 0, evb = 2 * RSB + 2 * VS, eph = h0, epw = w0;
            bool ev = false;
            if (top) {
#pragma unroll
                for (int tx = 0; tx < NTW; ++tx, ++e)
                    if (e == want) { evb = 1 * RSB + (tx * 16 + n + 2) * VS; eph = -1; epw = w0 + tx * 16 + n; ev = true; }
            }
            if (bot) {
#pragma unroll
                for (int tx = 0; tx < NTW; ++tx, ++e)
                    if (e == want) { evb = (MTH + 2) * RSB + (tx * 16 + n + 2) * VS; eph = Sh; epw = w0 + tx * 16 + n; ev = true; }
            }
            const bool cvalid = n <= MTH + 1 && (n >= 1 || top) && (n <= MTH || bot);      // lane n <-> row h0 - 1 + n
            const int crow = min(n, MTH + 1) + 1;
            if (left) {
                if (e == want) { evb = crow * RSB + 1 * VS; eph = h0 - 1 + min(n, MTH + 1); epw = -1; ev = cvalid; }
                ++e;
            }
            if (right) {
                if (e == want) { evb = crow * RSB + (TW + 2) * VS; eph = h0 - 1 + min(n, MTH + 1); epw = Sw; ev = cvalid; }
                ++e;
            }
            delta[xslot] = evb - lanebase; tph[NT + xslot] = eph; tpw[NT + xslot] = epw;
            if (ev && !nohalo) vmask |= 1u << (NT + xslot);
            if (__any((vmask >> (NT + xslot)) & 1u)) nx = xslot + 1;
        }
    }

    f32x4 biasv = {0.f, 0.f, 0.f, 0.f};
    if (!DG && bias_of(p, b) && row_ok) biasv = load4(bias_of(p, b) + cgl);
    const int Wp = pad8(Sw + 2);
    // epilogue operand of a tile, one 8-byte load per lane: MODE 0 the residual (HAS_RES), MODE 2 x (for x_hat, only where there are
    // statistics); output: MODE 2 the padded gradient scratch, MODE 0 y.  Uniform bases, byte offsets per lane.
    const char* ebase = DG ? reinterpret_cast<const char*>(reinterpret_cast<const bf16_t*>(sd0.x) + (e_st ? (long)bs0 * sd0.bstride : 0))
                           : reinterpret_cast<const char*>(reinterpret_cast<const bf16_t*>(p.residual) + (long)b * Sd * Sh * Sw * p.Cout);
    const unsigned eCb = DG ? (e_st ? (unsigned)sd0.C * 2u : 0u) : (unsigned)p.Cout * 2u;      // bytes per voxel of the operand (0: not needed)
    const unsigned elane = (row_ok && eCb) ? (unsigned)cgl * 2u : 0u;
    const unsigned eplane_b = (unsigned)(Sh * Sw) * eCb;
    char* ybase = DG ? reinterpret_cast<char*>(reinterpret_cast<bf16_t*>(p.y) + (long)b * (Sd + 2) * (Sh + 2) * Wp * p.Cin)
                     : reinterpret_cast<char*>(reinterpret_cast<bf16_t*>(p.y) + (long)b * Sd * Sh * Sw * p.Cout);
    unsigned eel[TT], oel[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        const int rh = min(max(reflect_idx(tph[t], Sh), 0), Sh - 1), rw = min(max(reflect_idx(tpw[t], Sw), 0), Sw - 1);
        eel[t] = (unsigned)(rh * Sw + rw) * eCb + elane;
        oel[t] = DG ? ((unsigned)((tph[t] + 1) * Wp + (tpw[t] + 1)) * (unsigned)p.Cin + (unsigned)cgl) * 2u
                    : ((unsigned)(tph[t] * Sw + tpw[t]) * (unsigned)p.Cout + (unsigned)cgl) * 2u;
    }
    const unsigned oplane_b = (DG ? (unsigned)((Sh + 2) * Wp) * (unsigned)p.Cin : (unsigned)(Sh * Sw) * (unsigned)p.Cout) * 2u;
    unsigned ob = (unsigned)(dlo + half + (DG ? 1 : 0)) * oplane_b;      // the lane's output plane of pair 0; + 2 planes per pair
    // SF: interior outputs go to dx itself (unpadded layout)
    char* fdx = SF ? reinterpret_cast<char*>(reinterpret_cast<bf16_t*>(sd0.dx) + (long)b * sd0.bstride) : nullptr;
    unsigned fel[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) fel[t] = ((unsigned)(tph[t] * Sw + tpw[t]) * (unsigned)sd0.C + (unsigned)cgl) * 2u;
    const unsigned fplane_b = (unsigned)(Sh * Sw) * (unsigned)sd0.C * 2u;
    unsigned fb = (unsigned)(dlo + half) * fplane_b;                     // (plane -1: wraps, never stored)
    const float vfx[2] = {((vmask >> NT) & 1u) ? 1.f : 0.f, ((vmask >> (NT + 1)) & 1u) ? 1.f : 0.f};
    u32x2 eop[2][EOP ? TT : 1];
    auto eload = [&](auto nxc, int o, u32x2 (&dst)[EOP ? TT : 1]) {
        if (!EOP) return;
        constexpr int TC = NT + decltype(nxc)::value;
        const int dpos = dlo + 2 * o + half;
        const unsigned pl = (unsigned)min(max(reflect_idx(dpos, Sd), 0), Sd - 1);
        const unsigned po = __umul24(pl, eplane_b);
#pragma unroll
        for (int t = 0; t < TC; ++t) dst[t] = gload_at<u32x2>(ebase, po + eel[t]);
    };
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2v[4] = {0.f, 0.f, 0.f, 0.f};
    const bool fstats = !DG && p.fst.stats != nullptr;

    // SF with HAS_RES (MODE 2): SegD.dx_add -- a contiguous tensor of dx's shape that is added into the final value (the gradient that
    // reached x through the identity branch of a residual block): one more 8-byte load per lane and tile, requested before the tap loop
    constexpr bool HAS_ADD = DG && SF && HAS_RES;
    const char* abase = HAS_ADD ? reinterpret_cast<const char*>(sd0.dx_add) + (long)b * Sd * Sh * Sw * sd0.C * 2 : nullptr;

    // output pair o (positions dlo + 2 o, + 1) from ring planes 2 o .. 2 o + 3
    auto compute = [&](auto nxc, int o, const u32x2 (&eo)[EOP ? TT : 1]) {
        constexpr int NX = decltype(nxc)::value, TC = NT + NX;
        const unsigned sbv = (unsigned)((2 * o) & (NSLOT - 1)) * (unsigned)PLANE;
        u32x2 addv[HAS_ADD ? NT : 1];
        if constexpr (HAS_ADD) {
            const unsigned pla = (unsigned)min(max(dlo + 2 * o + half, 0), Sd - 1) * fplane_b;
#pragma unroll
            for (int t = 0; t < NT; ++t) addv[t] = gload_at<u32x2>(abase, pla + fel[t]);
        }
        f32x4 acc[TC];
#pragma unroll
        for (int t = 0; t < TC; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (!(p.dbg & 2)) {
            bf16x8 cur[TC], nxt[TC];
            auto bload = [&](int s, bf16x8 (&dst)[TC]) {
                const char* base = lds + (((unsigned)pre[s] + sbv) & (unsigned)(G::RING - 1));
#pragma unroll
                for (int t = 0; t < NT; ++t) dst[t] = *reinterpret_cast<const bf16x8*>(base + (t / NTW) * RSB + (t % NTW) * 16 * VS);
#pragma unroll
                for (int x = 0; x < NX; ++x) dst[NT + x] = *reinterpret_cast<const bf16x8*>(base + delta[x]);
            };
            bload(0, cur);
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                if (s + 1 < NSTEP) bload(s + 1, nxt);
                __builtin_amdgcn_sched_barrier(0);             // keep the reads of step s + 1 ahead of the MFMAs of step s
#pragma unroll
                for (int t = 0; t < TC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[s], cur[t], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < TC; ++t) cur[t] = nxt[t];
            }
        }
        if (!(p.dbg & 4) && o < NP) {                          // (pairs past the segment: clamped garbage, nothing stored or counted)
#pragma unroll
            for (int t = 0; t < TC; ++t) {
                const unsigned off = ob + oel[t];
                if (!DG) {
                    f32x4 v = acc[t] + biasv;
                    if (HAS_RES) {
                        v[0] += __uint_as_float(eo[t].x << 16); v[1] += __uint_as_float(eo[t].x & 0xffff0000u);
                        v[2] += __uint_as_float(eo[t].y << 16); v[3] += __uint_as_float(eo[t].y & 0xffff0000u);
                    }
                    unsigned pk[2];
                    float rb[4];
                    round4_bf16(v, pk, rb);
                    if (row_ok) gstore_at<u32x2>(ybase, off, (u32x2){pk[0], pk[1]});
                    if (fstats) {                              // statistics of what was stored (workgroup-uniform branch)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            s1[j] += rb[j];
                            s2v[j] = __builtin_fmaf(rb[j], rb[j], s2v[j]);
                        }
                    }
                } else {
                    const u32x2 xr = eo[t];
                    const float xv[4] = {__uint_as_float(xr.x << 16), __uint_as_float(xr.x & 0xffff0000u),
                                         __uint_as_float(xr.y << 16), __uint_as_float(xr.y & 0xffff0000u)};
                    f32x4 gv, fv;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float xh = __builtin_fmaf(xv[j], ersd[j], eshf[j]);
                        const float gg = acc[t][j] * (xh > 0.f ? f_pos : f_neg);
                        gv[j] = gg;
                        if (SF) {                              // second half of the InstanceNorm backward, means known beforehand
                            fv[j] = e_st ? ersd[j] * __builtin_fmaf(-xh, fm2[j], gg - fm1[j]) : gg;
                        } else {
                            const float gs = t < NT ? gg : gg * vfx[t < NT ? 0 : t - NT];
                            s1[j] += gs;
                            s2v[j] = __builtin_fmaf(gs, xh, s2v[j]);
                        }
                    }
                    union { bf16_t h[4]; u32x2 u; } pk;
                    if (SF && t < NT) {
                        // interior plane: final value into dx; the padded planes -1 / Sd (the first / last pair of the volume) into the shell
                        const int dposl = dlo + 2 * o + half;
                        const bool pin = dposl >= 0 && dposl < Sd;
                        if constexpr (HAS_ADD) {
                            const u32x2 ar = addv[t < NT ? t : 0];
                            fv[0] += __uint_as_float(ar.x << 16); fv[1] += __uint_as_float(ar.x & 0xffff0000u);
                            fv[2] += __uint_as_float(ar.y << 16); fv[3] += __uint_as_float(ar.y & 0xffff0000u);
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) pk.h[j] = (bf16_t)fv[j];
                        if (row_ok && pin) gstore_at<u32x2>(fdx, fb + fel[t < NT ? t : 0], pk.u);
                        if (dlo + 2 * o < 0 || dlo + 2 * o + 1 >= Sd) {          // (uniform)
#pragma unroll
                            for (int j = 0; j < 4; ++j) pk.h[j] = (bf16_t)gv[j];
                            if (row_ok && !pin) gstore_at<u32x2>(ybase, off, pk.u);
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) pk.h[j] = (bf16_t)gv[j];
                        if (row_ok && ((vmask >> t) & 1u)) gstore_at<u32x2>(ybase, off, pk.u);
                    }
                }
            }
        }
        ob += 2u * oplane_b;
        if (SF) fb += 2u * fplane_b;
    };

    // ---- pipeline.  Iteration i: barrier; commit planes 2 i, 2 i + 1; fetch planes 2 i + 4, 2 i + 5; compute pair i - 2 (ring
    // planes 2 i - 4 .. 2 i - 1, all committed before this barrier; the two slots written now held planes 2 i - 8, 2 i - 7, last
    // read in iteration i - 2); fetch the epilogue operands of pair i.
    auto run = [&](auto nxc) {
        eload(nxc, 0, eop[0]);
        eload(nxc, 1, eop[1]);
        __syncthreads();                                       // ring zeroed before the first commit
#pragma unroll
        for (int i = 0; i < 2; ++i) {                          // planes 0 .. 3: nothing to compute yet
            if (i) __syncthreads();
            commit(2 * i, raw[2 * i]); commit(2 * i + 1, raw[2 * i + 1]);
            fetch(2 * i + 4, raw[2 * i]); fetch(2 * i + 5, raw[2 * i + 1]);
        }
        for (int i0 = 2; i0 < NP + 2; i0 += 2) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = i0 + ii;                         // planes 2 i, 2 i + 1 sit in raw[2 ii], raw[2 ii + 1]; pair i - 2 in operand slot ii
                __syncthreads();
                if (!(p.dbg & 1)) {                            // developer timing switches (PASSION_CONV_DBG): 1 no staging, 2 no MFMA loop, 4 no epilogue
                    commit(2 * i, raw[2 * ii]); commit(2 * i + 1, raw[2 * ii + 1]);
                    fetch(2 * i + 4, raw[2 * ii]); fetch(2 * i + 5, raw[2 * ii + 1]);
                }
                compute(nxc, i - 2, eop[ii]);
                eload(nxc, i, eop[ii]);
            }
        }
    };
    if (!DG || nx == 0) run(std::integral_constant<int, 0>{});
    else if (nx == 1) run(std::integral_constant<int, DG ? 1 : 0>{});
    else run(std::integral_constant<int, DG ? 2 : 0>{});

    if ((DG && !SF) || fstats) {
        // InstanceNorm-backward sums (MODE 2) / statistics of y (MODE 0): 16 voxel lanes (shuffles) -> 4 waves x 2 planes (LDS, fixed
        // order) -> this workgroup's own slot
        const float keep = (!DG || e_st) ? 1.f : 0.f;          // no statistics: nothing to sum
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
            const int c = tid >> 1, k2 = tid & 1, gq = c >> 2, j = c & 3;      // channel c = 4 gq + j: lane groups gq (first plane) and gq + 2
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) t += lsum[((w * 4 + gq) * 4 + j) * 2 + k2] + lsum[((w * 4 + gq + 2) * 4 + j) * 2 + k2];
            if (DG) {
                if (c < p.Cin) {
                    float* rs = p.sums + (long)slot * p.B * p.Cin * 2;
                    st_agent(&rs[((long)b * p.Cin + c) * 2 + k2], t);
                }
            } else {
                lsum[4 * 4 * 4 * 2 + tid] = t;
            }
        }
        if (!DG) {
            __syncthreads();
            fwd_stats_commit(p.fst, p.Cout, b, slot, 0, 8, lsum + 4 * 4 * 4 * 2, tid);
        } else if (mp.tickets) {
            // the sample's slots are folded HERE, by the workgroup that takes the last ticket (fixed slot order, in double: the sums
            // repeat bit for bit), into slot 0: the finish kernel reads one replica
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
}

template <int TW, int MODE, bool HAS_RES, bool SF = false, int NB = 1>
int launch_pair_res(ConvK& k, const PairP& mp, hipStream_t st) {
    using G = GeoP<TW, NB>;
    static bool granted = false;
    if (!granted) {
        HIP_RET(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pair_kernel<TW, MODE, HAS_RES, SF, NB>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, G::RING));
        granted = true;
    }
    if (NB > 1) passion_prof_tag("conv_pair_kernel<%d, %d, %d, 0, %d>", TW, MODE, HAS_RES ? 1 : 0, NB);
    else if (SF) passion_prof_tag("conv_pair_kernel<%d, %d, %d, 1>", TW, MODE, HAS_RES ? 1 : 0);
    else passion_prof_tag("conv_pair_kernel<%d, %d, %d>", TW, MODE, HAS_RES ? 1 : 0);
    hipLaunchKernelGGL((conv_pair_kernel<TW, MODE, HAS_RES, SF, NB>), dim3(k.B * mp.nwg_b), dim3(256), G::RING, st, k, mp);
    LAUNCH_RET();
    return 0;
}

template <int TW, int MODE, int NB = 1>
int launch_pair_one(ConvK& k, hipStream_t st) {
    PairP mp;
    mp.tickets = nullptr;
    mp.tilesH = k.Hi / MTH; mp.tilesW = k.Wi / TW;
    const bool nohalo = MODE == 2 && k.pad_mode != PASSION_PAD_REFLECT;
    const int dcount = (MODE == 2 && !nohalo) ? k.Di + 2 : k.Di;          // even (launch_conv_pair)
    const int tiles = mp.tilesH * mp.tilesW;
    // D segments: about g_conv_target workgroups per launch, at least 16 output planes per segment (two halo planes each), an even number
    int nseg = (int)hmax(1, (g_conv_target + (long)k.B * tiles - 1) / ((long)k.B * tiles));      // (final form 8 -> 8: 1024 workgroups 122 us, 2048 133)
    static const int min_planes = []() { const char* e = getenv("PASSION_MARCH_PLANES"); return e && atoi(e) > 0 ? atoi(e) : 16; }();
    nseg = (int)hmin(nseg, hmax(1, dcount / min_planes));
    if (MODE == 2 && !k.pwm) nseg = (int)hmin(nseg, hmax(1, SUMS_SLOTS / tiles));      // one sums slot per workgroup of a sample
    mp.planes_per_seg = ((dcount + nseg - 1) / nseg + 1) & ~1;
    mp.nseg_d = (dcount + mp.planes_per_seg - 1) / mp.planes_per_seg;
    mp.nwg_b = mp.nseg_d * tiles;
    static const bool edge_first = []() { const char* e = getenv("PASSION_MARCH_EDGEFIRST"); return !(e && e[0] == '0'); }();
    mp.edge_first = edge_first && mp.tilesH >= 3 && mp.tilesW >= 3;
    if (MODE == 2 && k.pwm) {                        // final form (means in k.pwm): reflect padding, statistics on the input
        if (TW != 16 || nohalo || !k.y) return 1;              // (a raw input -- no statistics -- takes the epilogue's `gg` branch: conv.hip raw_done)
        k.pw_done = 2;                               // 2: the shell of the padded scratch still has to be folded (shell_fold_kernel)
        if (k.seg[0].dx_add) return launch_pair_res<16, 2, true, true>(k, mp, st);      // ... + the identity branch's gradient (conv.hip: add_fused)
        return launch_pair_res<16, 2, false, true>(k, mp, st);
    }
    if (MODE == 2) {
        if (mp.nwg_b > SUMS_SLOTS) return 1;
        k.sums_rep = mp.nwg_b;
        static const bool fold_in = []() { const char* e = getenv("PASSION_MARCH_FOLD"); return !(e && e[0] == '0'); }();
        if (fold_in) {                               // the last workgroup of a sample folds its slots: one replica for the finish kernel
            void* zero = nullptr;
            const int rc = stream_scratch(st, (size_t)k.B * sizeof(unsigned), 0, &zero, nullptr);
            if (rc) return rc;
            mp.tickets = static_cast<unsigned*>(zero);
            k.sums_rep = 1;
        }
    }
    if (MODE == 0 && k.fst_want) {
        const int rc = fwd_stats_setup(k, mp.nwg_b, 1, st);
        if (rc) return rc < 0 ? rc : -rc;
    }
    if (MODE == 0 && k.residual) return launch_pair_res<TW, MODE, true, false, NB>(k, mp, st);
    return launch_pair_res<TW, MODE, false, false, NB>(k, mp, st);
}

}  // namespace

namespace pconv {

// 0 = launched, < 0 = HIP error, 1 = shape not covered (conv_march.hip goes on with its own kernel).  Called by launch_conv_march
// once conv_march_covers() holds with K8 = 1 and k.Mpad / k.wpack / k.dbg are set.
int launch_conv_pair(ConvK& k, int mode, hipStream_t st) {
    if (!g_conv_pair || k.Mpad != 16) return 1;
    const SegD& s0 = k.seg[0];
    const int rows = mode == 2 ? k.Cin : k.Cout;
    if (rows > 8 || rows % 4) return 1;
    int K8 = 0;
    for (int s = 0; s < k.nseg; ++s) { if (k.seg[s].C % 8) return 1; K8 += k.seg[s].C / 8; }
    if (mode == 0) { if (K8 * 8 != k.Cin || K8 > 2 || (K8 == 1 && k.nseg != 1)) return 1; }
    else if (k.nseg != 1 || k.Cout != 8 || !s0.dx || s0.cbase != 0 || s0.C != k.Cin) return 1;
    const bool nohalo = mode == 2 && k.pad_mode != PASSION_PAD_REFLECT;
    if (((mode == 2 && !nohalo) ? k.Di + 2 : k.Di) % 2) return 1;            // pairs of planes
    const long vox = (long)k.Di * k.Hi * k.Wi, pvox = (long)(k.Di + 2) * (k.Hi + 2) * pad8(k.Wi + 2);
    if (vox * 16 * 2 >= (1L << 32) || pvox * k.Cin * 2 >= (1L << 32)) return 1;      // 32-bit byte offsets inside one sample
    static const int force_tw = []() { const char* e = getenv("PASSION_PAIR_TW"); return e ? atoi(e) : 0; }();
    bool w32 = (k.Wi % 32 == 0) && mode == 0;
    if (force_tw == 16) w32 = false;
    if (force_tw == 32 && k.Wi % 32 == 0) w32 = true;
    static const bool nb2 = []() { const char* e = getenv("PASSION_PAIR_NB2"); return !(e && e[0] == '0'); }();
    if (mode == 0 && K8 == 2) return nb2 ? launch_pair_one<16, 0, 2>(k, st) : 1;      // 64 KB ring with 16-wide tiles (32-wide: 128 KB)
    if (w32) return mode == 2 ? launch_pair_one<32, 2>(k, st) : launch_pair_one<32, 0>(k, st);
    return mode == 2 ? launch_pair_one<16, 2>(k, st) : launch_pair_one<16, 0>(k, st);
}

}  // namespace pconv
