// Marching bf16 implicit-GEMM conv for the two finest levels (3x3x3, stride 1, <= 32 input and <= 16 output channels: the
// 128^3 / 64^3 layers that hold ~3/4 of the conv bytes of a step) -- forward (MODE 0) and the padded-domain dgrad (MODE 2) of
// general_conv3d_prenorm (reference code/models/blocks.py:300-316).  Same math, operand images, descriptors and epilogue
// semantics as conv_lean.hip; a different schedule, designed from what the counters said about that kernel: it is
// bound by instruction issue and by the latency of ONE brick of prefetch per workgroup, not by HBM.
//   * a workgroup owns one (8 x TW) tile of the (H, W) plane and MARCHES along D: every input plane is fetched once per tile
//     (halo overhead 10/8 x 34/32 = 1.33 instead of 6/4 x 10/8 x 10/8 = 2.34 per brick), the three planes a 3x3x3 tap needs
//     live in a 4-slot LDS ring, one barrier per plane;
//   * the tile is fixed for the workgroup's lifetime, so ALL index arithmetic (reflect / zero padding, halo, concat lookup, masks,
//     InstanceNorm statistics, LDS offsets) is done once: per plane a staging thread adds one plane offset to a constant;
//   * planes are fetched FOUR steps ahead into registers (4 x the bytes in flight of the one-brick prefetch) and normalised
//     (InstanceNorm + LeakyReLU, packed f32) on their way into LDS;
//   * weights of the whole K axis stay in registers as MFMA A fragments (<= 16 channels) -- the inner loop is one ds_read_b128
//     per v_mfma_f32_16x16x32_bf16;
//   * MODE 2 covers the reflect-padding adjoint without ragged tiles: tiles are aligned to the volume, the halo outputs
//     (position -1 and S of every axis) are extra column tiles of the edge workgroups, read through a zero guard ring around
//     the staged plane; the InstanceNorm-backward sums leave the workgroup as ONE plain store into its own slot (no atomics:
//     the fold in conv.hip is fixed-order, so the gradients are bitwise reproducible).
#include <cstdlib>
#include <type_traits>
#include "conv_common.h"
#include "prof.h"

namespace pconv { extern int g_conv_target, g_conv_dbg; int g_conv_march = 1; }
using namespace pconv;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int MTH = 8;                   // tile rows (two per wave)
constexpr int NSLOT = 4, ND = 4;         // LDS ring slots; planes in flight in registers

// FIN (the fold-free input gradient, below): three more zero columns -- voxel (row 1, column TW + 5) has an all-zero 3 x 3
// neighbourhood, the address a lane reads from when a mirror pass does not concern it -- and a fifth plane slot that stays zero
template <int NB, int TW, bool FIN = false> struct GeoM {
    static constexpr int EH = MTH + 4, EW = TW + 4 + (FIN ? 3 : 0);          // staged plane: tile + halo 1 + zero guard ring 1
    static constexpr int VS = 16 * NB;                        // bytes per voxel
    static constexpr int RSB = EW * VS;
    static constexpr int PLANE = EH * RSB;
    static constexpr int SV = (MTH + 2) * (TW + 2);           // voxels fetched per plane
    static constexpr int CH = SV * NB, NIT = (CH + 255) / 256;
    static constexpr int Q = 27 * NB, NSTEP = (Q + 3) / 4;
    static constexpr int NTW = TW / 16;                       // 16-voxel column tiles per row
    static constexpr int RING = (NSLOT + (FIN ? 1 : 0)) * PLANE;
    static constexpr int LDS = RING + (FIN ? 1024 : 0);       // FIN: + [16 channels][16 rows] f32 of the column mirror tile
    static constexpr int ZPATCH = 1 * RSB + (TW + 5) * VS;     // FIN: see above
};

struct alignas(16) MBlk {          // one 8-channel block of the K axis, resolved for this workgroup's sample
    const bf16_t* src;             // sample base + channel offset
    int C, on, has_norm, pad_;
    float shift[8], rstd[8];
};
struct alignas(16) MEpi {          // MODE 2 epilogue: the four output rows (= forward input channels) of lane group g
    const bf16_t* x;
    int C, ok, has_stats;
    float mval;
    float mean[4], rstd[4];
};

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Loads through pointers that reached the lane via LDS (the block descriptors) are FLAT loads unless told otherwise, and one
// pending flat_load makes the compiler wait with vmcnt(0) everywhere (flat returns out of order): every prefetch would be drained
// at its first use.  These are global pointers: say so.
template <typename T> __device__ __forceinline__ T gload(const void* ptr) {
    return *reinterpret_cast<const T __attribute__((address_space(1)))*>(reinterpret_cast<unsigned long long>(ptr));
}

__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (__bf16)0.f;
    return z;
}

struct MarchP {
    int nseg_d, planes_per_seg;    // D segments per tile and output planes per segment
    int tilesH, tilesW;
    int nwg_b;                     // workgroups per sample
    int edge_first;                // input gradients: face tiles first in dispatch order (needs tilesH, tilesW >= 3)
    int k8;                        // 8-channel blocks of the K axis that exist (<= NB: a 24-channel concat runs as NB = 4 with an empty block)
    unsigned* tickets;             // MODE 2: one zeroed counter per sample (left at zero): the workgroup that arrives last folds the sample's sums slots
};

// MT: 16-row tiles of the output (MODE 2 only: gradients with 17 .. 32 rows, e.g. the input gradient of a 24 / 32 -> 8 conv)
// FIN (MODE 2, MT 1): the FOLD-FREE input gradient with its final epilogue.  The adjoint of reflect padding adds the padded-domain
// values at positions -1 / S of an axis onto positions 1 / S - 2; instead of computing those halo outputs as tiles of their own,
// storing them and folding in a finish pass, the tap loop is run again for the target tile with the lane's base address moved to
// the mirror position (rows / columns: two staged voxels further out, where the zero guard ring supplies what lies outside; depth:
// only the taps whose plane exists, the others read a zero plane) and accumulates into the SAME accumulator.  With the
// InstanceNorm-backward means known beforehand (p.pwm: from the weight gradient, conv_pw.hip / march_ident_kernel) the epilogue
// writes dx = rstd (g - m1 - x_hat m2) itself: no scratch volume, no sums, no finish kernel.
template <int NB, int TW, int MODE, bool HAS_RES, int MT = 1, bool FIN = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(((MT == 2 && NB == 1) || FIN) ? 2 : 1)))      // 264 registers uncapped: one wave per SIMD
void conv_march_kernel(ConvK p, MarchP mp) {
    static_assert(!FIN || (MODE == 2 && MT == 1), "final form: input gradient, one row tile");
    using G = GeoM<NB, TW, FIN>;
    constexpr bool DG = (MODE == 2);
    // B fragments are requested PD K-steps ahead of the MFMAs that consume them.  Measured (tools/march_pd.sh, 8 -> 8 / 16 -> 8 at
    // 4 x 128^3, 16 -> 16 at 4 x 64^3): PD 1 .. 4 give the same times to 1 % wherever the extra registers keep the waves per SIMD --
    // LDS latency is not what the tap loop waits for (the loop is bound by VALU issue: ~210 VALU instructions per plane and wave
    // against 21 MFMAs) -- and 40 % slower where they do not (16 -> 8 forward at PD >= 2: 260 registers).
#ifndef MARCH_PD
#define MARCH_PD 1
#endif
    constexpr int PD = FIN ? 1 : MARCH_PD;
    constexpr int NIT = G::NIT, NSTEP = G::NSTEP, NTW = G::NTW, VS = G::VS, RSB = G::RSB, PLANE = G::PLANE;
    extern __shared__ __attribute__((aligned(16))) char lds[];            // [NSLOT planes]
    __shared__ float lsum[4 * MT * 16 * 2 + MT * 32];     // [wave][row][2] partial sums, then the workgroup's 16 MT x 2 totals

    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = (lane >> 4) & 3;
    const int wv = rfl(tid >> 6);
    const int Sd = p.Di, Sh = p.Hi, Sw = p.Wi;
    const int K8 = NB;

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
    // output planes [dlo, dhi) in "position" coordinates: MODE 0: 0 .. Sd-1; MODE 2: -1 .. Sd (the padded domain)
    const int dfirst = (DG && !FIN) ? -1 : 0, dcount = (DG && !FIN) ? Sd + 2 : Sd;
    const int dlo = dfirst + dseg * mp.planes_per_seg;
    const int dhi = min(dlo + mp.planes_per_seg, dfirst + dcount);
    const int L = dhi - dlo;
    if (L <= 0) return;

    // ---- prologue.  Nothing goes through LDS or waits on another thread: every staging thread resolves ITS 8-channel block of
    // the K axis from the launch descriptor (scalar reads) and issues, back to back, the first ND planes, the weights, the mask
    // byte and the InstanceNorm statistics -- one memory latency instead of a chain of three (descriptor -> barrier -> planes).
    const int jb = tid % NB;                                   // 256 % NB == 0: a thread always stages the same block
    const bf16_t* src;
    int Cst;
    const uint8_t* mptr = nullptr;                             // presence byte of the block's modality for this sample
    const float* sptr = nullptr;                               // (mean, rstd) x 8 channels
    bool found = true;                                         // MODE 0: the staged block exists (an NB = 4 launch over three blocks stages zeros for the fourth)
    if (DG) {
        src = reinterpret_cast<const bf16_t*>(p.dy) + (long)b * Sd * Sh * Sw * p.Cout + jb * 8;
        Cst = p.Cout;
    } else {
        src = reinterpret_cast<const bf16_t*>(p.seg[0].x); Cst = 8;
        int rr = jb;
        found = false;
#pragma unroll
        for (int s2 = 0; s2 < PASSION_MAX_SEG; ++s2) {
            if (s2 < p.nseg && !found) {
                const SegD& sd = p.seg[s2];
                const int nb = sd.C / 8;
                if (rr < nb) {
                    found = true;
                    const int bs = sd.bsrc(b), c0 = rr * 8;
                    src = reinterpret_cast<const bf16_t*>(sd.x) + (long)bs * sd.bstride + c0;
                    Cst = sd.C;
                    if (sd.mask) mptr = sd.mask + (long)b * sd.mask_ld + sd.mask_col;
                    if (sd.stats) sptr = sd.stats + ((long)bs * sd.C + c0) * 2;
                }
                rr -= nb;
            }
        }
    }
    if (NB == 1) Cst = rfl(Cst);
    const bool reflect = !DG && p.pad_mode == PASSION_PAD_REFLECT;
    unsigned srcel[NIT];                                       // in-plane source element offset (clamped)
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
        srcel[u] = (unsigned)(ih * Sw + iw) * (unsigned)Cst;
        dstoff[u] = (zh + 1) * RSB + (zw + 1) * VS + jb * 16;
        if (ok) keepbits |= 1u << u;
    }
    const unsigned plane_el = (unsigned)(Sh * Sw) * (unsigned)Cst;
    // virtual input planes of this segment: v = dlo - 1 + j, j = 0 .. L + 1
    auto plane_of = [&](int j, bool& live) {                  // source plane of virtual plane j; live = it holds data
        int v = dlo - 1 + j;
        live = j <= L + 1;
        if (reflect) v = reflect_idx(v, Sd);
        else live = live && v >= 0 && v < Sd;
        return min(max(v, 0), Sd - 1);
    };
    // Every global load of the plane loop is UNCONDITIONAL (clamped address, data discarded by a select) and requested ND
    // iterations before its use, in a fixed program order: vmcnt counts in order, so a single wait on a recent load (or one load
    // inside a branch, which makes the compiler fall back to vmcnt(0)) would drain the whole prefetch queue.
    u32x4 raw[ND][NIT];
    auto fetch = [&](int j, u32x4 (&dst)[NIT]) {
        bool live;
        const unsigned pbase = (unsigned)plane_of(j, live) * plane_el;
#pragma unroll
        for (int u = 0; u < NIT; ++u) dst[u] = gload<u32x4>(src + (pbase + srcel[u]));
    };
#pragma unroll
    for (int j = 0; j < ND; ++j) fetch(j, raw[j]);

    // weights: all A fragments of the K axis in registers.  pair q = 4 s + g -> (tap, block)
    bf16x8 wfr[MT][NSTEP];
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        const int q = min(4 * s + g, G::Q - 1), tap = q / NB, blk = q % NB;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            wfr[mt][s] = gload<bf16x8>(reinterpret_cast<const bf16x8*>(p.wpack) + wvec_of(p, b) + ((long)tap * mp.k8 + min(blk, mp.k8 - 1)) * p.Mpad + mt * 16 + n);
            if (4 * s + g >= G::Q || blk >= mp.k8) wfr[mt][s] = zero8();
        }
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
    // MODE 2 epilogue: the four output rows (= forward input channels 4 g .. 4 g + 3) of this lane
    const bf16_t* ex[MT];
    int eC[MT];
    bool e_ok[MT], e_st[MT];
    float emval[MT];
    f32x4 emean[MT], erstd[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        ex[mt] = nullptr; eC[mt] = 4; e_ok[mt] = false; e_st[mt] = false; emval[mt] = 1.f;
        emean[mt] = (f32x4){0.f, 0.f, 0.f, 0.f}; erstd[mt] = (f32x4){1.f, 1.f, 1.f, 1.f};
        if (DG) {
            const int cg = mt * 16 + g * 4;
            const uint8_t* emp = nullptr;
            const float* esp = nullptr;
#pragma unroll
            for (int s2 = 0; s2 < PASSION_MAX_SEG; ++s2) {
                if (s2 < p.nseg && cg < p.Cin) {
                    const SegD& sd = p.seg[s2];
                    if (cg >= sd.cbase && cg < sd.cbase + sd.C && sd.dx != nullptr) {
                        const int cl = cg - sd.cbase, bs = sd.bsrc(b);
                        e_ok[mt] = true; eC[mt] = sd.C; e_st[mt] = sd.stats != nullptr;
                        ex[mt] = reinterpret_cast<const bf16_t*>(sd.x) + (long)bs * sd.bstride + cl;
                        if (sd.mask) emp = sd.mask + (long)b * sd.mask_ld + sd.mask_col;
                        if (sd.stats) esp = sd.stats + ((long)bs * sd.C + cl) * 2;
                    }
                }
            }
            if (emp) emval[mt] = gload<uint8_t>(emp) ? 1.f : 0.f;
            if (esp) {
                const f32x4 a = gload<f32x4>(esp), c2 = gload<f32x4>(esp + 4);
                emean[mt] = (f32x4){a[0], a[2], c2[0], c2[2]};
                erstd[mt] = (f32x4){a[1], a[3], c2[1], c2[3]};
            }
        }
    }
    // FIN: where the lane's four rows go (the segment's dx, channels 4 g ..) and their means (m1, m2) / V
    bf16_t* fdx = nullptr;
    f32x4 fm1 = {0.f, 0.f, 0.f, 0.f}, fm2 = {0.f, 0.f, 0.f, 0.f};
    if (FIN) {
        const int cg = g * 4;
#pragma unroll
        for (int s2 = 0; s2 < PASSION_MAX_SEG; ++s2) {
            if (s2 < p.nseg && cg < p.Cin) {
                const SegD& sd = p.seg[s2];
                if (cg >= sd.cbase && cg < sd.cbase + sd.C && sd.dx != nullptr)
                    fdx = reinterpret_cast<bf16_t*>(sd.dx) + (long)b * sd.bstride + (cg - sd.cbase);
            }
        }
        if (cg < p.Cin) {
            const f32x4 a = gload<f32x4>(p.pwm + ((long)b * p.Cin + cg) * 2), c2 = gload<f32x4>(p.pwm + ((long)b * p.Cin + cg) * 2 + 4);
            fm1 = (f32x4){a[0], a[2], c2[0], c2[2]};
            fm2 = (f32x4){a[1], a[3], c2[1], c2[3]};
        }
    }
    // zero the whole ring once: the guard ring and the out-of-volume halo of zero-padded / gradient planes stay zero
    for (int i = tid; i < G::RING / 16; i += 256) reinterpret_cast<u32x4*>(lds)[i] = (u32x4){0u, 0u, 0u, 0u};

    // per-lane LDS offset of step s relative to the lane's output voxel and the plane of its depth tap
    int toff[NSTEP], tdsel[NSTEP];
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        const int q = 4 * s + g, tap = (q < G::Q) ? q / NB : 0, blk = q % NB;
        int td = tap / 9, th = (tap / 3) % 3, tw = tap % 3;
        if (DG) { td = 2 - td; th = 2 - th; tw = 2 - tw; }       // gather form: position i reads dY[i + 1 - t]
        tdsel[s] = td;
        toff[s] = (th - 1) * RSB + (tw - 1) * VS + blk * 16;
    }
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
    if (mbyte == 0u || !found) keepbits = 0u;                  // absent modality / block past the concat: the block stays zero
    const f32x2 slope2 = {p.slope, p.slope};
    auto commit = [&](int j, const u32x4 (&rv)[NIT]) {
        char* slot = lds + (j & (NSLOT - 1)) * PLANE;
        bool live;
        plane_of(j, live);
        const unsigned keep = live ? keepbits : 0u;            // zero padding is applied AFTER norm + activation: dead planes stay zero
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            u32x4 o = rv[u];
            if (has_norm) {                                    // workgroup-uniform for NB == 1, per staging thread otherwise
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

    // ---- output side: the column tiles of this lane.  t < NT: main tiles (row r0 + t / NTW, columns 16 (t % NTW) + n);
    // MODE 2 adds up to two halo tiles per wave (edge workgroups): the halo rows (position -1 / Sh) over the tile's columns and
    // the halo columns (position -1 / Sw) over rows h0 - 1 .. h0 + MTH, dealt over the four waves
    constexpr int NT = 2 * NTW, TT = (DG ? NT + 2 : NT);      // MODE 2: + two halo tiles (FIN: two mirror tiles, below)
    constexpr int ET = FIN ? NT : TT;                          // tiles with an epilogue
    const int r0 = 2 * wv;
    int vb[TT], tph[TT], tpw[TT];                              // LDS offset of the lane's voxel in a plane slot; its (h, w) position
    unsigned vmask = 0u;                                       // bit t: the lane's voxel of tile t exists
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int rr = t / NTW, tx = t % NTW;
        vb[t] = (r0 + rr + 2) * RSB + (tx * 16 + n + 2) * VS;
        tph[t] = h0 + r0 + rr; tpw[t] = w0 + tx * 16 + n;
        vmask |= 1u << t;
    }
    // zero-padded convs (the fusion blocks' 3x3x3 conv): the adjoint has no halo outputs -- positions -1 and S of every axis
    // are neither computed nor counted; the finish kernel then reads the interior of the padded scratch only (fold = 0)
    const bool nohalo = DG && p.pad_mode != PASSION_PAD_REFLECT;
    if (DG && !FIN) {
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
            vb[NT + xslot] = evb; tph[NT + xslot] = eph; tpw[NT + xslot] = epw;
            if (ev && !nohalo) vmask |= 1u << (NT + xslot);
        }
    }
    // which tiles this wave computes at all (wave-uniform): the extras only where some lane has one
    bool tile_on[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) tile_on[t] = t < NT || __any((vmask >> t) & 1u);
    // FIN (16-wide tiles: tile 0 = row r0, tile 1 = row r0 + 1): the mirror tiles.  Each is the tap loop read from mirror positions and
    // joins its target accumulators after the loop; a wave carries at most one.
    //   X0, rows (waves 0 / 3 of the top / bottom workgroups): the voxels of row 1 / Sh - 2 two staged rows further out (padded row
    //   -1 / Sh), lane for lane with the target tile;
    //   X1, columns (wave 1 of the left / right workgroups): lane n = row h0 - 1 + n of the padded column -1 / Sw (rows -1 / Sh = the
    //   corners, only in top / bottom workgroups); its 16 x 16 values cross to the waves that own the rows through 1 KB of LDS
    //   (one more barrier per plane, in these workgroups only), lane 1 / 14 of their tiles takes them.
    bool xh = false, xw = false, wtop = false, wbot = false;
    int xh_tgt = 0, wlane = 1;
    if (FIN) {
        static_assert(!FIN || NTW == 1, "final form: 16-wide tiles");
        const bool top = h0 == 0, bot = h0 + MTH == Sh, left = w0 == 0, right = w0 + TW == Sw;
        const bool hlo = top && wv == 0, hhi = bot && wv == 3;       // this wave holds row 1 (its second row) / row Sh - 2 (its first)
        xh = hlo || hhi; xw = left || right;                         // (Sh >= 16, Sw >= 32: one edge per axis at most)
        wtop = top && xw; wbot = bot && xw;
        xh_tgt = hlo ? 1 : 0;
        wlane = left ? 1 : 14;
        const int colv = (left ? 1 : TW + 2) * VS;                   // staged column of padded column -1 / Sw
        const bool cval = n <= MTH + 1 && (n >= 1 || top) && (n <= MTH || bot);
        vb[NT + 0] = hlo ? vb[1] - 2 * RSB : vb[0] + 2 * RSB;
        vb[NT + 1] = cval ? (n + 1) * RSB + colv : G::ZPATCH;
        tile_on[NT + 0] = xh; tile_on[NT + 1] = xw && wv == 1;
        tph[NT] = tph[NT + 1] = h0; tpw[NT] = tpw[NT + 1] = w0;
    }

    f32x4 biasv = {0.f, 0.f, 0.f, 0.f};
    if (!DG && bias_of(p, b) && g * 4 < p.Cout) biasv = load4(bias_of(p, b) + g * 4);
    const bool row_ok = g * 4 < (DG ? p.Cin : p.Cout);
    float f_pos[MT], f_neg[MT];                                // LeakyReLU' x modality mask
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { f_pos[mt] = emval[mt]; f_neg[mt] = e_st[mt] ? emval[mt] * p.slope : emval[mt]; }
    const int Wp = pad8(Sw + 2);
    bf16_t* ybase = FIN ? fdx
                  : DG ? reinterpret_cast<bf16_t*>(p.y) + (long)b * (Sd + 2) * (Sh + 2) * Wp * p.Cin
                       : reinterpret_cast<bf16_t*>(p.y) + (long)b * Sd * Sh * Sw * p.Cout;
    // epilogue operand of a tile, one 8-byte load per lane: MODE 0 the residual (HAS_RES), MODE 2 x (for x_hat); in-plane element
    // offset per tile (clamped / reflected to a valid voxel), plane offset added per step
    constexpr bool EOP = DG || HAS_RES;
    // lanes whose four rows do not exist (g * 4 >= channels) load from element 0 of a valid tensor: their x / residual is never used,
    // and "their" channels of the last voxel would lie past the end of the allocation
    const bf16_t* ebase[MT];
    int eCs[MT];
    unsigned eplane[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        ebase[mt] = DG ? (e_ok[mt] ? ex[mt] : reinterpret_cast<const bf16_t*>(p.seg[0].x))
                       : reinterpret_cast<const bf16_t*>(p.residual) + (long)b * Sd * Sh * Sw * p.Cout + (row_ok ? g * 4 : 0);
        eCs[mt] = DG ? ((e_ok[mt] && e_st[mt]) ? eC[mt] : 0) : (row_ok ? p.Cout : 0);      // x only where it is needed (x_hat for LeakyReLU' and the sums)
        eplane[mt] = (unsigned)(Sh * Sw) * (unsigned)eCs[mt];
    }
    unsigned eel[MT][ET], oel[ET];                             // in-plane element offsets: operand / output (row tile mt: + 16 mt channels)
#pragma unroll
    for (int t = 0; t < ET; ++t) {
        const int rh = min(max(reflect_idx(tph[t], Sh), 0), Sh - 1), rw = min(max(reflect_idx(tpw[t], Sw), 0), Sw - 1);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) eel[mt][t] = (unsigned)(rh * Sw + rw) * (unsigned)eCs[mt];
        oel[t] = FIN ? (unsigned)(tph[t] * Sw + tpw[t]) * (unsigned)eC[0]
               : DG ? (unsigned)((tph[t] + 1) * Wp + (tpw[t] + 1)) * (unsigned)p.Cin + (unsigned)(g * 4)
                    : (unsigned)(tph[t] * Sw + tpw[t]) * (unsigned)p.Cout + (unsigned)(g * 4);
    }
    const unsigned oplane = FIN ? (unsigned)(Sh * Sw) * (unsigned)eC[0]
                          : DG ? (unsigned)((Sh + 2) * Wp) * (unsigned)p.Cin : (unsigned)(Sh * Sw) * (unsigned)p.Cout;
    u32x2 eop[ND][EOP ? MT * ET : 1];
    auto eload = [&](int jout, u32x2 (&dst)[EOP ? MT * ET : 1]) {
        if (!EOP) return;
        const int dpos = dlo + jout;
        const unsigned pl = (unsigned)min(max(reflect_idx(dpos, Sd), 0), Sd - 1);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < ET; ++t) dst[mt * ET + t] = gload<u32x2>(ebase[mt] + (pl * eplane[mt] + eel[mt][t]));
    };
    float s1[MT][4], s2v[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) { s1[mt][j] = 0.f; s2v[mt][j] = 0.f; }
    const bool fstats = !DG && p.fst.stats != nullptr;

    // output plane jout (position dlo + jout) from ring planes jout, jout + 1, jout + 2
    const int lanebase = (r0 + 2) * RSB + (n + 2) * VS;        // main tile t: + (t / NTW) * RSB + (t % NTW) * 16 * VS (an immediate)
    char* yb8 = reinterpret_cast<char*>(ybase);
    auto compute = [&](auto edgec, int jout, const u32x2 (&eo)[EOP ? MT * ET : 1]) {
        constexpr int EDGE = decltype(edgec)::value;             // FIN: bit 0 = this wave may hold the row mirror tile, bit 1 = the column tile
        const int dpos = dlo + jout;
        const bool pvalid = jout >= 0 && jout < L && !(nohalo && (dpos < 0 || dpos >= Sd));
        f32x4 acc[MT][TT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < TT; ++t) acc[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // B fragments one K step ahead of the MFMAs that consume them (the compiler alternates read / wait / MFMA otherwise)
        // dm (FIN): 0 the planes of this output; 1 / 2: the padded plane -1 / Sd seen from output 1 / Sd - 2 -- only the taps that
        // read depth 0 / Sd - 1 have a plane (ring plane jout / jout + 2), the others read the zero slot
        // The tap loop over the tiles of bit mask M (compile time), B fragments one K step ahead.  RT: the run-time depth-mirror form
        // (FIN) -- dm 1 / 2 = the padded plane -1 / Sd seen from output 1 / Sd - 2: only the taps that read depth 0 / Sd - 1 have a
        // plane (ring plane jout / jout + 2), the others read the zero slot; the mirror tiles are switched by tile_on there.
        auto mfma_loop = [&](auto maskc, auto rtc, int dm) {
            constexpr unsigned M = decltype(maskc)::value;
            constexpr bool RT = decltype(rtc)::value;
            bf16x8 bq[NSTEP][TT];                                  // (fully unrolled: only PD + 1 steps' worth is ever live)
            auto bload = [&](int s, bf16x8 (&dst)[TT]) {
                int pl = ((jout + tdsel[s]) & (NSLOT - 1)) * PLANE;
                if (RT) {
                    const int want = dm == 1 ? 2 : 0, from = dm == 1 ? jout : jout + 2;
                    pl = (tdsel[s] == want) ? (from & (NSLOT - 1)) * PLANE : NSLOT * PLANE;
                }
                const int soff = toff[s] + pl;
                const char* base = lds + lanebase + soff;
#pragma unroll
                for (int t = 0; t < TT; ++t) {
                    if (!((M >> t) & 1u)) continue;
                    if (t < NT) dst[t] = *reinterpret_cast<const bf16x8*>(base + (t / NTW) * RSB + (t % NTW) * 16 * VS);
                    else if ((FIN && !RT) || tile_on[t]) dst[t] = *reinterpret_cast<const bf16x8*>(lds + vb[t] + soff);
                }
            };
#pragma unroll
            for (int s = 0; s < PD && s < NSTEP; ++s) bload(s, bq[s]);
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                if (s + PD < NSTEP) bload(s + PD, bq[s + PD]);
                __builtin_amdgcn_sched_barrier(0);                 // keep the reads of step s + PD ahead of the MFMAs of step s
#pragma unroll
                for (int t = 0; t < TT; ++t)
                    if (((M >> t) & 1u) && (t < NT || (FIN && !RT) || tile_on[t])) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[mt][s], bq[s][t], acc[mt][t], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        constexpr unsigned MALL = (1u << TT) - 1u, MMAIN = (1u << NT) - 1u;
        if (!(p.dbg & 2)) {
            if constexpr (FIN) {
                // one loop with the wave's tiles in flight together (at most three).  EDGE = the workgroup's edge class (bit 0 top /
                // bottom, bit 1 left / right): each class runs its own copy of the plane loop.  (Measured on 4 x 128^3, 8 -> 8: mirror
                // tiles as loops of their own = chains of dependent LDS reads and MFMAs, 194 us; both mirror tiles in every wave (zero
                // patch where there is none) = the LDS read path saturates, 187 us; the column tile in all four waves of an edge
                // workgroup = those workgroups run 45 % longer and set the kernel's duration, 183 us.)
                constexpr unsigned M0 = MMAIN | (1u << NT), M1 = MMAIN | (2u << NT);
                if ((EDGE & 1) && xh) mfma_loop(std::integral_constant<unsigned, M0>{}, std::false_type{}, 0);
                else if ((EDGE & 2) && wv == 1) mfma_loop(std::integral_constant<unsigned, M1>{}, std::false_type{}, 0);
                else mfma_loop(std::integral_constant<unsigned, MMAIN>{}, std::false_type{}, 0);
                const bool m1p = dpos == 1, m2p = dpos == Sd - 2;      // two planes of a volume
                if (m1p || m2p) {
#pragma unroll 1
                    for (int ps = (m1p ? 1 : 2); ps <= (m2p ? 2 : 1); ++ps) mfma_loop(std::integral_constant<unsigned, MALL>{}, std::true_type{}, ps);
                }
                if ((EDGE & 1) && xh) { if (xh_tgt) acc[0][1] += acc[0][NT]; else acc[0][0] += acc[0][NT]; }
                if constexpr ((EDGE & 2) != 0) {
                    float* xbuf = reinterpret_cast<float*>(lds + G::RING);
                    if (wv == 1) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) xbuf[(g * 4 + j) * 16 + n] = acc[0][NT + 1][j];
                    }
                    __syncthreads();                                   // (the barrier at the top of the next plane separates this read from the next write)
                    const float here = n == wlane ? 1.f : 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float* xr = xbuf + (g * 4 + j) * 16;
                        float a0 = xr[r0 + 1], a1 = xr[r0 + 2];
                        if (wtop && wv == 0) a1 += xr[0];              // the corners: padded row -1 onto row 1, row Sh onto row Sh - 2
                        if (wbot && wv == 3) a0 += xr[MTH + 1];
                        acc[0][0][j] += here * a0;
                        acc[0][1][j] += here * a1;
                    }
                }
            } else {
                mfma_loop(std::integral_constant<unsigned, MALL>{}, std::false_type{}, 0);
            }
        }
        if (p.dbg & 4) return;
        const unsigned ob = (unsigned)((DG && !FIN) ? dpos + 1 : dpos) * oplane;
#pragma unroll
        for (int t = 0; t < ET; ++t) {
            if (!tile_on[t]) continue;
            const bool valid = pvalid && ((vmask >> t) & 1u);
            const unsigned obyte = (ob + oel[t]) * 2u;         // < 2^32: one sample of the output / scratch volume
            if (!DG) {
                f32x4 v = acc[0][t] + biasv;
                if (HAS_RES) {
                    v[0] += __uint_as_float(eo[t].x << 16); v[1] += __uint_as_float(eo[t].x & 0xffff0000u);
                    v[2] += __uint_as_float(eo[t].y << 16); v[3] += __uint_as_float(eo[t].y & 0xffff0000u);
                }
                unsigned pk[2];
                float rb[4];
                round4_bf16(v, pk, rb);
                if (valid && row_ok) *reinterpret_cast<u32x2*>(yb8 + obyte) = (u32x2){pk[0], pk[1]};
                if (fstats) {                                  // statistics of what was stored (workgroup-uniform branch)
                    const float vf = valid ? 1.f : 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float r = rb[j] * vf;
                        s1[0][j] += r;
                        s2v[0][j] = __builtin_fmaf(r, r, s2v[0][j]);
                    }
                }
            } else {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const u32x2 xr = eo[mt * ET + t];
                    const float xv[4] = {__uint_as_float(xr.x << 16), __uint_as_float(xr.x & 0xffff0000u),
                                         __uint_as_float(xr.y << 16), __uint_as_float(xr.y & 0xffff0000u)};
                    const float vf = (valid && e_ok[mt] && e_st[mt]) ? 1.f : 0.f;
                    f32x4 gv = acc[mt][t];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float xh = e_st[mt] ? (xv[j] - emean[mt][j]) * erstd[mt][j] : 1.f;
                        const float gg = gv[j] * (xh > 0.f ? f_pos[mt] : f_neg[mt]);
                        if (FIN) {                             // second half of the InstanceNorm backward, means known beforehand
                            gv[j] = e_st[mt] ? erstd[mt][j] * (gg - fm1[j] - xh * fm2[j]) : gg;
                        } else {
                            gv[j] = gg;
                            const float gs = gg * vf;
                            s1[mt][j] += gs;
                            s2v[mt][j] = __builtin_fmaf(gs, xh, s2v[mt][j]);
                        }
                    }
                    if (valid && e_ok[mt]) store4(reinterpret_cast<bf16_t*>(yb8 + obyte + mt * 32), gv);
                }
            }
        }
    };

    // ---- pipeline.  iteration j: barrier; commit plane j; fetch plane j + ND; compute output j - 3 (ring planes j-3 .. j-1,
    // all committed before this barrier; slot j & 3 = the slot of plane j - 4, last read in iteration j - 1); fetch the
    // epilogue operands of output j - 3 + ND.  Dead iterations past the segment compute / load clamped garbage and store nothing.
#pragma unroll
    for (int j = 0; j < ND; ++j) eload(j, eop[j]);
    __syncthreads();                                           // ring zeroed before the first commit
#pragma unroll
    for (int j = 0; j < 3; ++j) {                              // planes 0 .. 2: nothing to compute yet
        __syncthreads();
        commit(j, raw[j]);
        fetch(j + ND, raw[j]);
    }
    auto run = [&](auto edgec) {
        for (int j0 = 3; j0 < L + 3; j0 += ND) {
#pragma unroll
            for (int jj = 0; jj < ND; ++jj) {
                const int j = j0 + jj;                             // raw slot (3 + jj) & 3, output j - 3 in operand slot jj
                __syncthreads();
                if (!(p.dbg & 1)) {                                // developer timing switches (PASSION_CONV_DBG): 1 no staging, 2 no MFMA loop, 4 no epilogue
                    commit(j, raw[(3 + jj) & 3]);
                    fetch(j + ND, raw[(3 + jj) & 3]);
                }
                compute(edgec, j - 3, eop[jj]);
                eload(j - 3 + ND, eop[jj]);
            }
        }
    };
    if constexpr (FIN) {
        const bool ewg_h = (h0 == 0 || h0 + MTH == Sh) && !(p.dbg & (8 | 256)), ewg_w = xw && !(p.dbg & (8 | 128));
        if (!ewg_h && !ewg_w) run(std::integral_constant<int, 0>{});
        else if (ewg_h && !ewg_w) run(std::integral_constant<int, 1>{});
        else if (!ewg_h) run(std::integral_constant<int, 2>{});
        else run(std::integral_constant<int, 3>{});
    } else {
        run(std::integral_constant<int, 0>{});
    }

    if ((DG && !FIN) || fstats) {
        // InstanceNorm-backward sums (MODE 2) / statistics of y (MODE 0): 16 voxel lanes (shuffles) -> 4 waves (LDS, fixed
        // order) -> this workgroup's own slot
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a1 = group16_sum(s1[mt][j]), a2 = group16_sum(s2v[mt][j]);
                if (n == 0) {
                    lsum[((wv * MT + mt) * 16 + g * 4 + j) * 2 + 0] = a1;
                    lsum[((wv * MT + mt) * 16 + g * 4 + j) * 2 + 1] = a2;
                }
            }
        __syncthreads();
        if (tid < 32 * MT) {
            const int c = tid >> 1, k2 = tid & 1;                  // c = 16 mt + row
            const float t = ((lsum[(0 * MT * 16 + c) * 2 + k2] + lsum[(1 * MT * 16 + c) * 2 + k2]) + lsum[(2 * MT * 16 + c) * 2 + k2]) +
                            lsum[(3 * MT * 16 + c) * 2 + k2];
            if (DG) {
                if (c < p.Cin) {
                    float* rs = p.sums + (long)slot * p.B * p.Cin * 2;
                    st_agent(&rs[((long)b * p.Cin + c) * 2 + k2], t);
                }
            } else {
                lsum[4 * MT * 16 * 2 + tid] = t;
            }
        }
        if (!DG) {
            __syncthreads();
            fwd_stats_commit(p.fst, p.Cout, b, slot, 0, 16, lsum + 4 * MT * 16 * 2, tid);
        } else if (mp.tickets) {
            // The sample's slots are folded HERE, by the workgroup that takes the last ticket (fixed slot order, in double: the sums
            // repeat bit for bit), into slot 0: the finish kernel reads one replica and the separate fold launch (14 per step) is gone.
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

template <int NB, int TW, int MODE, bool HAS_RES, int MT = 1, bool FIN = false>
int launch_march_res(ConvK& k, const MarchP& mp, hipStream_t st) {
    using G = GeoM<NB, TW, FIN>;
    static bool granted = false;
    if (!granted) {
        HIP_RET(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_march_kernel<NB, TW, MODE, HAS_RES, MT, FIN>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
        granted = true;
    }
    if (FIN) passion_prof_tag("conv_march_kernel<%d, %d, %d, %d, %d, 1>", NB, TW, MODE, HAS_RES ? 1 : 0, MT);
    else if (MT == 1) passion_prof_tag("conv_march_kernel<%d, %d, %d, %d>", NB, TW, MODE, HAS_RES ? 1 : 0);
    else passion_prof_tag("conv_march_kernel<%d, %d, %d, %d, %d>", NB, TW, MODE, HAS_RES ? 1 : 0, MT);
    hipLaunchKernelGGL((conv_march_kernel<NB, TW, MODE, HAS_RES, MT, FIN>), dim3(k.B * mp.nwg_b), dim3(256), G::LDS, st, k, mp);
    LAUNCH_RET();
    return 0;
}

template <int NB, int TW, int MODE>
int launch_march_one(ConvK& k, hipStream_t st) {
    using G = GeoM<NB, TW>;
    MarchP mp;
    mp.tickets = nullptr;
    mp.tilesH = k.Hi / MTH; mp.tilesW = k.Wi / TW;
    const bool fin = MODE == 2 && k.pwm != nullptr;          // fold-free input gradient with the final epilogue (means in k.pwm)
    const int dcount = (MODE == 2 && !fin) ? k.Di + 2 : k.Di;
    const int tiles = mp.tilesH * mp.tilesW;
    // D segments: about g_conv_target workgroups per launch, at least 8 output planes per segment (2 halo planes each)
    // (final form, 16 -> 8 over 4 x 128^3 with the face tiles first: 1024 workgroups 180 us, 1536 178, 2048 173, 4096 186)
    static const long fin_target = []() { const char* e = getenv("PASSION_MARCH_FIN_TARGET"); return e && atol(e) > 0 ? atol(e) : 2048L; }();
    // (two row tiles -- the 32 -> 16 input gradient of level 2 -- hold ONE workgroup per CU (352 registers): 256 workgroups are one round;
    //  4 x 64^3: 109 us at 512 .. 2048, 93 at 256)
    const long wg_target = ((fin) && fin_target) ? fin_target : ((MODE == 2 && k.Mpad == 32) ? 256L : (long)g_conv_target);
    int nseg = (int)hmax(1, (wg_target + (long)k.B * tiles - 1) / ((long)k.B * tiles));
    static const int min_planes = []() { const char* e = getenv("PASSION_MARCH_PLANES"); return e && atoi(e) > 0 ? atoi(e) : 16; }();
    const int nseg_want = nseg;
    nseg = (int)hmin(nseg, hmax(1, dcount / min_planes));
    // a launch that would leave CUs without a workgroup (one 64^3 sample: 32 tiles x 4 segments = 128 workgroups, each a chain of 19
    // planes) takes segments of 8 planes instead (round 6; PASSION_MARCH_FILL=0: off): 54 -> ~35 us for the batch-1 input gradients of
    // the fusion decoder's level 2, which run alone on the main stream
    static const bool fill = []() { const char* e = getenv("PASSION_MARCH_FILL"); return !(e && e[0] == '0'); }();
    if (fill && (long)k.B * tiles * nseg < 256) nseg = (int)hmin(nseg_want, hmax(nseg, hmin(dcount / 8, (255 + (long)k.B * tiles) / ((long)k.B * tiles))));
    if (MODE == 2 && !fin) nseg = (int)hmin(nseg, hmax(1, SUMS_SLOTS / tiles));      // one sums slot per workgroup of a sample
    mp.planes_per_seg = (dcount + nseg - 1) / nseg;
    mp.nseg_d = (dcount + mp.planes_per_seg - 1) / mp.planes_per_seg;
    mp.nwg_b = mp.nseg_d * tiles;
    static const bool edge_first = []() { const char* e = getenv("PASSION_MARCH_EDGEFIRST"); return !(e && e[0] == '0'); }();
    mp.edge_first = edge_first && mp.tilesH >= 3 && mp.tilesW >= 3;
    mp.k8 = k.march_k8 > 0 ? k.march_k8 : NB;
    if (fin) {
        if (TW != 16 || NB != 1 || k.Mpad != 16 || k.pad_mode != PASSION_PAD_REFLECT || k.Wi < 32 || k.Hi < 16) return 1;    // (a tile touches one edge per axis at most)
        k.pw_done = 1;
        return launch_march_res<1, 16, 2, false, 1, true>(k, mp, st);
    }
    if (MODE == 2) {
        if (mp.nwg_b > SUMS_SLOTS) return 1;
        k.sums_rep = mp.nwg_b;
        static const bool fold_in = []() { const char* e = getenv("PASSION_MARCH_FOLD"); return !(e && e[0] == '0'); }();
        if (fold_in && k.Cin <= 32) {                // the last workgroup of a sample folds its slots: one replica for the finish kernel
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
    if (MODE == 0 && k.residual) return launch_march_res<NB, TW, MODE, true>(k, mp, st);
    if (MODE == 2 && k.Mpad == 32) {
        if (TW != 16 || NB > 2) return 1;            // two row tiles: the 16-wide tiling only
        return launch_march_res<(NB > 2 ? 1 : NB), 16, 2, false, 2>(k, mp, st);
    }
    return launch_march_res<NB, TW, MODE, false>(k, mp, st);
}

}  // namespace

namespace pconv {

// 0 = launched, < 0 = HIP error, 1 = shape not covered (the caller goes on to the lean kernel)
// the shapes the marching kernel takes; K8 = 8-channel blocks of its K axis
bool conv_march_covers(const ConvK& k, int ksize, int stride, int mode, int* K8out) {
    if (!g_conv_march || ksize != 3 || stride != 1 || (mode != 0 && mode != 2)) return false;
    if (k.y_f32 || k.bd_rows || !(k.slope >= 0.f && k.slope <= 1.f)) return false;
    if (k.Hi % MTH || k.Wi % 16 || k.Di < 4) return false;
    const int rows = mode == 2 ? k.Cin : k.Cout;               // M axis
    if (rows > (mode == 2 ? 32 : 16) || rows % 4) return false;
    int K8;
    if (mode == 2) {
        if (k.Cout % 8) return false;
        K8 = k.Cout / 8;
        for (int s = 0; s < k.nseg; ++s) if (k.seg[s].dx && k.seg[s].bmod > 0) return false;
    } else {
        K8 = 0;
        for (int s = 0; s < k.nseg; ++s) {
            if (k.seg[s].C % 8) return false;
            K8 += k.seg[s].C / 8;
        }
        if (K8 * 8 != k.Cin) return false;
    }
    // 32 input channels forward (NB = 4: 108 weight registers, one wave per SIMD) beats the lean kernel from 2^19 voxels up
    // (4 x 64^3, 32 -> 16: 97 -> 54 us; 128^3, 32 -> 8: 173 -> 103; 4 x 32^3: 21 -> 36, stays lean); tools/k84_bench.sh
    static const bool k84 = []() { const char* e = getenv("PASSION_MARCH_K84"); return !(e && e[0] == '0'); }();
    const bool big4 = K8 == 4 && mode == 0 && k84 && (long)k.B * k.Di * k.Hi * k.Wi >= (1L << 19);
    if (K8 != 1 && K8 != 2 && !(K8 == 3 && mode == 0) && !big4) return false;   // a 24-channel concat (three 8-channel segments: RFNet's level-1 decoder) runs as NB = 4 with an empty block: 1270 -> 819 us at 10 x 128^3
    if (K8out) *K8out = K8;
    return true;
}

int launch_conv_march(ConvK& k, int ksize, int stride, int mode, const void* packed_bf16, hipStream_t st) {
    int K8 = 0;
    if (!conv_march_covers(k, ksize, stride, mode, &K8)) return 1;
    static const int force_tw = []() { const char* e = getenv("PASSION_MARCH_TW"); return e ? atoi(e) : 0; }();
    bool w32 = (k.Wi % 32 == 0) && K8 <= 2;
    if (force_tw == 16) w32 = false;
    if (mode == 2 && force_tw != 32) w32 = false;          // the dgrad keeps six tiles of operands in flight: 16-wide tiles fit three waves per SIMD
    const int ntap = 27;
    if (mode == 2) {
        k.Mpad = pad16(k.Cin);
        k.wpack = reinterpret_cast<const bf16_t*>(packed_bf16) + (long)ntap * pad8(k.Cin) * pad16(k.Cout);
    } else {
        k.Mpad = pad16(k.Cout);
        k.wpack = packed_bf16;
    }
    k.dbg = g_conv_dbg;
    if (K8 == 1 || (K8 == 2 && mode == 0)) {               // <= 8 rows: the depth-pair kernel (conv_pair.hip)
        const int rc = launch_conv_pair(k, mode, st);
        if (rc <= 0) return rc;
    }
#define MARCH_CASE(NBv)                                                                             \
    if (K8 == NBv) {                                                                                \
        if (w32) return mode == 2 ? launch_march_one<NBv, 32, 2>(k, st) : launch_march_one<NBv, 32, 0>(k, st); \
        return mode == 2 ? launch_march_one<NBv, 16, 2>(k, st) : launch_march_one<NBv, 16, 0>(k, st);           \
    }
    MARCH_CASE(1)
    MARCH_CASE(2)
#undef MARCH_CASE
    if (K8 == 3 && mode == 0) { k.march_k8 = 3; const int rc = launch_march_one<4, 16, 0>(k, st); k.march_k8 = 0; return rc; }
    if (K8 == 4 && mode == 0) return launch_march_one<4, 16, 0>(k, st);
    return 1;
}

}  // namespace pconv
