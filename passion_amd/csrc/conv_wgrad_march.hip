// Marching bf16 weight gradient of the 3x3x3 stride-1 convs with EIGHT output channels (the 128^3 level of every encoder /
// decoder of the reference, code/models/mmformer.py:24-64,66-189 via general_conv3d_prenorm, blocks.py:300-316):
//   dW[co][ci][tap] = sum_vox dY[vox][co] * act(x)[vox + tap][ci]      (+ dbias[co] = sum_vox dY[vox][co])
// Same operand images, descriptors, slab layout and fixed-order fold as wgrad_lean_kernel (conv_wgrad_lean.hip), which this
// kernel replaces where it applies.  What the counters and the arithmetic said about that kernel at Cout = 8: it is bound by
// LDS reads and MFMA issue, not by HBM -- a 16-row MFMA tile holds 8 useful rows, a 4x8x8 brick re-stages 2.34x its voxels
// as halo, and every K step re-reads both operands through ds_read_b64_tr_b16.  Two changes:
//   * the idle half of the M tile carries a SECOND depth tap.  Rows 0..7 are dY[v][co], rows 8..15 are dY[v - e_d][co] (the
//     previous plane, a per-lane address, no extra bytes); the columns are X[v + t'][ci] for the 18 taps t' with depth
//     offset -1 and 0.  Row block 0 yields the depth taps -1 and 0, row block 1 yields 0 (dropped) and +1: 9 column tiles
//     of 16 instead of 14, 27/36 of the MFMA work useful instead of 27/56;
//   * a workgroup owns an 8 x 16 tile of the (H, W) plane and MARCHES along D (as conv_march.hip does for forward / dgrad): each
//     input plane is fetched once per tile (halo 10/8 x 18/16 = 1.41 instead of 2.34), planes live in a 4-slot LDS ring, one
//     barrier per plane, four planes of loads in flight in registers, all index arithmetic (padding, concat lookup, mask,
//     statistics) done once per tile.  Each wave owns two tile rows = one K step of 32 voxels per plane and all 9 column
//     tiles; the four waves' partial sums meet in LDS once, in fixed order, before the slot is written.
// The bias gradient is summed by the threads that stage dY (packed f32 adds), folded in fixed order at the end.
#include "conv_common.h"
#include "prof.h"

namespace pconv { int g_wgrad_march = 1; extern int g_conv_dbg; }
using namespace pconv;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <typename T> __device__ __forceinline__ T gload(const void* ptr) {      // a GLOBAL load, whatever the pointer's history
    return *reinterpret_cast<const T __attribute__((address_space(1)))*>(reinterpret_cast<unsigned long long>(ptr));
}

__device__ __host__ constexpr int tapoff(int hw) { return ((hw / 3) * 20 + hw % 3) * 16; }     // staged-plane byte offset of in-plane tap hw

__device__ __forceinline__ float uniform(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

__device__ __forceinline__ bf16x8 tr_read8(const char* p0, const char* p1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p1));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

constexpr int WTH = 8, WTW = 16;                  // tile of the (H, W) plane
constexpr int XRW = 20;                           // voxels per staged row: 18 used; 320 bytes = 64 mod 256, so the three rows a
                                                  // tap column touches start in different bank quarters
constexpr int XVOX = (WTH + 2) * (WTW + 2);       // 180 voxels fetched per plane
constexpr int XSLOT = (WTH + 2) * XRW * 16;       // 3200 bytes
constexpr int YSLOT = WTH * WTW * 16 + 64;        // 2112: the previous plane's rows sit 64 bytes off in bank space
constexpr int NSL = 4;                            // ring slots = planes in flight in registers
constexpr int NT = 9;                             // column tiles: 18 taps x 8 channels
constexpr int XBYTES = NSL * XSLOT, YBYTES = NSL * YSLOT;
constexpr int RED_BYTES = 4 * NT * 64 * 16;       // [wave][tile][lane] f32x4
constexpr int LDS_BYTES = (XBYTES + YBYTES > RED_BYTES) ? XBYTES + YBYTES : RED_BYTES;

struct WMarchP {
    int wgs_per_b;                 // workgroups (= slots of the partial slab) per sample
    int ncols, tilesW;             // (H, W) tiles per sample
    int nseg_d, planes_per_seg;    // D segments per tile; a unit = (tile, segment), dealt round-robin to the sample's workgroups
};

// IND: a second set of column tiles over the 0/1 image of x_hat > 0 (same taps, same rows): P[co][ci][tap] = sum_vox dY[vox][co] *
// [x_hat[vox + tap][ci] > 0] goes to the slab p.spos.  With it the InstanceNorm-backward means of the layer follow from this
// weight gradient (march_ident_kernel below; conv_pw.hip has the identities) and the input-gradient kernel writes dx itself.
template <int DBG, bool IND>
#ifndef WM_IND_WAVES
#define WM_IND_WAVES 2      // (round 6: 148 -> 134 us at 8 -> 8, 263 -> 228 at 16 -> 8 over 4 x 128^3 against three waves per SIMD: the 19 fragments of a plane in flight instead of a third of them; one wave: 199 / 346)
#endif
#ifndef WM_WAVES
#define WM_WAVES 2          // (round 6, with the plane tiles carried in registers: 32 -> 8 over 2 x 128^3 159 -> 138 us, 24 -> 8 206 -> 178, 8 -> 8 at batch 1 and
                            // 512 slots 35 -> 28; four waves and 128 registers with the carry: 44, without: 35; three waves: 136 / 186 / 32)
#endif
#ifndef WM_IND_DBG
#define WM_IND_DBG 0        // developer timing builds of the indicator kernel (tools/build_variant.sh): the DBG bits below
#endif
#ifndef WM_CARRY
#define WM_CARRY 1          // plane tiles carried in registers from one K plane to the next (below): -1 = the indicator kernel only, 0 / 1 = none / all
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(IND ? WM_IND_WAVES : WM_WAVES, IND ? WM_IND_WAVES : WM_WAVES))) void wgrad_march_kernel(ConvK p, WMarchP mp, float* __restrict__ dW, float* __restrict__ dbias) {
    wgrad_clear(p);
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES + (IND ? XBYTES : 0)];
    char* const lx = lds;
    char* const ly = lds + XBYTES;
    char* const li = lds + XBYTES + YBYTES;           // IND: ring of the indicator planes (behind the reduction buffer's span)

    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = n >> 2, pq = n & 3;
    const int gb0 = blockIdx.y;                       // this workgroup's 8-channel block of the virtual concat
    const int Sd = p.Di, Sh = p.Hi, Sw = p.Wi;

    const int bx = xcd_swizzle(blockIdx.x, gridDim.x);
    const int b = bx / mp.wgs_per_b;                  // >= B: spare workgroup, flushes a zero slot
    const int i0 = bx % mp.wgs_per_b;
    const int nunits = (b < p.B) ? mp.ncols * mp.nseg_d : 0;

    // ---- once per workgroup: which tensor / channels, mask, statistics
    int sg = 0, c0 = 0, C = 8, cbase = 0;
    bool exist = false;
    {
        int r = gb0;
#pragma unroll
        for (int s2 = 0; s2 < PASSION_MAX_SEG; ++s2) {
            if (s2 < p.nseg && !exist) {
                const int nb = p.seg[s2].C / 8;
                if (r < nb) { exist = true; sg = s2; c0 = r * 8; C = p.seg[s2].C; cbase = p.seg[s2].cbase; }
                r -= nb;
            }
        }
    }
    const bf16_t* src = nullptr;
    bool present = false, has_norm = false;
    f32x2 sc[4], sh[4];                                // x_hat = x * rstd + shift, channel pairs (the forward marching kernel's form)
#pragma unroll
    for (int c = 0; c < 4; ++c) { sc[c] = (f32x2){1.f, 1.f}; sh[c] = (f32x2){0.f, 0.f}; }
    if (exist && i0 < nunits) {
#pragma unroll
        for (int s2 = 0; s2 < PASSION_MAX_SEG; ++s2) {
            if (s2 == sg) {
                const SegD& sd = p.seg[s2];
                const int bs = sd.bsrc(b);
                present = !sd.mask || sd.mask[(long)b * sd.mask_ld + sd.mask_col] != 0;
                src = reinterpret_cast<const bf16_t*>(sd.x) + (long)bs * sd.bstride + c0;
                has_norm = sd.stats != nullptr;
                if (sd.stats) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const f32x4 st4 = *reinterpret_cast<const f32x4*>(sd.stats + ((long)bs * sd.C + c0 + 2 * c) * 2);   // mean, rstd, mean, rstd
                        sc[c] = (f32x2){st4[1], st4[3]};
                        sh[c] = (f32x2){uniform(-st4[0] * st4[1]), uniform(-st4[2] * st4[3])};      // workgroup-uniform: keep them scalar
                    }
                }
            }
        }
    }
    const bf16_t* dyb = reinterpret_cast<const bf16_t*>(p.dy) + (long)(b < p.B ? b : 0) * Sd * Sh * Sw * 8;      // Cout == 8

    // ---- per-lane constants of the MFMA loop.  Wave wv owns tile rows 2 wv, 2 wv + 1; K index g * 8 + e of its step is the
    // voxel (row 2 wv + (g >> 1), column (g & 1) * 8 + e); a transposing read hands lane n channel n of four voxels, so the lane
    // LOADS 8 bytes of voxel q (+ 4 in the second read): A = channels (pq & 1) * 4.. of dY, current plane for pq < 2 (rows 0..7),
    // previous plane for pq >= 2 (rows 8..15); B = channels (pq & 1) * 4.. of tap 2 j + (pq >> 1) of column tile j.
    const int row = 2 * wv + (g >> 1), col = (g & 1) * 8 + q;
    const int yoff = (row * WTW + col) * 16 + (pq & 1) * 8;
    const bool yprev = pq >= 2;
    // Column tile j = two taps (lane half h = pq >> 1 picks one) whose staged voxels differ by a CONSTANT, so that three lane
    // bases + instruction immediates address all nine tiles: in-plane tap pairs (0,1) (3,4) (6,7) are one voxel apart, (2,5) one
    // row; tap 8 of plane P - 1 pairs with tap 8 of plane P (tile 4).  Tiles 0..3 read plane P - 1, tiles 5..8 plane P.
    const int h = pq >> 1;
    const int xlane = (row * XRW + col) * 16 + (pq & 1) * 8;
    const int xlaneA = xlane + h * 16, xlaneB = xlane + h * (XRW * 16);
    constexpr bool CARRY = WM_CARRY < 0 ? IND : (WM_CARRY != 0);
    f32x4 acc[NT], acc2[IND ? NT : 1];
    bf16x8 bprev[4], iprev[IND ? 4 : 1];              // plane P - 1's four in-plane tiles, carried from the plane before
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < (IND ? NT : 1); ++j) acc2[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool bias_block = (dbias != nullptr) && (gb0 == 0);
    f32x2 bsum[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) bsum[c] = (f32x2){0.f, 0.f};
    const f32x2 slope2 = {p.slope, p.slope};

    // ---- staging roles: threads 0..179 one voxel of the X plane (tile + halo), threads 128..255 one voxel of the dY plane
    const bool xthr = tid < XVOX, ythr = tid >= 128;
    const int zh = tid / (WTW + 2), zw = tid % (WTW + 2);
    const int xlds = (zh * XRW + zw) * 16;
    const int yv = tid - 128, vh = yv >> 4, vw = yv & 15;
    const int ylds = yv * 16;
    const bool reflect = p.pad_mode == PASSION_PAD_REFLECT;
    const long plane_vox = (long)Sh * Sw;

    for (int unit = i0; unit < nunits; unit += mp.wgs_per_b) {
        const int colidx = unit % mp.ncols, dseg = unit / mp.ncols;
        const int h0 = (colidx / mp.tilesW) * WTH, w0 = (colidx % mp.tilesW) * WTW;
        const int dlo = dseg * mp.planes_per_seg, dhi = min(dlo + mp.planes_per_seg, Sd);
        const int L = dhi - dlo;
        if (L <= 0) continue;
        const int nstage = L + 2;                     // stage j = plane dlo - 1 + j of X and of dY (zero outside [dlo, dhi))
        // this thread's source voxel inside a plane, padding resolved once per tile.  Every global load of the plane loop is
        // UNCONDITIONAL (clamped address, data dropped at commit time) and issued NSL iterations before its use: one load inside
        // a branch makes the compiler wait with vmcnt(0) and drains the whole prefetch queue (conv_march.hip).  Threads without
        // a voxel of their own (X: 180.., dY: ..127) re-read one fixed voxel -- a single cache line per wave instruction.
        bool xok = xthr && present;
        unsigned xel, yel;
        {
            int s_h = h0 - 1 + (xthr ? zh : 1), s_w = w0 - 1 + (xthr ? zw : 1);
            if (reflect) { s_h = reflect_idx(s_h, Sh); s_w = reflect_idx(s_w, Sw); }
            else xok = xok && s_h >= 0 && s_h < Sh && s_w >= 0 && s_w < Sw;
            s_h = min(max(s_h, 0), Sh - 1); s_w = min(max(s_w, 0), Sw - 1);
            xel = (unsigned)(s_h * Sw + s_w) * (unsigned)C;
            yel = (unsigned)((h0 + (ythr ? vh : 0)) * Sw + w0 + (ythr ? vw : 0)) * 8u;
        }
        const unsigned xplane_el = (unsigned)(Sh * Sw) * (unsigned)C, yplane_el = (unsigned)(Sh * Sw) * 8u;
        u32x4 ra[NSL], rd[NSL];
        auto fetch = [&](int u, int j) {
            if (DBG & 1) return;                      // developer timing builds (PASSION_CONV_DBG): 1 no loads, 2 no MFMA loop,
                                                      // 8 one B read per plane, 16 reads but no MFMAs, 32 no commit
            const int pl = dlo - 1 + j;
            int xp = reflect ? reflect_idx(pl, Sd) : pl;
            xp = min(max(xp, 0), Sd - 1);
            const int yp = min(max(pl, dlo), dhi - 1);
            ra[u] = gload<u32x4>(src + ((unsigned)xp * xplane_el + xel));
            rd[u] = gload<u32x4>(dyb + ((unsigned)yp * yplane_el + yel));
        };
        auto commit = [&](int u, int j) {
            const int pl = dlo - 1 + j;
            const bool okd = reflect || (pl >= 0 && pl < Sd);
            const bool yin = pl >= dlo && pl < dhi;       // (stages past the unit's end are never committed)
            if (xthr) {
                u32x4 o = (u32x4){0u, 0u, 0u, 0u};    // zero padding is applied AFTER norm + activation
                u32x4 oi = (u32x4){0u, 0u, 0u, 0u};   // IND: bf16 1.0 where x_hat > 0 (0x3f80 per half)
                if (xok && okd) {
                    o = ra[u];
                    if (has_norm) {
                        bf16x8 ob;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {      // bf16 pairs -> f32: low half = even channel
                            const f32x2 xv = {__uint_as_float(ra[u][c] << 16), __uint_as_float(ra[u][c] & 0xffff0000u)};
                            f32x2 t = __builtin_elementwise_fma(xv, sc[c], sh[c]);
                            const f32x2 ts = t * slope2;   // LeakyReLU for 0 <= slope <= 1: max(t, slope t)
                            t.x = fmaxf(t.x, ts.x);
                            t.y = fmaxf(t.y, ts.y);
                            ob[2 * c] = (__bf16)t.x; ob[2 * c + 1] = (__bf16)t.y;
                        }
                        o = __builtin_bit_cast(u32x4, ob);
                        if (IND) {                     // x_hat > 0 <=> the activated bf16 value is a positive number: min(max(v, 0), 1) * 0x3f80 on the
                                                       // packed halves, three instructions per channel pair instead of two compares, two selects and an or
                                                       // (as instructions: the vector form of this came back as compares and selects; 16 -> 8: 186 -> 181 us)
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                unsigned v;
                                asm("v_pk_max_i16 %0, %1, 0" : "=v"(v) : "v"(o[c]));
                                asm("v_pk_min_i16 %0, %1, %2" : "=v"(v) : "v"(v), "v"(0x00010001u));
                                asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(oi[c]) : "v"(v), "v"(0x3f803f80u));
                            }
                        }
                    }
                }
                *reinterpret_cast<u32x4*>(lx + u * XSLOT + xlds) = o;
                if (IND) *reinterpret_cast<u32x4*>(li + u * XSLOT + xlds) = oi;
            }
            if (ythr) {
                const u32x4 o = yin ? rd[u] : (u32x4){0u, 0u, 0u, 0u};
                *reinterpret_cast<u32x4*>(ly + u * YSLOT + ylds) = o;
                if (bias_block && yin) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        bsum[c] += (f32x2){__uint_as_float(o[c] << 16), __uint_as_float(o[c] & 0xffff0000u)};
                }
            }
        };
        // Column tiles 0..3 of a K plane (plane P - 1's in-plane tap pairs) are the fragments tiles 5..8 read one plane earlier (same
        // lanes, same addresses, the slot that was `scur` then): they stay in registers (round 6) -- 22 transposing reads per plane
        // and wave instead of 38 (12 instead of 20 without the indicator).
        auto plane_tile = [&](int jj, int slot) { return jj < 3 ? lx + xlaneA + slot * XSLOT + tapoff(3 * jj) : lx + xlaneB + slot * XSLOT + tapoff(2); };
        auto load_prev = [&](int slot) {                 // a unit's first plane: nothing was read before it
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const char* bp = plane_tile(jj, slot);
                bprev[jj] = tr_read8(bp, bp + 64);
                if (IND) { const char* bi = bp + (li - lx); iprev[jj] = tr_read8(bi, bi + 64); }
            }
        };
        // One K plane: reads, the four tiles whose operands are in registers already, the five new ones.  (Measured and not kept, round 6,
        // 8 -> 8 over 4 x 128^3 at 106 us: the reads and the register-operand MFMAs ahead of the iteration's commit in program order
        // -- same time, and 44 against 34 us on the plain kernel; eight planes of loads in flight instead of four: 110; eight waves per
        // workgroup, four that stage and four that multiply: 150 -- the staging waves alone take 128, 60 without their global loads,
        // 93 when every load hits the same lines; three waves per SIMD over 768 slots with a three-way depth split: 147.)
        bf16x8 fa, fb[NT - 4], fi[IND ? NT - 4 : 1];
        auto compute_reads = [&](int sp, int scur) {     // ring slots of plane P - 1 and of plane P: both committed before this barrier
            const char* ya = ly + yoff + (yprev ? sp : scur) * YSLOT;
            fa = tr_read8(ya, ya + 64);
            const int x4 = xlane + tapoff(8) + (h ? scur : sp) * XSLOT;
#pragma unroll
            for (int j = (CARRY ? 4 : 0); j < NT; ++j) {
                const int jj = j < 4 ? j : j - 5;
                const char* bp = (j == 4) ? lx + x4 : plane_tile(jj, j < 4 ? sp : scur);
                if ((DBG & 8) && j > 0) bp = lx + xlaneA + sp * XSLOT;
                if (j < 4) { bprev[j] = tr_read8(bp, bp + 64); if (IND) { const char* bi = bp + (li - lx); iprev[j] = tr_read8(bi, bi + 64); } }
                else { fb[j - 4] = tr_read8(bp, bp + 64); if (IND) { const char* bi = bp + (li - lx); fi[j - 4] = tr_read8(bi, bi + 64); } }
            }
        };
        auto mma = [&](int j, const bf16x8& bv, const bf16x8& iv) {
            if (DBG & 16) acc[j][0] += (float)bv[0] + (float)fa[0];
            else acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, bv, acc[j], 0, 0, 0);
            if (IND) acc2[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, iv, acc2[j], 0, 0, 0);
        };
        auto compute_prev = [&]() {                      // tiles 0..3: operands in registers since the plane before
#pragma unroll
            for (int j = 0; j < 4; ++j) mma(j, bprev[j], iprev[IND ? j : 0]);
        };
        auto compute_new = [&]() {
#pragma unroll
            for (int j = 4; j < NT; ++j) {
                mma(j, fb[j - 4], fi[IND ? j - 4 : 0]);
                if (j > 4) { bprev[j - 5] = fb[j - 4]; if (IND) iprev[j - 5] = fi[j - 4]; }
            }
        };
        // ---- pipeline.  iteration j: barrier; commit stage j; fetch stage j + NSL; compute K plane c = j - 2 from stages j - 2 and
        // j - 1, both committed before this barrier (slot j & 3 held stage j - 4, last read in iteration j - 2).  The body has no
        // branch around a load or an MFMA: dead iterations past the unit load clamped planes, commit ZERO dY planes and so add
        // nothing.
        __syncthreads();                              // the previous unit's last readers are done with the ring
#pragma unroll
        for (int u = 0; u < NSL; ++u) fetch(u, u);
#pragma unroll
        for (int j = 0; j < 2; ++j) {                 // stages 0, 1: nothing to compute yet
            __syncthreads();
            if (!(DBG & 32)) commit(j, j);
            fetch(j, j + NSL);
            if (j == 1 && CARRY) load_prev(0);     // stage 0 is behind this iteration's barrier
        }
        for (int j0 = 2; j0 <= nstage; j0 += NSL) {
#pragma unroll
            for (int jj = 0; jj < NSL; ++jj) {
                const int j = j0 + jj, u = (2 + jj) % NSL;
                __syncthreads();
                if (!(DBG & 32)) commit(u, j);
                fetch(u, j + NSL);
                if (!(DBG & 2)) { compute_reads((u + 2) % NSL, (u + 3) % NSL); compute_prev(); compute_new(); }
            }
        }
    }

    // ---- the four waves hold partial sums over disjoint voxels of the SAME outputs: combine through LDS in fixed order, and
    // write this workgroup's slot of the partial-sum slab (plain stores; reduce_parts_kernel folds the slots)
    f32x4* red = reinterpret_cast<f32x4*>(lds);
    auto flush = [&](const f32x4 (&av)[NT], float* slab) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NT; ++j) red[(wv * NT + j) * 64 + lane] = av[j];
        __syncthreads();
        if (slab && exist) {
            float* slot = slab + (long)bx * p.wslot;
            for (int e = tid; e < NT * 64; e += 256) {
                const int j = e >> 6, l2 = e & 63, n2 = l2 & 15, g2 = l2 >> 4;
                f32x4 v = red[e];
#pragma unroll
                for (int w2 = 1; w2 < 4; ++w2) {
                    const f32x4 o = red[w2 * NT * 64 + e];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += o[r];
                }
                const int h2 = n2 >> 3, ci = n2 & 7, jj = j < 4 ? j : j - 5;
                const int dsel = (j == 4) ? h2 : (j > 4 ? 1 : 0), s = g2 >> 1;
                const int hw = (j == 4) ? 8 : (jj < 3 ? 3 * jj + h2 : 2 + 3 * h2);
                if (s == 1 && dsel == 0) continue;        // depth tap 0 a second time
                const int tap = (dsel + s) * 9 + hw;
                const long cig = cbase + c0 + ci;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = (g2 & 1) * 4 + r;
                    slot[((long)co * p.Cin + cig) * 27 + tap] = v[r];
                }
            }
        }
    };
    flush(acc, dW);
    if (IND) {                                         // the indicator slab: same slots (stride p.wslot), base p.spos
        f32x4 a2[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) a2[j] = acc2[j];
        flush(a2, p.spos);
    }
    if (bias_block) {
        __syncthreads();
        float* bred = reinterpret_cast<float*>(lds);              // [128][9], then [8][8]
        float* bred2 = bred + 128 * 9;
        if (ythr) {
#pragma unroll
            for (int c = 0; c < 8; ++c) bred[yv * 9 + c] = bsum[c >> 1][c & 1];
        }
        __syncthreads();
        if (tid < 64) {
            const int c = tid & 7, part = tid >> 3;
            float sacc = 0.f;
            for (int t = part * 16; t < part * 16 + 16; ++t) sacc += bred[t * 9 + c];
            bred2[part * 8 + c] = sacc;
        }
        __syncthreads();
        if (tid < 8) {
            float sacc = 0.f;
#pragma unroll
            for (int part = 0; part < 8; ++part) sacc += bred2[part * 8 + tid];
            dbias[(long)bx * p.wslot + tid] = sacc;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// SIXTEEN output channels (the 64^3 level: 16 -> 16, 32 -> 16): the same march with all 16 MFMA rows holding dY of the CURRENT
// plane, so a K plane needs the three X planes P - 1 .. P + 1 and the 27 taps x 8 channels are 14 column tiles (tile j = taps 2 j,
// 2 j + 1 for the lane halves; tap 27 does not exist).  wgrad_lean_kernel<3, 1> ran these layers at 1.2-1.6 TB/s (a 4x8x8 brick
// re-stages 2.34x its voxels as halo and pays two barriers per brick).
constexpr int Y16SLOT = WTH * WTW * 32 + 64;
constexpr int NT16 = 14;
constexpr int Y16BYTES = NSL * Y16SLOT;
constexpr int RED16_BYTES = 4 * (NT16 / 2) * 64 * 16;          // the waves' partial sums meet in two halves of seven tiles
constexpr int LDS16_BYTES = (XBYTES + Y16BYTES > RED16_BYTES) ? XBYTES + Y16BYTES : RED16_BYTES;

__global__ __launch_bounds__(256) void wgrad_march16_kernel(ConvK p, WMarchP mp, float* __restrict__ dW, float* __restrict__ dbias) {
    wgrad_clear(p);
    __shared__ __attribute__((aligned(16))) char lds[LDS16_BYTES];
    char* const lx = lds;
    char* const ly = lds + XBYTES;

    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = n >> 2, pq = n & 3;
    const int gb0 = blockIdx.y;                       // this workgroup's 8-channel block of the virtual concat
    const int Sd = p.Di, Sh = p.Hi, Sw = p.Wi;

    const int bx = xcd_swizzle(blockIdx.x, gridDim.x);
    const int b = bx / mp.wgs_per_b;                  // >= B: spare workgroup, flushes a zero slot
    const int i0 = bx % mp.wgs_per_b;
    const int nunits = (b < p.B) ? mp.ncols * mp.nseg_d : 0;

    int sg = 0, c0 = 0, C = 8, cbase = 0;
    bool exist = false;
    {
        int r = gb0;
#pragma unroll
        for (int s2 = 0; s2 < PASSION_MAX_SEG; ++s2) {
            if (s2 < p.nseg && !exist) {
                const int nb = p.seg[s2].C / 8;
                if (r < nb) { exist = true; sg = s2; c0 = r * 8; C = p.seg[s2].C; cbase = p.seg[s2].cbase; }
                r -= nb;
            }
        }
    }
    const bf16_t* src = nullptr;
    bool present = false, has_norm = false;
    f32x2 sc[4], sh[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { sc[c] = (f32x2){1.f, 1.f}; sh[c] = (f32x2){0.f, 0.f}; }
    if (exist && i0 < nunits) {
#pragma unroll
        for (int s2 = 0; s2 < PASSION_MAX_SEG; ++s2) {
            if (s2 == sg) {
                const SegD& sd = p.seg[s2];
                const int bs = sd.bsrc(b);
                present = !sd.mask || sd.mask[(long)b * sd.mask_ld + sd.mask_col] != 0;
                src = reinterpret_cast<const bf16_t*>(sd.x) + (long)bs * sd.bstride + c0;
                has_norm = sd.stats != nullptr;
                if (sd.stats) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const f32x4 st4 = *reinterpret_cast<const f32x4*>(sd.stats + ((long)bs * sd.C + c0 + 2 * c) * 2);
                        sc[c] = (f32x2){st4[1], st4[3]};
                        sh[c] = (f32x2){uniform(-st4[0] * st4[1]), uniform(-st4[2] * st4[3])};
                    }
                }
            }
        }
    }
    const bf16_t* dyb = reinterpret_cast<const bf16_t*>(p.dy) + (long)(b < p.B ? b : 0) * Sd * Sh * Sw * 16;      // Cout == 16

    // per-lane constants of the MFMA loop (see wgrad_march_kernel): wave wv owns tile rows 2 wv, 2 wv + 1; lane (q, pq) LOADS 8 bytes of
    // voxel q: A = channels 4 pq .. of dY (rows = the 16 output channels), B = channels (pq & 1) * 4 .. of tap 2 j + (pq >> 1) of tile j
    const int row = 2 * wv + (g >> 1), col = (g & 1) * 8 + q;
    const int yoff = (row * WTW + col) * 32 + pq * 8;
    const int h = pq >> 1;
    const int xlane = (row * XRW + col) * 16 + (pq & 1) * 8;
    int boff[NT16];                                   // in-plane byte offset of the lane's tap of tile j; its plane (0 .. 2) in bits 30..31
#pragma unroll
    for (int j = 0; j < NT16; ++j) {
        const int t = min(2 * j + h, 26);
        boff[j] = xlane + tapoff(t % 9) + ((t / 9) << 28);
    }
    f32x4 acc[NT16];
#pragma unroll
    for (int j = 0; j < NT16; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool bias_block = (dbias != nullptr) && (gb0 == 0);
    f32x2 bsum[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) bsum[c] = (f32x2){0.f, 0.f};
    const f32x2 slope2 = {p.slope, p.slope};

    // staging roles: threads 0..179 one voxel of the X plane (tile + halo); every thread one 16-byte half of a dY voxel
    const bool xthr = tid < XVOX;
    const int zh = tid / (WTW + 2), zw = tid % (WTW + 2);
    const int xlds = (zh * XRW + zw) * 16;
    const int yv = tid >> 1, yhalf = tid & 1, vh = yv >> 4, vw = yv & 15;
    const int ylds = yv * 32 + yhalf * 16;
    const bool reflect = p.pad_mode == PASSION_PAD_REFLECT;

    for (int unit = i0; unit < nunits; unit += mp.wgs_per_b) {
        const int colidx = unit % mp.ncols, dseg = unit / mp.ncols;
        const int h0 = (colidx / mp.tilesW) * WTH, w0 = (colidx % mp.tilesW) * WTW;
        const int dlo = dseg * mp.planes_per_seg, dhi = min(dlo + mp.planes_per_seg, Sd);
        const int L = dhi - dlo;
        if (L <= 0) continue;
        bool xok = xthr && present;
        unsigned xel, yel;
        {
            int s_h = h0 - 1 + (xthr ? zh : 1), s_w = w0 - 1 + (xthr ? zw : 1);
            if (reflect) { s_h = reflect_idx(s_h, Sh); s_w = reflect_idx(s_w, Sw); }
            else xok = xok && s_h >= 0 && s_h < Sh && s_w >= 0 && s_w < Sw;
            s_h = min(max(s_h, 0), Sh - 1); s_w = min(max(s_w, 0), Sw - 1);
            xel = (unsigned)(s_h * Sw + s_w) * (unsigned)C;
            yel = (unsigned)((h0 + vh) * Sw + w0 + vw) * 16u + (unsigned)yhalf * 8u;
        }
        const unsigned xplane_el = (unsigned)(Sh * Sw) * (unsigned)C, yplane_el = (unsigned)(Sh * Sw) * 16u;
        u32x4 ra[NSL], rd[NSL];
        // stage j = X plane dlo - 1 + j and dY plane dlo - 1 + j (zero outside [dlo, dhi))
        auto fetch = [&](int u, int j) {
            const int pl = dlo - 1 + j;
            int xp = reflect ? reflect_idx(pl, Sd) : pl;
            xp = min(max(xp, 0), Sd - 1);
            const int yp = min(max(pl, dlo), dhi - 1);
            ra[u] = gload<u32x4>(src + ((unsigned)xp * xplane_el + xel));
            rd[u] = gload<u32x4>(dyb + ((unsigned)yp * yplane_el + yel));
        };
        auto commit = [&](int u, int j) {
            const int pl = dlo - 1 + j;
            const bool okd = reflect || (pl >= 0 && pl < Sd);
            const bool yin = pl >= dlo && pl < dhi;
            if (xthr) {
                u32x4 o = (u32x4){0u, 0u, 0u, 0u};    // zero padding is applied AFTER norm + activation
                if (xok && okd) {
                    o = ra[u];
                    if (has_norm) {
                        bf16x8 ob;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const f32x2 xv = {__uint_as_float(ra[u][c] << 16), __uint_as_float(ra[u][c] & 0xffff0000u)};
                            f32x2 t = __builtin_elementwise_fma(xv, sc[c], sh[c]);
                            const f32x2 ts = t * slope2;
                            t.x = fmaxf(t.x, ts.x);
                            t.y = fmaxf(t.y, ts.y);
                            ob[2 * c] = (__bf16)t.x; ob[2 * c + 1] = (__bf16)t.y;
                        }
                        o = __builtin_bit_cast(u32x4, ob);
                    }
                }
                *reinterpret_cast<u32x4*>(lx + u * XSLOT + xlds) = o;
            }
            const u32x4 oy = yin ? rd[u] : (u32x4){0u, 0u, 0u, 0u};
            *reinterpret_cast<u32x4*>(ly + u * Y16SLOT + ylds) = oy;
            if (bias_block && yin) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    bsum[c] += (f32x2){__uint_as_float(oy[c] << 16), __uint_as_float(oy[c] & 0xffff0000u)};
            }
        };
        // K plane from the ring slots s0, s1, s2 of X planes P - 1, P, P + 1; dY of plane P sits in slot s1
        auto compute = [&](int s0, int s1, int s2) {
            const char* ya = ly + yoff + s1 * Y16SLOT;
            const bf16x8 a = tr_read8(ya, ya + 128);
            const int so[3] = {s0 * XSLOT, s1 * XSLOT, s2 * XSLOT};
#pragma unroll
            for (int j = 0; j < NT16; ++j) {
                const int pd = boff[j] >> 28;                          // per lane half only for tile 4 (taps 8 | 9)
                const char* bp = lx + (boff[j] & 0x0fffffff) + (pd == 0 ? so[0] : (pd == 1 ? so[1] : so[2]));
                const bf16x8 bv = tr_read8(bp, bp + 64);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bv, acc[j], 0, 0, 0);
            }
        };

        // pipeline.  iteration j: barrier; commit stage j; fetch stage j + NSL; compute K plane j - 3 (stages j - 3 .. j - 1, all committed
        // before this barrier; slot j & 3 held stage j - 4, last read in iteration j - 1).  Dead iterations commit zero dY planes.
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NSL; ++u) fetch(u, u);
#pragma unroll
        for (int j = 0; j < 3; ++j) {                 // stages 0 .. 2: nothing to compute yet
            __syncthreads();
            commit(j, j);
            fetch(j, j + NSL);
        }
        for (int j0 = 3; j0 < L + 3; j0 += NSL) {
#pragma unroll
            for (int jj = 0; jj < NSL; ++jj) {
                const int j = j0 + jj, u = (3 + jj) % NSL;
                __syncthreads();
                commit(u, j);
                fetch(u, j + NSL);
                compute((u + 1) % NSL, (u + 2) % NSL, (u + 3) % NSL);
            }
        }
    }

    // the four waves hold partial sums over disjoint voxels of the same outputs: combine through LDS in fixed order (two halves of seven
    // tiles: the buffer shares the ring's space), write this workgroup's slot of the partial-sum slab
    f32x4* red = reinterpret_cast<f32x4*>(lds);
    constexpr int HT = NT16 / 2;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < HT; ++j) red[(wv * HT + j) * 64 + lane] = acc[half * HT + j];
        __syncthreads();
        if (dW && exist) {
            float* slot = dW + (long)bx * p.wslot;
            for (int e = tid; e < HT * 64; e += 256) {
                const int j = half * HT + (e >> 6), l2 = e & 63, n2 = l2 & 15, g2 = l2 >> 4;
                f32x4 v = red[e];
#pragma unroll
                for (int w2 = 1; w2 < 4; ++w2) {
                    const f32x4 o = red[w2 * HT * 64 + e];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += o[r];
                }
                const int tap = 2 * j + (n2 >> 3), ci = n2 & 7;
                if (tap > 26) continue;
                const long cig = cbase + c0 + ci;
#pragma unroll
                for (int r = 0; r < 4; ++r) slot[((long)(g2 * 4 + r) * p.Cin + cig) * 27 + tap] = v[r];
            }
        }
    }
    if (bias_block) {
        __syncthreads();
        float* bred = reinterpret_cast<float*>(lds);              // [256][9]: thread t holds channels 8 (t & 1) .. of voxel t >> 1
        float* bred2 = bred + 256 * 9;
#pragma unroll
        for (int c = 0; c < 8; ++c) bred[tid * 9 + c] = bsum[c >> 1][c & 1];
        __syncthreads();
        if (tid < 128) {                                          // (channel 0 .. 15) x (8 slices of 16 voxels)
            const int c = tid & 15, part = tid >> 4;
            float sacc = 0.f;
            for (int v = part * 16; v < part * 16 + 16; ++v) sacc += bred[(2 * v + (c >> 3)) * 9 + (c & 7)];
            bred2[part * 16 + c] = sacc;
        }
        __syncthreads();
        if (tid < 16) {
            float sacc = 0.f;
#pragma unroll
            for (int part = 0; part < 8; ++part) sacc += bred2[part * 16 + tid];
            dbias[(long)bx * p.wslot + tid] = sacc;
        }
    }
}

}  // namespace

namespace pconv {

// 0 = launched, < 0 = HIP error, 1 = shape not covered (the caller goes on to wgrad_lean_kernel).  Same slab plan as the lean
// kernel: gx slots in sample order, gy = 8-channel blocks of the virtual concat; k.per is set so that the grouped fold
// (conv.hip) derives the same workgroups-per-sample from it.
int launch_wgrad_march(ConvK& k, int ksize, int stride, int MT, int gx, int gy, float* dW, float* db, hipStream_t st) {
    static const bool m16 = []() { const char* e = getenv("PASSION_WGRAD_MARCH16"); return !(e && e[0] == '0'); }();
    // measured at 64^3 (tools/conv_microbench.py): 32 -> 16 at batch 4 62.6 us against 75.0 on wgrad_lean_kernel<3, 1>, 16 -> 16 at batch 4
    // 41.8 against 44.3, 16 -> 16 at batch 1 30.0 against 23.4 (a column of 64 planes does not amortise the prologue and the flush):
    // only where the launch has enough (sample, channel block) pairs
    const bool c16 = k.Cout == 16 && m16 && !k.spos && k.B * (k.Cin / 8) >= 8;
    if (!g_wgrad_march || ksize != 3 || stride != 1 || !(k.Cout == 8 || c16) || k.y_f32 || MT != 1) return 1;
    if (!wgrad_lean_covers(k, ksize, stride, MT, gx, gy)) return 1;
    if (k.Hi % WTH || k.Wi % WTW || k.Di < 8) return 1;
    k.Mpad = pad16(k.Cout);
    const int tiles_per_b = k.tilesD * k.tilesH * k.tilesW;
    const int wgs = gx / k.B;
    k.per = (tiles_per_b + wgs - 1) / wgs;
    WMarchP mp;
    mp.wgs_per_b = (tiles_per_b + k.per - 1) / k.per;
    mp.tilesW = k.Wi / WTW;
    mp.ncols = (k.Hi / WTH) * mp.tilesW;
    // D segments: the split that keeps most of the sample's workgroups busy in every round; ties go to the longer march
    int best = 1;
    double best_eff = -1.0;
    for (int ns = 1; ns <= 64; ns *= 2) {
        const int pps = (k.Di + ns - 1) / ns;
        if (ns > 1 && pps < 8) break;
        const long units = (long)mp.ncols * ((k.Di + pps - 1) / pps);
        const long rounds = (units + mp.wgs_per_b - 1) / mp.wgs_per_b;
        const double eff = (double)units / (double)(rounds * mp.wgs_per_b);
        if (eff > best_eff + 0.02) { best_eff = eff; best = ns; }
    }
    mp.planes_per_seg = (k.Di + best - 1) / best;
    mp.nseg_d = (k.Di + mp.planes_per_seg - 1) / mp.planes_per_seg;
    if (c16) {
        passion_prof_tag("wgrad_march16_kernel");
        hipLaunchKernelGGL(wgrad_march16_kernel, dim3(gx, gy), dim3(256), 0, st, k, mp, dW, db);
        LAUNCH_RET();
        return 0;
    }
    if (k.spos) {                  // finish-free backward (conv.hip: march_final): the indicator slab rides along
        passion_prof_tag("wgrad_march_kernel<0, 1>");
        hipLaunchKernelGGL((wgrad_march_kernel<WM_IND_DBG, true>), dim3(gx, gy), dim3(256), 0, st, k, mp, dW, db);
        LAUNCH_RET();
        return 0;
    }
    passion_prof_tag("wgrad_march_kernel<0>");
    switch (g_conv_dbg) {          // developer timing builds; 0 is the product
#define WM_CASE(D) case D: hipLaunchKernelGGL((wgrad_march_kernel<D, false>), dim3(gx, gy), dim3(256), 0, st, k, mp, dW, db); break;
        WM_CASE(1) WM_CASE(2) WM_CASE(8) WM_CASE(16) WM_CASE(32) WM_CASE(33) WM_CASE(3) WM_CASE(34)
#undef WM_CASE
        default: hipLaunchKernelGGL((wgrad_march_kernel<0, false>), dim3(gx, gy), dim3(256), 0, st, k, mp, dW, db);
    }
    LAUNCH_RET();
    return 0;
}

}  // namespace pconv
