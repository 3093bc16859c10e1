// First encoder layer on the matrix cores (reference nn.Conv3d(1, 8, 3, padding_mode='reflect'), code/models/mmformer.py:28): the weight
// gradient of the Cin = 1 conv as a GEMM over voxels.
//
// The vector kernel it replaces (wgrad_c1_kernel, conv_wgrad_lean.hip) keeps 9 x 8 partial sums per thread and spends 216 FMAs per
// voxel: VALU-bound at 10 % of the HBM roofline, 4 x 44 us + 4 folds per step -- one launch per modality, on the largest activation of
// the net.  Here
//   dW[co][tap] = sum_v dY[v][co] * x[v + tap]      D[m = co][n = tap] with K = 32 voxels per v_mfma_f32_16x16x32_bf16
// and every operand fragment is ONE aligned LDS read:
//   * A (dY, k-strided in its voxel-major image [voxel][8 channels]): ds_read_b64_tr_b16, as the other weight-gradient kernels do; the
//     tile comes in by LDS-DMA (16 bytes per lane, lane-linear = the image itself: no registers, no ds_write);
//   * B (x: lane n = tap holds 8 CONSECUTIVE voxels of a W run, shifted by the tap): x has one channel, so a row is 2 bytes per voxel
//     and is staged THREE times, shifted by -1 / 0 / +1 (reflect applied while staging): the W tap picks the copy, the H tap the row,
//     the D tap the plane slot of a 4-slot ring, and the read is a 16-byte aligned ds_read_b128.  Row / copy / slot strides are 1 / 3 /
//     9 sixteen-byte bank slots apart (mod 16), and column n of tile t carries the tap whose slot index th + 3 tw + 9 td is 16 t + n:
//     the sixteen lanes of a read group hit sixteen different slots (one 2-way conflict per group from the g offset).
//   * column 27 (tile 1, n = 11) is all ones: dbias rides in the same MFMAs.
// A workgroup owns 8 rows x the whole width of the (H, W) plane and marches along D; its four waves take two rows each (8 K steps
// per plane at W = 128), keep D in registers (2 tiles x 4 registers) and meet in LDS at the end: one slab slot per workgroup, slots in
// sample order (the grouped launch of the four modality encoders: one launch + one fold for all of them).
// 151 MB per grouped launch: the kernel is a stream of dY (16 bytes per voxel) with 6 MFMA cycles per voxel-wave behind it.
#include <cstdlib>
#include "conv_common.h"
#include "prof.h"

namespace pconv { int g_c1_mfma = 1; }
using namespace pconv;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int TH1 = 8;                    // rows per workgroup
constexpr int XROWS = TH1 + 2;            // staged rows of x (halo 1)

__device__ __forceinline__ bf16x8 tr_read8c(const char* p0, const char* p1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p1));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

struct C1MP {
    int nseg_d, planes_per_seg;   // D segments per (sample, row tile) and planes per segment
    int tilesH;                   // row tiles per sample
    int per_b;                    // slab slots per sample (>= tilesH * nseg_d: the spare ones are flushed with zeros)
};

// strides of the x image (bytes): a row holds W elements + one 16-byte slot, a copy XROWS rows + 9 slots, a plane slot 3 copies
template <int W> struct XGeo {
    static constexpr int RS = W * 2 + 16;
    static constexpr int COPY = XROWS * RS + ((3 - (XROWS * RS / 16) % 16 + 16) % 16) * 16;       // = 3 slots (mod 16)
    static constexpr int SLOT = 3 * COPY;                                                          // = 9 slots (mod 16)
    static constexpr int RING = 4 * SLOT;
    static constexpr int DYT = TH1 * W * 16;                  // one dY tile
    static constexpr int LDS = RING + 2 * DYT + 64;           // (+ 64: the transposed read of the last voxel reaches 16 bytes past it)
    static_assert((COPY / 16) % 16 == 3, "bank-slot strides");
};

template <int W>
__global__ __launch_bounds__(256) void wgrad_c1m_kernel(ConvK p, float* __restrict__ dW, float* __restrict__ dbias, C1MP mp) {
    wgrad_clear(p);
    using G = XGeo<W>;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* xr = lds;                                   // [4 plane slots][3 copies][XROWS rows][W]
    char* ldy = lds + G::RING;                        // [2][TH1 rows][W][8 channels]
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4, wv = tid >> 6;
    const int q = n >> 2, pq = n & 3;
    const int D = p.Di, H = p.Hi;
    const SegD& sd = p.seg[0];

    // ---- which slot / tile
    const int slot = blockIdx.x;
    const int b = slot / mp.per_b, r = slot % mp.per_b;
    float* out = dW + (long)slot * p.wslot;
    float* outb = dbias ? dbias + (long)slot * p.wslot : nullptr;
    if (r >= mp.tilesH * mp.nseg_d) {                 // a spare slot of the slab: zeros
        if (tid < 216) out[tid] = 0.f;
        if (outb && tid < 8) outb[tid] = 0.f;
        return;
    }
    const int th_i = r % mp.tilesH, dseg = r / mp.tilesH;
    const int h0 = th_i * TH1;
    const int d0 = dseg * mp.planes_per_seg, d1 = min(d0 + mp.planes_per_seg, D);
    const bool present = !sd.mask || sd.mask[(long)b * sd.mask_ld + sd.mask_col] != 0;      // absent modality: x reads as zero
    const unsigned short* xb = reinterpret_cast<const unsigned short*>(sd.x) + (long)sd.bsrc(b) * sd.bstride;
    const char* dyb = reinterpret_cast<const char*>(p.dy) + (long)b * D * H * W * 16;

    // ---- staging of one x plane: 16 lanes per row (8 elements each), rows h0 - 1 .. h0 + TH1 (reflected); the two shifted copies
    // take their edge element from the neighbour lane (the row's ends from the row itself: reflect)
    constexpr int CPR = W / 8;                        // 16-byte chunks per row
    const int xrow = tid / CPR, xc = tid % CPR;       // this thread's (row, chunk) when tid < XROWS * CPR
    const bool xon = tid < XROWS * CPR;
    const int xh = min(max(reflect_idx(h0 - 1 + min(xrow, XROWS - 1), H), 0), H - 1);
    auto xfetch = [&](int pl) -> u32x4 {              // plane index (unreflected)
        const int pd = min(max(reflect_idx(pl, D), 0), D - 1);
        const u32x4 v = *reinterpret_cast<const u32x4 __attribute__((address_space(1)))*>(
            reinterpret_cast<unsigned long long>(xb + ((long)pd * H + xh) * W + xc * 8));
        return (xon && present) ? v : (u32x4){0u, 0u, 0u, 0u};
    };
    auto xcommit = [&](int j, u32x4 v) {              // into ring slot j & 3
        // element e of this chunk = x[8 xc + e]; copy s holds x[i + s - 1] at index i
        unsigned up = __shfl_up(v[3], 1, 64) >> 16;                       // previous chunk's element 7
        unsigned dn = __shfl_down(v[0], 1, 64) & 0xffffu;                 // next chunk's element 0
        if (xc == 0) up = v[0] >> 16;                                     // x[-1] = x[1]
        if (xc == CPR - 1) dn = v[3] & 0xffffu;                           // x[W] = x[W - 2]
        u32x4 c0, c2;                                                     // c0[i] = x[i - 1], c2[i] = x[i + 1]
        c0[0] = up | (v[0] << 16);
        c0[1] = (v[0] >> 16) | (v[1] << 16);
        c0[2] = (v[1] >> 16) | (v[2] << 16);
        c0[3] = (v[2] >> 16) | (v[3] << 16);
        c2[0] = (v[0] >> 16) | (v[1] << 16);
        c2[1] = (v[1] >> 16) | (v[2] << 16);
        c2[2] = (v[2] >> 16) | (v[3] << 16);
        c2[3] = (v[3] >> 16) | (dn << 16);
        if (xon) {
            char* base = xr + (j & 3) * G::SLOT + xrow * G::RS + xc * 16;
            *reinterpret_cast<u32x4*>(base) = c0;
            *reinterpret_cast<u32x4*>(base + G::COPY) = v;
            *reinterpret_cast<u32x4*>(base + 2 * G::COPY) = c2;
        }
    };
    // ---- one dY tile (TH1 rows x W voxels x 16 bytes, contiguous in memory) by LDS-DMA: 16 bytes per lane, 1 KB per wave instruction
    constexpr int DYI = G::DYT / 4096;
    auto dyfetch = [&](int d, int buf) {
        const char* src = dyb + ((long)d * H + h0) * W * 16 + tid * 16;
        char* dst = ldy + buf * G::DYT + wv * 1024;
#pragma unroll
        for (int i = 0; i < DYI; ++i)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const void __attribute__((address_space(1)))*>(reinterpret_cast<unsigned long long>(src + i * 4096)),
                                             (void __attribute__((address_space(3)))*)(dst + i * 4096), 16, 0, 0);
    };

    // ---- this lane's B addresses: column n of tile t <-> slot index 16 t + n = th + 3 tw + 9 td (27: the ones column, > 27: nothing)
    int boff[2], btd[2];
    bool bones[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int idx = 16 * t + n, ic = min(idx, 26);
        const int td = ic / 9, tw = (ic % 9) / 3, th = ic % 3;
        btd[t] = td;
        boff[t] = tw * G::COPY + th * G::RS + g * 16;
        bones[t] = idx == 27;
    }
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
    const int aoff = (8 * g + q) * 16 + pq * 8;       // + K-step base; second half 4 voxels on

    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};

    // ---- prologue: planes d0 - 1, d0 (ring slots 0, 1), plane d0 + 1 in registers, dY tile of plane d0 on its way
    xcommit(0, xfetch(d0 - 1));
    xcommit(1, xfetch(d0));
    u32x4 xn = xfetch(d0 + 1);
    dyfetch(d0, 0);
    for (int d = d0; d < d1; ++d) {
        const int j = d - d0;                         // planes d - 1, d, d + 1 = ring slots j, j + 1, j + 2
        xcommit(j + 2, xn);                           // (slot (j + 2) & 3 held plane d - 2: last read in iteration d - 1, behind that iteration's closing barrier)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of the dY tile have landed
        __syncthreads();
        if (d + 1 < d1) {
            xn = xfetch(d + 2);
            dyfetch(d + 1, (j + 1) & 1);              // (buffer (j + 1) & 1 was read in iteration d - 1)
        }
        const char* ay = ldy + (j & 1) * G::DYT + aoff;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int hl = 2 * wv + rr;               // local output row
#pragma unroll
            for (int ks = 0; ks < W / 32; ++ks) {
                const char* ap = ay + (hl * W + ks * 32) * 16;
                const bf16x8 a = tr_read8c(ap, ap + 4 * 16);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const char* bp = xr + ((j + btd[t]) & 3) * G::SLOT + boff[t] + hl * G::RS + ks * 64;
                    bf16x8 bv = *reinterpret_cast<const bf16x8*>(bp);
                    if (bones[t]) bv = ones;
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bv, acc[t], 0, 0, 0);
                }
            }
        }
        // (no closing barrier: the next iteration writes ring slot (j + 3) & 3 and dY buffer (j + 1) & 1 -- read last in iteration
        // d - 1, which every wave left before it passed THIS iteration's barrier)
    }
    __syncthreads();                                  // the last plane's reads done: the ring becomes the reduction buffer

    // ---- the four waves' partial sums meet in LDS (fixed order); lane (n, g) holds rows 4 g .. 4 g + 3 (= co for g < 2) of column n
    float* red = reinterpret_cast<float*>(lds);       // [4 waves][2 tiles][64 lanes][4]
#pragma unroll
    for (int t = 0; t < 2; ++t)
        *reinterpret_cast<f32x4*>(red + ((wv * 2 + t) * 64 + lane) * 4) = acc[t];
    __syncthreads();
    if (tid < 128) {
        const int t = tid >> 6, l = tid & 63, nn = l & 15, gg = l >> 4;
        f32x4 s = *reinterpret_cast<const f32x4*>(red + ((0 * 2 + t) * 64 + l) * 4);
#pragma unroll
        for (int w2 = 1; w2 < 4; ++w2) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(red + ((w2 * 2 + t) * 64 + l) * 4);
            s += v;
        }
        const int idx = 16 * t + nn;
        if (gg < 2) {
            if (idx < 27) {
                const int td = idx / 9, tw = (idx % 9) / 3, th = idx % 3, tap = td * 9 + th * 3 + tw;
#pragma unroll
                for (int rj = 0; rj < 4; ++rj) out[(4 * gg + rj) * 27 + tap] = s[rj];
            } else if (idx == 27 && outb) {
#pragma unroll
                for (int rj = 0; rj < 4; ++rj) outb[4 * gg + rj] = s[rj];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Forward of the same layer on the matrix cores (the vector kernel conv_c1_fwd.hip: 216 FMAs per voxel, 102 us per grouped launch).
//   y[v][co] = sum_tap W[co][tap] x[v + tap] + bias[co]
// with the trick of conv_pair.hip: the 16 MFMA rows carry TWO output planes (rows 0..7: plane d, rows 8..15: plane d + 1), so the K axis
// runs over the FOUR input planes d - 1 .. d + 2: K = 16 (plane, row tap) slots x 4 W offsets (-1 .. +2; the fourth and slots 12..15 carry
// zero weights) = 64 = two v_mfma_f32_16x16x32_bf16 per 16 voxels x 2 planes.  Lane (voxel n, group g) of MFMA j holds K slots 8 j + 2 g and
// + 1: two runs of four consecutive W positions starting at w - 1.  x is staged FOUR times, shifted by -1 .. +2, so that the run of
// voxel n starts on an 8-byte boundary of copy n & 3: every fragment half is one aligned ds_read_b64.  A workgroup owns 8 rows x the
// whole width and marches along D two planes at a time (ring of 6 plane slots); the weights (2 fragments) live in registers, the bias
// is the accumulators' initial value, the epilogue rounds, stores 8 bytes per lane (32 lanes = 256 contiguous bytes of one plane) and
// sums the InstanceNorm statistics of what it stored (slot / ticket fold of conv_common.h, as conv_c1_fwd.hip).
typedef unsigned int u32x2c __attribute__((ext_vector_type(2)));

template <int W> struct FGeo {
    static constexpr int RS = W * 2 + 16;
    static constexpr int COPY = XROWS * RS + 32;
    static constexpr int SLOT = 4 * COPY;
    static constexpr int NSL = 6;
    static constexpr int LDS = NSL * SLOT;
};

struct C1FP { int nseg_d, planes_per_seg, tilesH, nwg_b; };

template <int W>
__global__ __launch_bounds__(256) void conv_c1m_fwd_kernel(ConvK p, C1FP mp) {
    using G = FGeo<W>;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    __shared__ float lsum[4 * 4 * 4 * 2 + 16];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4, wv = tid >> 6;
    const int D = p.Di, H = p.Hi;
    const SegD& sd = p.seg[0];
    const int b = blockIdx.y, wq = blockIdx.x;
    const int th_i = wq % mp.tilesH, dseg = wq / mp.tilesH;
    const int h0 = th_i * TH1;
    const int d0 = dseg * mp.planes_per_seg, d1 = min(d0 + mp.planes_per_seg, D);      // even bounds (launcher)
    const bool present = !sd.mask || sd.mask[(long)b * sd.mask_ld + sd.mask_col] != 0;
    const unsigned short* xb = reinterpret_cast<const unsigned short*>(sd.x) + (long)sd.bsrc(b) * sd.bstride;

    // ---- staging of one x plane: as in the weight gradient above, four shifted copies (copy c holds x[i + c - 1] at index i)
    constexpr int CPR = W / 8;
    const int xrow = tid / CPR, xc = tid % CPR;
    const bool xon = tid < XROWS * CPR;
    const int xh = min(max(reflect_idx(h0 - 1 + min(xrow, XROWS - 1), H), 0), H - 1);
    auto xfetch = [&](int pl) -> u32x4 {
        const int pd = min(max(reflect_idx(pl, D), 0), D - 1);
        const u32x4 v = *reinterpret_cast<const u32x4 __attribute__((address_space(1)))*>(
            reinterpret_cast<unsigned long long>(xb + ((long)pd * H + xh) * W + xc * 8));
        return (xon && present) ? v : (u32x4){0u, 0u, 0u, 0u};
    };
    auto xcommit = [&](int slot, u32x4 v) {
        unsigned up = __shfl_up(v[3], 1, 64) >> 16;                       // previous chunk's element 7
        unsigned dn = __shfl_down(v[0], 1, 64);                           // next chunk's elements 0, 1
        if (xc == 0) up = v[0] >> 16;                                     // x[-1] = x[1]
        if (xc == CPR - 1) dn = (v[3] & 0xffffu) | (v[2] & 0xffff0000u);  // x[W] = x[W - 2], x[W + 1] = x[W - 3]
        u32x4 c0, c2, c3;
        c0[0] = up | (v[0] << 16);
        c0[1] = (v[0] >> 16) | (v[1] << 16);
        c0[2] = (v[1] >> 16) | (v[2] << 16);
        c0[3] = (v[2] >> 16) | (v[3] << 16);
        c2[0] = (v[0] >> 16) | (v[1] << 16);
        c2[1] = (v[1] >> 16) | (v[2] << 16);
        c2[2] = (v[2] >> 16) | (v[3] << 16);
        c2[3] = (v[3] >> 16) | (dn << 16);
        c3[0] = v[1]; c3[1] = v[2]; c3[2] = v[3]; c3[3] = dn;
        if (xon) {
            char* base = lds + slot * G::SLOT + xrow * G::RS + xc * 16;
            *reinterpret_cast<u32x4*>(base) = c0;
            *reinterpret_cast<u32x4*>(base + G::COPY) = v;
            *reinterpret_cast<u32x4*>(base + 2 * G::COPY) = c2;
            *reinterpret_cast<u32x4*>(base + 3 * G::COPY) = c3;
        }
    };

    // ---- weights: A[m][k], k = 32 j + 8 g + e: K slot r = 8 j + 2 g + (e >> 2) = 3 pd + th (r < 12), W offset e & 3 (tw; 3: nothing);
    // rows m < 8: channel m of plane d (depth tap pd), rows m >= 8: channel m - 8 of plane d + 1 (depth tap pd - 1)
    bf16x8 wfr[2];
    {
        const bf16_t* wimg = reinterpret_cast<const bf16_t*>(p.wpack) + wvec_of(p, b) * 8;      // forward image [tap][1 block][16 rows][8]
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int r = 8 * j + 2 * g + (e >> 2), tw = e & 3, pd = r / 3, th = r % 3;
                const int td = pd - (n >> 3), co = n & 7;
                const bool ok = r < 12 && tw < 3 && td >= 0 && td <= 2;
                const int tap = min(max(td, 0), 2) * 9 + th * 3 + min(tw, 2);
                const bf16_t wv_ = wimg[((long)tap * 16 + co) * 8];
                wfr[j][e] = ok ? (__bf16)wv_ : (__bf16)0.f;
            }
    }
    // B addresses: lane (n, g) of MFMA j, half q2: K slot r = 8 j + 2 g + q2 (clamped to 11: zero weights beyond), copy n & 3, element
    // index 4 (n >> 2) of the 16-voxel tile
    int boff[2][2], bpd[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
            const int r = min(8 * j + 2 * g + q2, 11), pd = r / 3, th = r % 3;
            bpd[j][q2] = pd;
            boff[j][q2] = (n & 3) * G::COPY + th * G::RS + (n >> 2) * 8;
        }
    const float* bias = bias_of(p, b);
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) bv = load4(bias + 4 * (g & 1));
    const int half = g >> 1, cg = 4 * (g & 1);
    char* yb = reinterpret_cast<char*>(p.y) + (long)b * D * H * W * 16 + cg * 2;
    const bool fstats = p.fst.stats != nullptr;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};

    // ---- prologue: planes d0 - 1 .. d0 + 2 in ring slots 0 .. 3, planes d0 + 3, d0 + 4 in registers
#pragma unroll
    for (int j = 0; j < 4; ++j) xcommit(j, xfetch(d0 - 1 + j));
    u32x4 xn0 = xfetch(d0 + 3), xn1 = xfetch(d0 + 4);
    int base = 0;                                     // ring slot of plane d - 1
    for (int d = d0; d < d1; d += 2) {
        __syncthreads();                              // planes d - 1 .. d + 2 staged; the slots of planes d - 3, d - 2 free
        if (d + 2 < d1) {
            xcommit((base + 4) % G::NSL, xn0);
            xcommit((base + 5) % G::NSL, xn1);
            xn0 = xfetch(d + 5); xn1 = xfetch(d + 6);
        }
        int sl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) sl[j] = ((base + j) % G::NSL) * G::SLOT;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int hl = 2 * wv + rr;
#pragma unroll
            for (int t = 0; t < W / 16; ++t) {
                f32x4 acc = bv;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    union { u32x2c u[2]; bf16x8 v; } bf;
#pragma unroll
                    for (int q2 = 0; q2 < 2; ++q2) {
                        int so = sl[0];
                        if (bpd[j][q2] == 1) so = sl[1];
                        if (bpd[j][q2] == 2) so = sl[2];
                        if (bpd[j][q2] == 3) so = sl[3];
                        bf.u[q2] = *reinterpret_cast<const u32x2c*>(lds + so + boff[j][q2] + hl * G::RS + t * 32);
                    }
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[j], bf.v, acc, 0, 0, 0);
                }
                unsigned pk[2];
                float rb[4];
                round4_bf16(acc, pk, rb);
                const long vox = (((long)(d + half) * H + h0 + hl) * W + t * 16 + n) * 16;
                *reinterpret_cast<u32x2c __attribute__((address_space(1)))*>(reinterpret_cast<unsigned long long>(yb + vox)) = (u32x2c){pk[0], pk[1]};
                if (fstats) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) { s1[c] += rb[c]; s2[c] = __builtin_fmaf(rb[c], rb[c], s2[c]); }
                }
            }
        }
        base = (base + 2) % G::NSL;
    }
    if (!fstats) return;
    // statistics: 16 voxel lanes (shuffles) -> 4 waves x 4 lane groups (LDS, fixed order) -> the workgroup's slot
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float a1 = group16_sum(s1[c]), a2 = group16_sum(s2[c]);
        if (n == 0) {
            lsum[((wv * 4 + g) * 4 + c) * 2 + 0] = a1;
            lsum[((wv * 4 + g) * 4 + c) * 2 + 1] = a2;
        }
    }
    __syncthreads();
    if (tid < 16) {
        const int c = tid >> 1, k2 = tid & 1, gq = c >> 2, j = c & 3;      // channel c = 4 gq + j: lane groups gq (plane d) and gq + 2 (plane d + 1)
        float t = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) t += lsum[((w2 * 4 + gq) * 4 + j) * 2 + k2] + lsum[((w2 * 4 + gq + 2) * 4 + j) * 2 + k2];
        lsum[4 * 4 * 4 * 2 + tid] = t;
    }
    __syncthreads();
    fwd_stats_commit(p.fst, 8, b, wq, 0, 8, lsum + 4 * 4 * 4 * 2, tid);
}

template <int W>
int launch_c1m_fwd(ConvK& k, const C1FP& mp, hipStream_t st) {
    using G = FGeo<W>;
    static bool granted = false;
    if (!granted) {
        HIP_RET(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_c1m_fwd_kernel<W>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
        granted = true;
    }
    passion_prof_tag("conv_c1m_fwd_kernel<%d>", W);
    hipLaunchKernelGGL((conv_c1m_fwd_kernel<W>), dim3(mp.nwg_b, k.B), dim3(256), G::LDS, st, k, mp);
    LAUNCH_RET();
    return 0;
}

template <int W>
int launch_c1m(ConvK& k, const C1MP& mp, int gx, float* dW, float* db, hipStream_t st) {
    using G = XGeo<W>;
    static bool granted = false;
    if (!granted) {
        HIP_RET(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_c1m_kernel<W>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
        granted = true;
    }
    passion_prof_tag("wgrad_c1m_kernel<%d>", W);
    hipLaunchKernelGGL((wgrad_c1m_kernel<W>), dim3(gx), dim3(256), G::LDS, st, k, dW, db, mp);
    LAUNCH_RET();
    return 0;
}

}  // namespace

namespace pconv {

// Cin = 1, Cout = 8, 3x3x3, stride 1, reflect padding, bf16 x and dY, no InstanceNorm in the loader, W in {32, 64, 128}, H % 8 == 0.
// gx = slots of the partial slab, a multiple of B (slots in sample order: a grouped launch's weight sets are runs of samples).
// 0 = launched, 1 = not this shape (the vector kernel takes it), < 0 = HIP error
bool wgrad_c1m_covers(const ConvK& k, int ksize, int stride, int gx) {
    if (!g_c1_mfma || ksize != 3 || stride != 1 || k.nseg != 1 || k.Cin != 1 || k.seg[0].C != 1 || k.Cout != 8 || k.seg[0].stats || k.y_f32) return false;
    if (k.pad_mode != PASSION_PAD_REFLECT || k.Hi % TH1 || (k.Wi != 32 && k.Wi != 64 && k.Wi != 128) || k.Di < 2 || k.Hi < 2) return false;
    if (gx % k.B || gx / k.B < k.Hi / TH1) return false;
    return (long)k.Di * k.Hi * k.Wi * 16 < (1L << 31);
}

int launch_wgrad_c1m(ConvK& k, int ksize, int stride, int gx, float* dW, float* db, hipStream_t st) {
    if (!wgrad_c1m_covers(k, ksize, stride, gx) || !dW) return 1;
    C1MP mp;
    mp.per_b = gx / k.B;
    mp.tilesH = k.Hi / TH1;
    // D segments: as many as the sample's slots allow, at least 8 planes each (two halo planes of x per segment)
    int nseg = (int)hmin(mp.per_b / mp.tilesH, hmax(1, k.Di / 8));
    mp.planes_per_seg = (k.Di + nseg - 1) / nseg;
    mp.nseg_d = (k.Di + mp.planes_per_seg - 1) / mp.planes_per_seg;
    if (k.Wi == 128) return launch_c1m<128>(k, mp, gx, dW, db, st);
    if (k.Wi == 64) return launch_c1m<64>(k, mp, gx, dW, db, st);
    return launch_c1m<32>(k, mp, gx, dW, db, st);
}

// forward: the same shapes, an even depth.  0 = launched, 1 = not this shape (conv_c1_fwd.hip takes it), < 0 = HIP error
int launch_conv_c1m_fwd(ConvK& k, int ksize, int stride, const void* packed_bf16, hipStream_t st) {
    if (!g_c1_mfma || ksize != 3 || stride != 1 || k.nseg != 1 || k.Cin != 1 || k.seg[0].C != 1 || k.Cout != 8) return 1;
    if (k.seg[0].stats || k.residual || k.y_f32 || k.seg[0].bmod > 0 || k.pad_mode != PASSION_PAD_REFLECT) return 1;
    if (k.Hi % TH1 || (k.Wi != 32 && k.Wi != 64 && k.Wi != 128) || k.Di % 2 || k.Di < 4 || k.Hi < 2) return 1;
    if ((long)k.Di * k.Hi * k.Wi * 16 >= (1L << 31)) return 1;
    C1FP mp;
    mp.tilesH = k.Hi / TH1;
    // D segments: ~1024 workgroups per launch, at least 8 planes each, an even number of planes, at most 1024 statistics slots per sample
    int nseg = (int)hmax(1, hmin((1024 + (long)k.B * mp.tilesH - 1) / ((long)k.B * mp.tilesH), k.Di / 8));
    nseg = (int)hmin(nseg, hmax(1, 1024 / mp.tilesH));
    mp.planes_per_seg = ((k.Di + nseg - 1) / nseg + 1) & ~1;
    mp.nseg_d = (k.Di + mp.planes_per_seg - 1) / mp.planes_per_seg;
    mp.nwg_b = mp.tilesH * mp.nseg_d;
    k.Mpad = pad16(k.Cout);
    k.wpack = packed_bf16;
    if (k.fst_want) {
        const int rc = fwd_stats_setup(k, mp.nwg_b, 1, st);
        if (rc) return rc;
    }
    if (k.Wi == 128) return launch_c1m_fwd<128>(k, mp, st);
    if (k.Wi == 64) return launch_c1m_fwd<64>(k, mp, st);
    return launch_c1m_fwd<32>(k, mp, st);
}

}  // namespace pconv
