// Lean persistent bf16 implicit-GEMM conv (forward and dgrad of the stride-1 convs, every tensor a multiple of
// 8 channels wide) -- same math, operand layouts and epilogues as conv_bf16.hip, restructured around what the
// counters said about that kernel on MI355X (level-1 8->8 conv, 128^3): it is VALU/SALU-issue bound, not memory
// bound: per wavefront and brick 895 VALU + 508 SALU instructions against 28 MFMAs, 302 + 328 of them before the
// first load is even issued (decoding the launch descriptor, the virtual-concat lookup, masks, weight staging,
// index arithmetic with runtime divisors), 366 in the brick staging, 138 in the epilogue.  So here:
//   * a workgroup is persistent over `per` consecutive bricks of ONE sample: descriptor decoding, the concat
//     lookup, masks and InstanceNorm statistics are resolved once into a small LDS table (BlkDesc per 8-channel
//     block), weights stay in LDS, brick coordinates advance incrementally (no runtime divisions per brick);
//   * everything that does not depend on the brick is a per-lane constant computed once: source voxel offsets of
//     the staging threads, LDS tap offsets of the MFMA lanes, output voxel offsets of the epilogue lanes;
//   * interior bricks (no padding involved) take a branch-free addressing path: address = brick base + constant;
//   * the next step's brick is fetched into registers while the matrix cores work on the current one.
#include <cstdlib>
#include "conv_common.h"
#include "prof.h"

namespace pconv { extern int g_conv_target, g_conv_dbg; }
using namespace pconv;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr bool LDS_ROW_PAD = false;   // see GeoL::RS: superseded by the row pairing of the N tiles (rows h and h + 4)
template <int KS, int CKB> struct GeoL {
    static constexpr int HALO = (KS == 3) ? 1 : 0;
    static constexpr int ID = TD + 2 * HALO, IH = TH + 2 * HALO, IW = TW + 2 * HALO;
    static constexpr int VOL = ID * IH * IW;
    static constexpr int NC8 = CKB / 8;
    static constexpr int VS = CKB * 2 + (CKB > 8 ? 16 : 0);   // bytes per voxel in LDS (padded: spreads 16-B slots over banks)
    static constexpr int NTAP = KS * KS * KS;
    static constexpr int Q = NTAP * NC8;                      // (tap, 8-channel block) pairs per chunk
    static constexpr int NSTEP = (Q + 3) / 4;                 // MFMA K-steps per chunk (4 pairs = K 32 each)
    static constexpr int TOT = VOL * NC8, NIT = (TOT + 255) / 256;
    // LDS layout of the staged brick: voxel (zd, zh, zw) at zd * PS + zh * RS + zw * VS.  A B-fragment read covers, per 16
    // lanes, 8 voxels of row h and 8 of row h + 1: with 8-channel voxels (16 B) and dense rows (160 B) the two runs overlap
    // in two of the sixteen 16-byte bank groups; a row stride of 384 B puts the second run exactly 128 B (mod 256) further.
    static constexpr int RS = (KS == 3 && CKB == 8 && LDS_ROW_PAD) ? 384 : IW * VS;
    static constexpr int PS = IH * RS;
    static constexpr int BRICK = (ID * PS + 15) / 16 * 16;
};

struct alignas(16) BlkDesc {     // one 8-channel block of the K axis, resolved for this workgroup's sample
    const bf16_t* src;           // sample base + channel offset
    int C, nch, has_norm, pad_;  // voxel stride (elements); 8 or 0 (absent modality); InstanceNorm + LeakyReLU in the loader
    float shift[8], rstd[8];     // normalised value = x * rstd + shift  (shift = -mean * rstd: one FMA per element)
};
static_assert(sizeof(BlkDesc) == 96, "BlkDesc layout");
struct alignas(16) EpiDesc {     // dgrad epilogue: the four output rows (= forward input channels) of one lane group
    const bf16_t* x;             // forward input of those channels (sample base + channel), for x_hat
    bf16_t* dx;
    int C, ok, has_stats;
    float mval;                  // modality mask of the segment for this sample
    float mean[4], rstd[4];
};
static_assert(sizeof(EpiDesc) == 64, "EpiDesc layout");
constexpr int MAXBLK = 64;       // K <= 512 channels in 32-channel chunks (the chunk mask has 32 bits: 8-channel chunks stop at 256)
constexpr int DESC_BYTES = MAXBLK * (int)sizeof(BlkDesc);

// Values that are the same in every lane but arrive through LDS or per-thread arithmetic look divergent to the compiler:
// branches on them become exec-masked regions (with copies of every live accumulator) and their address arithmetic runs
// on the vector ALU.  readfirstlane moves them to scalar registers.
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T> __device__ __forceinline__ T* rfl_ptr(T* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = (unsigned)rfl((int)(unsigned)v), hi = (unsigned)rfl((int)(unsigned)(v >> 32));
    return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}

// Pointers that reached a lane through the LDS descriptor tables are generic to the compiler: their loads become flat_load, and
// one pending flat access turns every later wait into vmcnt(0) (flat returns out of order) -- the brick prefetch was drained at
// its first use.  They are global pointers: say so.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ T gload(const void* ptr) {
    return *reinterpret_cast<const T __attribute__((address_space(1)))*>(reinterpret_cast<unsigned long long>(ptr));
}
__device__ __forceinline__ uint2 gload2(const void* ptr) {
    const u32x2 v = gload<u32x2>(ptr);
    return make_uint2(v.x, v.y);
}
__device__ __forceinline__ void gstore4(bf16_t* ptr, f32x4 v) {
    union { bf16_t h[4]; u32x2 u; } t;
    t.h[0] = (bf16_t)v[0]; t.h[1] = (bf16_t)v[1]; t.h[2] = (bf16_t)v[2]; t.h[3] = (bf16_t)v[3];
    *reinterpret_cast<u32x2 __attribute__((address_space(1)))*>(reinterpret_cast<unsigned long long>(ptr)) = t.u;
}

__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (__bf16)0.f;
    return z;
}

// S: stride of the forward conv; only MODE 2 (dgrad over the padded domain) runs with S = 2: dY is staged zero-stuffed
// (position u of the fine grid holds dy[u / 2] for even u, zero otherwise) and the 27 flipped taps gather as for S = 1.
template <int KS, int MT, int MODE, int CKB, int S = 1>
__device__ __forceinline__ void conv_lean_body(const ConvK& p) {
    static_assert(S == 1 || MODE == 2, "stride 2 only for the padded-domain dgrad");
    constexpr bool DG = (MODE >= 1);
    // MODE 2: dgrad of a reflect-padded conv computed over the PADDED input domain (output coordinate op <-> input
    // position op - 1, from -1 to S): every brick costs the same 27 taps, no mirror-tap bookkeeping at all; the halo
    // planes are folded onto planes 1 / S-2 by the InstanceNorm-backward finish kernel, which reads dx anyway.
    // (Mirror taps made every brick of the small deep levels a 'border' brick: 8^3 x 128 channels took 230 us.)
    constexpr bool PADOUT = (MODE == 2);
    using G = GeoL<KS, CKB>;
    constexpr int NIT = G::NIT, NC8 = G::NC8, VS = G::VS;
    constexpr int WV = G::NSTEP * 4 * MT * 16;            // 16-byte weight vectors per chunk
    extern __shared__ __attribute__((aligned(16))) char dyn_lds[];
    char* lds = dyn_lds;                                   // [brick | block descriptors | weight fragments]
    BlkDesc* desc = reinterpret_cast<BlkDesc*>(dyn_lds + G::BRICK);
    bf16x8* wl0 = reinterpret_cast<bf16x8*>(dyn_lds + G::BRICK + DESC_BYTES);
    __shared__ float lsum[4 * MT * 16 * 2 + (DG ? 0 : MT * 16 * 2)];        // [wave][row][2] (+ forward: the workgroup's totals)
    __shared__ EpiDesc edesc[DG ? MT * 4 : 1];
    __shared__ unsigned short mlist[128];
    __shared__ int mcnt[4];
    __shared__ unsigned char mtab[3][5][8];   // [axis d,h,w][option][local coordinate] = tap | valid << 2 | brick position << 3

    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int wv = rfl(tid >> 6);
    const int mg = blockIdx.y;
    // stride 1: the GEMM's output volume (fwd: y, dgrad: dx) and its source volume (fwd: x, dgrad: dy) coincide
    const int Sd = p.Di, Sh = p.Hi, Sw = p.Wi;
    const int Wp = pad8(Sw + 2);      // row pitch of the padded gradient scratch (MODE 2): rows start on 128-byte lines for 8-channel tensors
    const int Qd = (S == 2) ? p.Do : Sd, Qh = (S == 2) ? p.Ho : Sh, Qw = (S == 2) ? p.Wo : Sw;   // source volume (dgrad: dY)
    const int K8 = DG ? (p.Cout + 7) / 8 : p.Cin / 8;
    const int nchunk = (K8 + NC8 - 1) / NC8;
    const bool wres = p.wres != 0;
    const bf16x8* wp0 = reinterpret_cast<const bf16x8*>(p.wpack);

    const int tiles_per_b = p.tilesD * p.tilesH * p.tilesW;
    const int wgs_per_b = (tiles_per_b + p.per - 1) / p.per;
    const int bx = xcd_swizzle(blockIdx.x, gridDim.x);
    const int b = bx / wgs_per_b;
    const bf16x8* wp = wp0 + wvec_of(p, b);                 // grouped launch: this sample's weight set
    // Brick schedule of this workgroup.  Forward: `per` consecutive bricks (measured best: 46 vs 54 us on the level-1
    // conv).  Dgrad: border bricks are several times more expensive (mirror taps), a workgroup that owns a whole border
    // row would set the kernel's duration -- so round i of the sample's bricks is [i*W, (i+1)*W), W = workgroups per
    // sample, and workgroup w takes slot (w + 17 i) mod W of it: within a round neighbouring workgroups (same XCD) work
    // on neighbouring bricks, from round to round a workgroup drifts diagonally through the (h, w) tile grid
    // (level-1 dgrad 118 -> 99 us).
    __shared__ int bricktab[256];
    const int wq = bx % wgs_per_b;
    constexpr bool DIAG = DG && !PADOUT;
    auto brick_of = [&](int i) { return DIAG ? (wq + 17 * i) % wgs_per_b + i * wgs_per_b : wq * p.per + i; };
    if (tid < p.per) {
        const int t = brick_of(tid);
        int code = -1;
        if (t < tiles_per_b) {
            const int tw = t % p.tilesW, th = (t / p.tilesW) % p.tilesH, td = t / (p.tilesW * p.tilesH);
            code = (td << 20) | (th << 10) | tw;
        }
        bricktab[tid] = code;
    }
    // valid bricks form a prefix of the table (dgrad: only the last round can be short)
    const int nbricks = DIAG ? (brick_of(p.per - 1) < tiles_per_b ? p.per : p.per - 1)
                           : max(0, min(p.per, tiles_per_b - wq * p.per));
    if (nbricks <= 0) return;

    // ---- once per workgroup: resolve the K axis into block descriptors
    if (tid < K8) {
        BlkDesc d;
        d.pad_ = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) { d.shift[c] = 0.f; d.rstd[c] = 1.f; }
        if (DG && p.y_f32) {        // f32 dY of a logit head: one block of Cout <= 8 channels, converted while staging
            d.src = reinterpret_cast<const bf16_t*>(reinterpret_cast<const float*>(p.dy) + (long)b * Qd * Qh * Qw * p.Cout);
            d.C = p.Cout; d.nch = 8; d.has_norm = 0;
        } else if (DG) {
            d.src = reinterpret_cast<const bf16_t*>(p.dy) + (long)b * Qd * Qh * Qw * p.Cout + tid * 8;
            d.C = p.Cout; d.nch = 8; d.has_norm = 0;
        } else {
            int r = tid;
            bool found = false;
            d.src = nullptr; d.C = 8; d.nch = 0; d.has_norm = 0;
#pragma unroll
            for (int s2 = 0; s2 < PASSION_MAX_SEG; ++s2) {
                if (s2 < p.nseg && !found) {
                    const SegD& sd = p.seg[s2];
                    const int nb = sd.C / 8;
                    if (r < nb) {
                        found = true;
                        const int bs = sd.bsrc(b), c0 = r * 8;
                        const bool on = !sd.mask || sd.mask[(long)b * sd.mask_ld + sd.mask_col] != 0;
                        d.src = reinterpret_cast<const bf16_t*>(sd.x) + (long)bs * sd.bstride + c0;
                        d.C = sd.C; d.nch = on ? 8 : 0; d.has_norm = sd.stats != nullptr;
                        if (sd.stats) {
#pragma unroll
                            for (int c = 0; c < 8; ++c) {
                                const float mean = sd.stats[((long)bs * sd.C + c0 + c) * 2 + 0];
                                d.rstd[c] = sd.stats[((long)bs * sd.C + c0 + c) * 2 + 1];
                                d.shift[c] = -mean * d.rstd[c];
                            }
                        }
                    }
                    r -= nb;
                }
            }
        }
        desc[tid] = d;
    }
    if (DG && tid >= 64 && tid < 64 + MT * 4) {     // dgrad epilogue descriptors (second wave, concurrently with the above)
        const int e = tid - 64, cg = (mg * MT + e / 4) * 16 + (e % 4) * 4;
        EpiDesc d;
        d.x = nullptr; d.dx = nullptr; d.C = 4; d.ok = 0; d.has_stats = 0; d.mval = 1.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { d.mean[j] = 0.f; d.rstd[j] = 1.f; }
#pragma unroll
        for (int s2 = 0; s2 < PASSION_MAX_SEG; ++s2) {
            if (s2 < p.nseg && cg < p.Cin) {
                const SegD& sd = p.seg[s2];
                if (cg >= sd.cbase && cg < sd.cbase + sd.C && sd.dx != nullptr) {
                    const int cl = cg - sd.cbase, bs = sd.bsrc(b);
                    d.ok = 1; d.C = sd.C; d.has_stats = sd.stats != nullptr;
                    d.x = reinterpret_cast<const bf16_t*>(sd.x) + (long)bs * sd.bstride + cl;
                    d.dx = PADOUT ? reinterpret_cast<bf16_t*>(p.y) + (long)b * (Sd + 2) * (Sh + 2) * pad8(Sw + 2) * p.Cin + cg
                                  : reinterpret_cast<bf16_t*>(sd.dx) + (long)b * sd.bstride + cl;
                    if (sd.mask) d.mval = sd.mask[(long)b * sd.mask_ld + sd.mask_col] ? 1.f : 0.f;
                    if (sd.stats) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            d.mean[j] = sd.stats[((long)bs * sd.C + cl + j) * 2 + 0];
                            d.rstd[j] = sd.stats[((long)bs * sd.C + cl + j) * 2 + 1];
                        }
                    }
                }
            }
        }
        edesc[e] = d;
    }
    // weight fragments of chunk c -> LDS slot dst; padding pairs (q >= Q) and blocks past K8 are zero, so the MFMA
    // loop never predicates (absent modalities are zeros on the ACTIVATION side: staged as zero blocks)
    auto load_weights = [&](int c, bf16x8* dst) {
        for (int v = tid; v < WV; v += 256) {
            const int row = v % (MT * 16), gg = (v / (MT * 16)) % 4, s = v / (MT * 16 * 4);
            const int q = 4 * s + gg, tap = q / NC8, blk = q % NC8;
            const int gb = c * NC8 + blk, rowg = (mg * MT) * 16 + row;
            bool ok = (q < G::Q) && (gb < K8);
            long src = ((long)tap * K8 + gb) * p.Mpad + rowg;
            if (DG && p.bd_rows) {               // block-diagonal: K block gb = (segment, block of the real Cout); see conv.hip
                ok = ok && (rowg / p.bd_rows == gb / p.bd_k8);
                src = ((long)tap * p.bd_k8 + gb % p.bd_k8) * p.Mpad + rowg;
            }
            dst[v] = ok ? wp[src] : zero8();
        }
    };
    if (wres)
        for (int c = 0; c < nchunk; ++c) load_weights(c, wl0 + c * WV);
    __syncthreads();
    unsigned chunk_live = 0u;                     // chunks with at least one present block
    for (int gb = 0; gb < K8; ++gb)
        if (desc[gb].nch > 0) chunk_live |= 1u << (gb / NC8);
    chunk_live = (unsigned)rfl((int)chunk_live);

    // ---- per-lane constants
    const int jj = tid % NC8;
    int voxoff[NIT];                              // source voxel of staging iteration u relative to the brick's first voxel
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
        const int tv = (tid + 256 * u) / NC8;
        const int zw = tv % G::IW, zh = (tv / G::IW) % G::IH, zd = tv / (G::IW * G::IH);
        // S == 2 (zero-stuffed dY): brick origins are even, so the parity of a staged position is the parity of its
        // brick-local coordinate -- a per-thread constant; only the all-even ones carry data (source voxel = half)
        voxoff[u] = (S == 2) ? ((zd >> 1) * Qh + (zh >> 1)) * Qw + (zw >> 1) : (zd * Qh + zh) * Qw + zw;
        if (S == 2 && ((zd | zh | zw) & 1)) voxoff[u] = -1;
    }
    static_assert(G::RS == G::IW * VS, "dense rows: the staged brick is linear in the staging index");
    const int ldsoff0 = (tid / NC8) * VS + jj * 16;       // iteration u writes at ldsoff0 + u * (256 / NC8) * VS
    // N tile nt of a wave = rows nt and nt + 4 of its depth slice, 8 voxels of W each.  Rows four apart (not neighbours):
    // with 8-channel voxels a staged row is 10 x 16 B, so rows h and h + 4 start 640 B = 128 B (mod 256) apart and the two
    // 128-byte runs of a B-fragment read fall into disjoint LDS bank groups (neighbouring rows overlapped in two of sixteen).
    const int lh0 = 4 * (n >> 3), lw = n & 7, ld_ = wv;
    const int lh[4] = {lh0, lh0 + 1, lh0 + 2, lh0 + 3};
    // B fragments: this lane's voxel (row nt) at tap 0, plus its 8-channel block inside a 32-channel voxel
    const int pos0 = ld_ * G::PS + lh0 * G::RS + lw * VS + (NC8 == 4 ? g * 16 : 0);
    int toffv[NC8 == 1 ? G::NSTEP : 1];           // CKB 8: lane group g owns tap 4s+g (per-lane offset); CKB 32: tap s for all
    if (NC8 == 1) {
#pragma unroll
        for (int s = 0; s < G::NSTEP; ++s) {
            const int tap = 4 * s + g;
            int td = 0, th = 0, tw = 0;
            if (KS == 3) { td = tap / 9; th = (tap / 3) % 3; tw = tap % 3; }
            if (DG && KS == 3) { td = 2 - td; th = 2 - th; tw = 2 - tw; }   // gather form: voxel i reads u = i + 1 - t
            toffv[s] = (tap < G::NTAP) ? td * G::PS + th * G::RS + tw * VS : 0;
        }
    }
    const int wlane = (g * MT * 16 + n) * 16;     // A fragment of step s, row tile mt: wl + wlane + s*4*MT*256 + mt*256
    const int lanevox0 = (ld_ * Sh + lh0) * Sw + lw;      // output voxel of row nt: + nt * Sw
    f32x4 biasv[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int co = (mg * MT + mt) * 16 + g * 4;
        biasv[mt] = (!DG && bias_of(p, b) && co < p.Cout) ? load4(bias_of(p, b) + co) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const bool reflect_dg = DG && !PADOUT && (KS == 3) && (p.pad_mode == PASSION_PAD_REFLECT);
    constexpr int SHIFT = PADOUT ? 1 : 0;        // padded output coordinate -> position

    // ---- staging: FETCH (global -> registers) / COMMIT (registers -> normalise -> LDS)
    u32x4 raw[NIT];
    unsigned okbits = 0u;
    // f32 dY (DG && p.y_f32): C = Cout floats per voxel -> 8 bf16 (upper channels zero)
    auto load_vox = [&](const bf16_t* src, long vox, int C) -> u32x4 {
        if (DG && p.y_f32) {
            const float* f = reinterpret_cast<const float*>(src) + vox * C;
            const f32x4 lo = gload<f32x4>(f);
            f32x4 hi = {0.f, 0.f, 0.f, 0.f};
            if (C == 8) hi = gload<f32x4>(f + 4);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) { o[j] = (__bf16)lo[j]; o[4 + j] = (__bf16)hi[j]; }
            return __builtin_bit_cast(u32x4, o);
        }
        return gload<u32x4>(src + vox * C);
    };
    auto fetch = [&](int chunk, int o0d, int o0h, int o0w) {
        const BlkDesc* dp = desc + chunk * NC8 + jj;
        const bf16_t* src = dp->src;
        int C = dp->C;
        bool have = dp->nch > 0 && chunk * NC8 + jj < K8;
        if (NC8 == 1) {                               // one block per chunk: the descriptor is the same for every thread
            src = rfl_ptr(src); C = rfl(C); have = rfl(have ? 1 : 0) != 0;
        }
        okbits = 0u;
        const bool interior = o0d >= G::HALO && o0h >= G::HALO && o0w >= G::HALO && o0d + TD + G::HALO <= S * Qd &&
                              o0h + TH + G::HALO <= S * Qh && o0w + TW + G::HALO <= S * Qw;
        if (interior && !(DG && p.y_f32)) {       // workgroup-uniform: no padding anywhere in this brick
            // 32-bit element offsets from the sample base (a sample has < 2^31 elements; voxoff and C are < 2^24):
            // one 24-bit multiply and one add per load instead of a 64-bit multiply-add chain
            const int basevox = (((o0d - G::HALO) / S) * Qh + (o0h - G::HALO) / S) * Qw + (o0w - G::HALO) / S;
            const unsigned base_el = (unsigned)basevox * (unsigned)C;
#pragma unroll
            for (int u = 0; u < NIT; ++u) {
                raw[u] = (u32x4){0u, 0u, 0u, 0u};
                if (have && (tid + 256 * u < G::TOT) && (S == 1 || voxoff[u] >= 0)) {
                    const unsigned el = base_el + __umul24((unsigned)voxoff[u], (unsigned)C);
                    raw[u] = gload<u32x4>(src + el);
                    okbits |= 1u << u;
                }
            }
        } else if (interior) {
            const int basevox = (((o0d - G::HALO) / S) * Qh + (o0h - G::HALO) / S) * Qw + (o0w - G::HALO) / S;
#pragma unroll
            for (int u = 0; u < NIT; ++u) {
                raw[u] = (u32x4){0u, 0u, 0u, 0u};
                if (have && (tid + 256 * u < G::TOT) && (S == 1 || voxoff[u] >= 0)) {
                    raw[u] = load_vox(src, (long)(basevox + voxoff[u]), C);
                    okbits |= 1u << u;
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < NIT; ++u) {
                const int tv = (tid + 256 * u) / NC8;
                const int zw = tv % G::IW, zh = (tv / G::IW) % G::IH, zd = tv / (G::IW * G::IH);
                int sd = o0d - G::HALO + zd, sh = o0h - G::HALO + zh, sw = o0w - G::HALO + zw;
                bool ok = have && (tid + 256 * u < G::TOT);
                if (!DG && p.pad_mode == PASSION_PAD_REFLECT) {
                    sd = reflect_idx(sd, Sd); sh = reflect_idx(sh, Sh); sw = reflect_idx(sw, Sw);
                    sd = min(max(sd, 0), Sd - 1); sh = min(max(sh, 0), Sh - 1); sw = min(max(sw, 0), Sw - 1);
                } else {
                    if (S == 2) {                             // zero-stuffed dY: only even fine-grid positions carry data
                        ok = ok && sd >= 0 && sh >= 0 && sw >= 0 && !((sd | sh | sw) & 1);
                        sd >>= 1; sh >>= 1; sw >>= 1;
                    }
                    ok = ok && (sd >= 0 && sd < Qd && sh >= 0 && sh < Qh && sw >= 0 && sw < Qw);
                }
                raw[u] = (u32x4){0u, 0u, 0u, 0u};
                if (ok) {
                    raw[u] = load_vox(src, ((long)sd * Qh + sh) * Qw + sw, C);
                    okbits |= 1u << u;
                }
            }
        }
    };
    auto commit = [&](int chunk) {
        const BlkDesc* dp = desc + chunk * NC8 + jj;
        const bool hn = (NC8 == 1) ? rfl(dp->has_norm) != 0 : dp->has_norm != 0;
        const bool norm = hn && okbits != 0u;
        // channel pairs as 2-vectors: v_pk_fma_f32 / v_pk_mul_f32 work on two channels per instruction
        f32x2 sc[4], sh[4];
        if (norm) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                sc[c] = (f32x2){dp->rstd[2 * c], dp->rstd[2 * c + 1]};
                sh[c] = (f32x2){dp->shift[2 * c], dp->shift[2 * c + 1]};
            }
        }
        const f32x2 slope2 = {p.slope, p.slope};
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            if (tid + 256 * u >= G::TOT) continue;
            u32x4 o = raw[u];
            if (norm && ((okbits >> u) & 1u)) {
                bf16x8 ob;
#pragma unroll
                for (int c = 0; c < 4; ++c) {      // bf16 pairs -> f32: low half = even channel
                    const f32x2 v = {__uint_as_float(raw[u][c] << 16), __uint_as_float(raw[u][c] & 0xffff0000u)};
                    f32x2 t = __builtin_elementwise_fma(v, sc[c], sh[c]);
                    const f32x2 ts = t * slope2;
                    t.x = fmaxf(t.x, ts.x);               // LeakyReLU for 0 <= slope <= 1: one max instead of compare + select
                    t.y = fmaxf(t.y, ts.y);
                    ob[2 * c] = (__bf16)t.x; ob[2 * c + 1] = (__bf16)t.y;
                }
                o = __builtin_bit_cast(u32x4, ob);
            }
            *reinterpret_cast<u32x4*>(lds + ldsoff0 + u * (256 / NC8) * VS) = o;
        }
    };

    // ---- walk the flattened (brick, chunk) steps
    int brick_i = 0;
    const int code0 = rfl(bricktab[0]);
    int tw_ = code0 & 1023, th_ = (code0 >> 10) & 1023, td_ = code0 >> 20;
    int chunk = 0;
    const int nsteps = nbricks * nchunk;
    const bool nostage = (p.dbg & 1) != 0;
    if (((chunk_live >> 0) & 1u) && !nostage) fetch(0, td_ * TD - SHIFT, th_ * TH - SHIFT, tw_ * TW - SHIFT);
    f32x4 acc[MT][4];
    int nmirror = 0;
    // dgrad: InstanceNorm-backward sums (sum g, sum g * x_hat) stay in registers over all bricks of the workgroup
    // forward with fused statistics (p.fst.stats): (sum y, sum y^2) of the stored bf16 values, likewise
    float s1[MT][4], s2v[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) { s1[mt][j] = 0.f; s2v[mt][j] = 0.f; }
    const bool fstats = !DG && p.fst.stats != nullptr;     // the launcher grants it only when every brick is whole
    uint2 xpre[4];                                // MT == 1: x of the four output voxels, fetched ahead of the MFMA loop

    for (int step = 0; step < nsteps; ++step) {
        const int o0d = td_ * TD, o0h = th_ * TH, o0w = tw_ * TW;
        const int cb8 = chunk * NC8;
        const bf16x8* wl = wres ? wl0 + chunk * WV : wl0;
        const bool live = (chunk_live >> chunk) & 1u;
        __syncthreads();                              // everybody is done reading the previous step's brick / tables
        if (chunk == 0) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            nmirror = 0;
            if (reflect_dg) {                         // mirror-tap bookkeeping of this brick (adjoint of reflect padding)
                const bool lowD = (o0d == 0) && (Sd >= 2), lowH = (o0h == 0) && (Sh >= 2), lowW = (o0w == 0) && (Sw >= 2);
                const bool hiD = p.hi_t[0] >= 0 && (o0d <= Sd - 2) && (Sd - 2 < o0d + TD);
                const bool hiH = p.hi_t[1] >= 0 && (o0h <= Sh - 2) && (Sh - 2 < o0h + TH);
                const bool hiW = p.hi_t[2] >= 0 && (o0w <= Sw - 2) && (Sw - 2 < o0w + TW);
                if (lowD || lowH || lowW || hiD || hiH || hiW) {       // workgroup-uniform
                    if (tid < 3 * 5 * 8) {
                        const int ax = tid / 40, a = (tid / 8) % 5, il = tid % 8;
                        const int i0 = ax == 0 ? o0d : (ax == 1 ? o0h : o0w), Sn = ax == 0 ? Sd : (ax == 1 ? Sh : Sw);
                        int t, pos2; bool v;
                        axis_tap<KS, G::HALO, DG, 1>(a, il, i0, Sn, p.hi_t[ax], p.hi_u[ax], t, pos2, v);
                        mtab[ax][a][il] = (unsigned char)((t & 3) | (v ? 4 : 0) | ((pos2 & 31) << 3));
                    }
                    bool f = false;
                    int code = 0;
                    if (tid < 125) {
                        const int ad = tid / 25, ah = (tid / 5) % 5, aw = tid % 5;
                        f = !(ad < 3 && ah < 3 && aw < 3) && (ad != 3 || lowD) && (ad != 4 || hiD) && (ah != 3 || lowH) &&
                            (ah != 4 || hiH) && (aw != 3 || lowW) && (aw != 4 || hiW);
                        code = ad | (ah << 3) | (aw << 6);
                    }
                    const unsigned long long bal = __ballot(f);
                    const int prefix = __popcll(bal & ((1ull << lane) - 1ull));
                    if (lane == 0 && wv < 2) mcnt[wv] = __popcll(bal);
                    __syncthreads();
                    if (f) mlist[(wv == 1 ? mcnt[0] : 0) + prefix] = (unsigned short)code;
                    __syncthreads();
                    nmirror = mcnt[0] + mcnt[1];
                }
            }
        }
        if (live && !nostage) commit(chunk);
        if (!wres && live) load_weights(chunk, wl0);
        __syncthreads();

        // ---- next step: advance the brick coordinates incrementally, get its loads in flight
        int n_chunk = chunk + 1, n_tw = tw_, n_th = th_, n_td = td_;
        if (n_chunk == nchunk) {
            n_chunk = 0;
            if (++brick_i < nbricks) {
                const int code = rfl(bricktab[brick_i]);
                n_tw = code & 1023; n_th = (code >> 10) & 1023; n_td = code >> 20;
            }
        }
        if (step + 1 < nsteps && ((chunk_live >> n_chunk) & 1u) && !nostage)
            fetch(n_chunk, n_td * TD - SHIFT, n_th * TH - SHIFT, n_tw * TW - SHIFT);

        // dgrad epilogue geometry of this lane's four output voxels, once per brick: voxel index in x (x_hat source;
        // PADOUT: the reflected position) and in the dx target (PADOUT: the padded scratch volume), -1 outside the volume.
        // Bricks that touch no border (workgroup-uniform test) are brick base + per-lane constant.
        int xvv[4], dvv[4];
        if (DG && chunk == nchunk - 1) {
            if (PADOUT) {
                const bool inner = o0d >= 1 && o0h >= 1 && o0w >= 1 && o0d + TD <= Sd + 1 && o0h + TH <= Sh + 1 && o0w + TW <= Sw + 1;
                if (inner) {
                    const int xb = ((o0d - 1 + ld_) * Sh + (o0h - 1 + lh0)) * Sw + (o0w - 1 + lw);
                    const int db = ((o0d + ld_) * (Sh + 2) + (o0h + lh0)) * Wp + (o0w + lw);
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) { xvv[nt] = xb + nt * Sw; dvv[nt] = db + nt * Wp; }
                } else {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const int od = o0d + ld_, oh = o0h + lh[nt], ow = o0w + lw;
                        const bool in = od < Sd + 2 && oh < Sh + 2 && ow < Sw + 2;
                        int rd = od - 1, rh = oh - 1, rw = ow - 1;
                        rd = rd < 0 ? -rd : (rd >= Sd ? 2 * Sd - 2 - rd : rd);
                        rh = rh < 0 ? -rh : (rh >= Sh ? 2 * Sh - 2 - rh : rh);
                        rw = rw < 0 ? -rw : (rw >= Sw ? 2 * Sw - 2 - rw : rw);
                        xvv[nt] = in ? (rd * Sh + rh) * Sw + rw : -1;
                        dvv[nt] = (od * (Sh + 2) + oh) * Wp + ow;
                    }
                }
            } else {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int od = o0d + ld_, oh = o0h + lh[nt], ow = o0w + lw;
                    const bool in = od < Sd && oh < Sh && ow < Sw;
                    xvv[nt] = in ? (od * Sh + oh) * Sw + ow : -1;
                    dvv[nt] = xvv[nt];
                }
            }
        }
        if (DG && MT == 1 && chunk == nchunk - 1) {
            const EpiDesc& e = edesc[g];
            if (e.ok && e.has_stats) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    xpre[nt] = make_uint2(0u, 0u);
                    if (xvv[nt] >= 0) xpre[nt] = gload2(e.x + (unsigned)xvv[nt] * (unsigned)e.C);      // < 2^32 elements per sample
                }
            }
        }
        if (live && !(p.dbg & 2)) {
            const char* bbase = lds + pos0;
            const char* abase = reinterpret_cast<const char*>(wl) + wlane;
#pragma unroll
            for (int s = 0; s < G::NSTEP; ++s) {
                int toff;
                if (NC8 == 1) {
                    toff = toffv[s];
                } else {                              // tap s, block g (already in pos0): a compile-time offset
                    int td = 0, th = 0, tw = 0;
                    if (KS == 3) { td = s / 9; th = (s / 3) % 3; tw = s % 3; }
                    if (DG && KS == 3) { td = 2 - td; th = 2 - th; tw = 2 - tw; }
                    toff = td * G::PS + th * G::RS + tw * VS;
                }
                bf16x8 a[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    a[mt] = *reinterpret_cast<const bf16x8*>(abase + s * (4 * MT * 256) + mt * 256);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const bf16x8 bv = *reinterpret_cast<const bf16x8*>(bbase + toff + nt * G::RS);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], bv, acc[mt][nt], 0, 0, 0);
                }
            }

            // ---- mirror taps (adjoint of reflect padding); only bricks touching planes 1 / Sin-2.  The brick's
            // list of (tap, block) pairs is packed four per MFMA step exactly like the regular taps.
            if (nmirror > 0) {
                const int P = nmirror * NC8;
                auto prep = [&](int s2, bf16x8 (&am)[MT], int (&mpos)[4], bool (&mval)[4]) {
                    const int pr = s2 * 4 + g;
                    const int ent = pr / NC8, blk = pr % NC8;
                    const bool pv = (ent < nmirror) && (cb8 + blk < K8);
                    const int code = mlist[pv ? ent : 0];
                    const int cd = mtab[0][code & 7][ld_], cw = mtab[2][(code >> 6) & 7][lw];
                    const int ah = (code >> 3) & 7;
                    int thp = 0;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const int ch = mtab[1][ah][lh[nt]];
                        thp = ch & 3;
                        mval[nt] = pv && (cd & 4) && (ch & 4) && (cw & 4);
                        mpos[nt] = mval[nt] ? (cd >> 3) * G::PS + (ch >> 3) * G::RS + (cw >> 3) * VS + blk * 16 : 0;
                    }
                    const int tap = ((cd & 3) * KS + thp) * KS + (cw & 3);
                    // the mirror terms reuse the weights of regular taps, already in LDS:
                    // pair q = tap * NC8 + blk sits at wl[(q * MT + mt) * 16 + row]
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) am[mt] = pv ? wl[((tap * NC8 + blk) * MT + mt) * 16 + n] : zero8();
                };
                bf16x8 am[MT]; int mpos[4]; bool mval[4];
                prep(0, am, mpos, mval);
                for (int s2 = 0; s2 * 4 < P; ++s2) {
                    bf16x8 am_n[MT]; int mpos_n[4]; bool mval_n[4];
                    const bool more_m = (s2 + 1) * 4 < P;
                    if (more_m) prep(s2 + 1, am_n, mpos_n, mval_n);
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        bf16x8 bv = zero8();
                        if (mval[nt]) bv = *reinterpret_cast<const bf16x8*>(lds + mpos[nt]);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[mt], bv, acc[mt][nt], 0, 0, 0);
                    }
                    if (more_m) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) am[mt] = am_n[mt];
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) { mpos[nt] = mpos_n[nt]; mval[nt] = mval_n[nt]; }
                    }
                }
            }
        }

        if (chunk == nchunk - 1 && !(p.dbg & 4)) {
            const int oh4[4] = {o0h + lh[0], o0h + lh[1], o0h + lh[2], o0h + lh[3]};
            if (DG) {
                // back through LeakyReLU / mask; the InstanceNorm-backward sums accumulate in registers.  Written without
                // divergent branches (only the load / store are predicated): `continue`s on per-lane conditions turn the
                // unrolled loops into exec-masked regions with copies of the live registers.
                const int dxC = PADOUT ? p.Cin : 0;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const EpiDesc& e = edesc[mt * 4 + g];
                    const bool ok = e.ok != 0, hs = e.has_stats != 0;
                    f32x4 mean, rstd;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { mean[j] = e.mean[j]; rstd[j] = e.rstd[j]; }     // (0, 1) without statistics
                    const float f_pos = e.mval, f_neg = hs ? e.mval * p.slope : e.mval;     // LeakyReLU' x modality mask
                    const int xC = e.C, dC = PADOUT ? dxC : e.C;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const bool valid = ok && xvv[nt] >= 0;
                        uint2 xr = make_uint2(0u, 0u);
                        if (MT == 1) xr = xpre[nt];
                        else if (valid && hs) xr = gload2(e.x + (unsigned)xvv[nt] * (unsigned)xC);
                        const float xv[4] = {__uint_as_float(xr.x << 16), __uint_as_float(xr.x & 0xffff0000u),
                                             __uint_as_float(xr.y << 16), __uint_as_float(xr.y & 0xffff0000u)};
                        const float vf = (valid && hs) ? 1.f : 0.f;
                        f32x4 gv = acc[mt][nt];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float xh = (xv[j] - mean[j]) * rstd[j];
                            const float gg = gv[j] * (xh > 0.f ? f_pos : f_neg);
                            gv[j] = gg;
                            const float gs = gg * vf;
                            s1[mt][j] += gs;
                            s2v[mt][j] = __builtin_fmaf(gs, xh, s2v[mt][j]);
                        }
                        if (valid) gstore4(e.dx + (unsigned)dvv[nt] * (unsigned)dC, gv);
                    }
                }
            } else if (o0d + TD <= Sd && o0h + TH <= Sh && o0w + TW <= Sw) {       // whole brick inside: no per-lane bounds
                // sample base (64-bit, uniform) + 32-bit element offsets: brick (uniform) + lane constant + row tile
                const long sample_el = (long)b * Sd * Sh * Sw * p.Cout;
                bf16_t* y = reinterpret_cast<bf16_t*>(p.y) + sample_el;
                float* yf = reinterpret_cast<float*>(p.y) + sample_el;
                const bf16_t* res = p.residual ? reinterpret_cast<const bf16_t*>(p.residual) + sample_el : nullptr;
                const unsigned brick_el = (unsigned)((o0d * Sh + o0h) * Sw + o0w) * (unsigned)p.Cout;
                const unsigned nt_el = (unsigned)Sw * (unsigned)p.Cout;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int co = (mg * MT + mt) * 16 + g * 4;
                    if (co >= p.Cout) continue;
                    const unsigned lane_el = brick_el + (unsigned)lanevox0 * (unsigned)p.Cout + (unsigned)co;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const unsigned e = lane_el + nt * nt_el;
                        f32x4 v = acc[mt][nt] + biasv[mt];
                        if (p.y_f32) { *reinterpret_cast<f32x4*>(yf + e) = v; continue; }   // logit head
                        if (res) v += load4(res + e);
                        if (fstats) {                       // workgroup-uniform
                            unsigned pk[2];
                            float rb[4];
                            round4_bf16(v, pk, rb);
                            *reinterpret_cast<u32x2*>(y + e) = (u32x2){pk[0], pk[1]};
#pragma unroll
                            for (int j = 0; j < 4; ++j) { s1[mt][j] += rb[j]; s2v[mt][j] = __builtin_fmaf(rb[j], rb[j], s2v[mt][j]); }
                        } else {
                            store4(y + e, v);
                        }
                    }
                }
            } else if (p.y_f32) {
                epilogue_fwd<float, MT>(p, acc, b, mg, o0d + ld_, o0w + lw, oh4, g);
            } else {
                epilogue_fwd<bf16_t, MT>(p, acc, b, mg, o0d + ld_, o0w + lw, oh4, g);
            }
        }
        chunk = n_chunk; tw_ = n_tw; th_ = n_th; td_ = n_td;
    }

    if (fstats) {
        // statistics of y: 16 voxel lanes (shuffles) -> 4 waves (LDS, fixed order) -> the workgroup's slot -> ticket fold
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a1 = group16_sum(s1[mt][j]), a2 = group16_sum(s2v[mt][j]);
                if (n == 0) {
                    lsum[((wv * MT + mt) * 16 + g * 4 + j) * 2 + 0] = a1;
                    lsum[((wv * MT + mt) * 16 + g * 4 + j) * 2 + 1] = a2;
                }
            }
        }
        __syncthreads();
        float* wsum = lsum + 4 * MT * 16 * 2;
        for (int i = tid; i < MT * 16 * 2; i += 256) {
            const int row = i >> 1, k2 = i & 1;
            wsum[i] = ((lsum[(0 * MT * 16 + row) * 2 + k2] + lsum[(1 * MT * 16 + row) * 2 + k2]) + lsum[(2 * MT * 16 + row) * 2 + k2]) +
                      lsum[(3 * MT * 16 + row) * 2 + k2];
        }
        __syncthreads();
        fwd_stats_commit(p.fst, p.Cout, b, wq, mg * MT * 16, MT * 16, wsum, tid);
    }
    if (DG) {
        // one reduction per workgroup: 16 voxel lanes (shuffles) -> 4 waves (LDS slots, summed in a fixed order) -> one replica of
        // the global sums.  The global add is a 64-bit INTEGER atomic on a 2^-40 fixed-point image of the partial sum (sums_fixed):
        // integer addition is associative, so the result does not depend on the order in which the workgroups arrive and the
        // gradients repeat bit for bit (f32 atomics made bf16 gradients differ from run to run by percents downstream).
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const bool st = edesc[mt * 4 + g].ok && edesc[mt * 4 + g].has_stats;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a1 = group16_sum(s1[mt][j]), a2 = group16_sum(s2v[mt][j]);
                if (n == 0) {
                    lsum[((wv * MT + mt) * 16 + g * 4 + j) * 2 + 0] = st ? a1 : 0.f;
                    lsum[((wv * MT + mt) * 16 + g * 4 + j) * 2 + 1] = st ? a2 : 0.f;
                }
            }
        }
        __syncthreads();
        for (int i = tid; i < MT * 16 * 2; i += 256) {
            const int row = i >> 1, k2 = i & 1;
            const int cg = mg * MT * 16 + row;
            const float t = ((lsum[(0 * MT * 16 + row) * 2 + k2] + lsum[(1 * MT * 16 + row) * 2 + k2]) + lsum[(2 * MT * 16 + row) * 2 + k2]) +
                            lsum[(3 * MT * 16 + row) * 2 + k2];
            if (cg < p.Cin && t != 0.f) {
                const long idx = (long)(blockIdx.x % p.sums_rep) * p.B * p.Cin * 2 + ((long)b * p.Cin + cg) * 2 + k2;
                if (p.sums_fixed) {
                    double v = (double)t * 1099511627776.0;                      // 2^40
                    v = fmin(fmax(v, -9.0e18), 9.0e18);
                    atomicAdd(reinterpret_cast<unsigned long long*>(p.sums) + idx, (unsigned long long)(long long)llrint(v));
                } else {
                    atomicAdd(&p.sums[idx], t);
                }
            }
        }
    }
}

template <int KS, int MT, int MODE, int CKB, int S = 1>
__global__ __launch_bounds__(256) void conv_lean_kernel(ConvK p) {
    conv_lean_body<KS, MT, MODE, CKB, S>(p);
}
// the same body held to 128 registers = four waves per SIMD (experiment, PASSION_CONV_OCC4=1): the padded-domain dgrad of
// the 8-channel levels needs 160 and runs three
template <int KS, int MT, int MODE, int CKB, int S = 1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void conv_lean_kernel_occ4(ConvK p) {
    conv_lean_body<KS, MT, MODE, CKB, S>(p);
}

template <int KS, int MT, int MODE, int CKB, int S = 1>
int launch_lean_one(ConvK& k, int K8, int gy, hipStream_t st) {
    using G = GeoL<KS, CKB>;
    constexpr int wchunk = G::NSTEP * 4 * MT * 16 * 16;
    const int nchunk = (K8 + G::NC8 - 1) / G::NC8;
    const int fixed = G::BRICK + DESC_BYTES;
    k.wres = (nchunk == 1 || fixed + nchunk * wchunk <= 48 * 1024) ? 1 : 0;
    static const int lds_pad = []() { const char* e = getenv("PASSION_LDS_PAD_KB"); return e ? atoi(e) * 1024 : 0; }();   // occupancy experiments
    const int bytes = fixed + (k.wres ? nchunk : 1) * wchunk + lds_pad;
    // persistent workgroups: about g_conv_target of them per launch, each walking `per` consecutive bricks of one sample
    const int tiles_per_b = k.tilesD * k.tilesH * k.tilesW;
    const long tiles = (long)k.B * tiles_per_b;
    int per = (int)hmax(1, (tiles * gy + g_conv_target - 1) / g_conv_target);
    per = (int)hmin(hmin(per, tiles_per_b), 256);      // 256 = entries of the workgroup's brick table
    k.per = per;
    const int gx = k.B * ((tiles_per_b + per - 1) / per);
    k.fst.stats = nullptr;
    if (MODE == 0 && k.fst_want && !k.y_f32 && k.Do % TD == 0 && k.Ho % TH == 0 && k.Wo % TW == 0) {
        const int rc = fwd_stats_setup(k, (tiles_per_b + per - 1) / per, gy, st);      // whole bricks only: the ragged epilogue does not count
        if (rc) return rc < 0 ? rc : -rc;
    }
    static int attr_bytes = 0;        // > 64 KB of dynamic LDS has to be granted per kernel
    if (bytes > attr_bytes) {
        HIP_RET(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_lean_kernel<KS, MT, MODE, CKB, S>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        attr_bytes = bytes;
    }
    // measured on the level-1 dgrad (8 -> 8 channels, 128^3): 75.7 -> 61.6 us with the fourth wave (15 dwords spilled to scratch);
    // with two K chunks (16 -> 16 at 64^3) the spills cost more than the wave brings (24.3 -> 27.0 us): single-chunk launches only
    // Opt-in (PASSION_CONV_OCC4=1): alone the kernel gains 19 %, inside the training step nothing measurable (its launches share
    // the chip with the side streams), and the spills show up as scratch traffic: FETCH+WRITE per launch 1.9x -> 2.8x algorithmic.
    static const bool occ4 = []() { const char* e = getenv("PASSION_CONV_OCC4"); return e && e[0] == '1'; }();
    const bool use_occ4 = occ4 && MODE == 2 && CKB == 8 && MT == 1 && S == 1 && KS == 3 && K8 == 1;
    if (S == 1) passion_prof_tag(use_occ4 ? "conv_lean_kernel_occ4<%d, %d, %d, %d, 1>" : "conv_lean_kernel<%d, %d, %d, %d, 1>", KS, MT, MODE, CKB);
    else passion_prof_tag("conv_lean_kernel<%d, %d, %d, %d, %d>", KS, MT, MODE, CKB, S);
    if (use_occ4) {
        static int attr2 = 0;
        if (bytes > attr2) {
            HIP_RET(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_lean_kernel_occ4<KS, MT, MODE, CKB, S>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
            attr2 = bytes;
        }
        hipLaunchKernelGGL((conv_lean_kernel_occ4<KS, MT, MODE, CKB, S>), dim3(gx, gy), dim3(256), bytes, st, k);
    } else
        hipLaunchKernelGGL((conv_lean_kernel<KS, MT, MODE, CKB, S>), dim3(gx, gy), dim3(256), bytes, st, k);
    LAUNCH_RET();
    return 0;
}

template <int KS, int MODE, int S = 1>
int launch_lean_geo(ConvK& k, int K8, int cmax, int mtiles, hipStream_t st) {
    const long tiles = (long)k.B * k.tilesD * k.tilesH * k.tilesW;
    // row tiles per workgroup: two share a staged brick, but their weight fragments (2 x 27.6 KB per 32-channel chunk) leave room for
    // ONE workgroup per CU.  One row tile for the padded-domain dgrad (PASSION_LEAN_MT_TILES=4096) ran the grouped level-3 launch
    // (4 x 34^3, 32 -> 32: 900 bricks) in 41.8 instead of 69.9 us ALONE, but inside the training step the same launches stayed at
    // ~115 us and the step rate did not move (83.9 vs 83.9 vol/s): not adopted.
    static const long mt_env = []() { const char* e = getenv("PASSION_LEAN_MT_TILES"); return e ? atol(e) : 0L; }();
    const long mt_tiles = mt_env > 0 ? mt_env : 512L;
    if (cmax >= 32) {                 // 32-channel chunks; weight fragments of a chunk fit LDS up to two row tiles
        int MT = (mtiles % 2 == 0) ? 2 : 1;
        while (MT > 1 && tiles * (mtiles / MT) < mt_tiles) MT >>= 1;
        if (MT == 2) return launch_lean_one<KS, 2, MODE, 32, S>(k, K8, mtiles / 2, st);
        return launch_lean_one<KS, 1, MODE, 32, S>(k, K8, mtiles, st);
    }
    int MT = mtiles >= 4 ? 4 : mtiles;
    if (mtiles % MT) MT = (mtiles % 2 == 0) ? 2 : 1;
    while (MT > 1 && tiles * (mtiles / MT) < 512) MT >>= 1;
    if (MT == 4) return launch_lean_one<KS, 4, MODE, 8, S>(k, K8, mtiles / 4, st);
    if (MT == 2) return launch_lean_one<KS, 2, MODE, 8, S>(k, K8, mtiles / 2, st);
    return launch_lean_one<KS, 1, MODE, 8, S>(k, K8, mtiles, st);
}

}  // namespace

namespace pconv {

// mode 0 forward, 1 dgrad, 2 dgrad over the padded domain (see PADOUT; k.tiles* cover (S+2)^3, k.y = padded scratch)
// returns 0 = launched, < 0 = HIP error, 1 = shape not covered (caller falls back to the general kernel)
bool lean_covers(const ConvK& k, int stride, bool dgrad) {
    if (stride != 1 && !(stride == 2 && dgrad)) return false;     // stride 2: only the padded-domain dgrad (mode 2)
    if (dgrad) return (k.Cout % 8 == 0 || k.y_f32) && (k.Cout + 7) / 8 <= (k.Cout >= 32 ? MAXBLK : 32);
    int K8 = 0, cmax = 0;
    for (int s = 0; s < k.nseg; ++s) {
        if (k.seg[s].C % 8) return false;
        K8 += k.seg[s].C / 8;
        cmax = k.seg[s].C > cmax ? k.seg[s].C : cmax;
    }
    return K8 * 8 == k.Cin && K8 <= (cmax >= 32 ? MAXBLK : 32);
}

int launch_conv_lean(ConvK& k, int ksize, int stride, int mode, const void* packed_bf16, hipStream_t st) {
    {   // the two finest levels: marching kernel (conv_march.hip); > 0 = not covered, go on
        const int rc = launch_conv_march(k, ksize, stride, mode, packed_bf16, st);
        if (rc <= 0) return rc;
    }
    const bool dgrad = mode >= 1;
    if (stride == 2 && mode == 2 && ksize == 3) {   // gather form (conv_s2_dgrad.hip); > 0 = not covered: the zero-stuffed path below
        const int rc = launch_conv_s2_dgrad(k, packed_bf16, st);
        if (rc <= 0) return rc;
    }
    if (stride != 1 && !(stride == 2 && mode == 2)) return 1;
    if (!(k.slope >= 0.f && k.slope <= 1.f)) return 1;      // the loader's LeakyReLU is max(t, slope * t)
    if (mode == 2 && ksize != 3) return 1;
    const int ntap = ksize * ksize * ksize;
    int K8, cmax = 0;
    if (dgrad) {
        if (k.Cout % 8 && !k.y_f32) return 1;
        K8 = (k.Cout + 7) / 8;
        cmax = k.Cout;
    } else {
        K8 = 0;
        for (int s = 0; s < k.nseg; ++s) {
            if (k.seg[s].C % 8) return 1;
            K8 += k.seg[s].C / 8;
            cmax = k.seg[s].C > cmax ? k.seg[s].C : cmax;
        }
        if (K8 * 8 != k.Cin) return 1;
    }
    if (K8 > (cmax >= 32 ? MAXBLK : 32)) return 1;
    if (dgrad) {
        k.Mpad = pad16(k.Cin);
        const int cout_w = k.bd_rows ? k.bd_k8 * 8 : k.Cout;      // the packed images are those of the real weight
        k.wpack = reinterpret_cast<const bf16_t*>(packed_bf16) + (long)ntap * pad8(k.Cin) * pad16(cout_w);
    } else {
        k.Mpad = pad16(k.Cout);
        k.wpack = packed_bf16;
    }
    k.dbg = g_conv_dbg;
    k.sums_rep = SUMS_REP_LEAN;
    k.sums_fixed = k.sums_float_only ? 0 : 1;
    const int mtiles = k.Mpad / 16;
    if (ksize == 1 && mode == 0 && stride == 1) {             // wide volumes: the streaming pointwise kernel (conv_pw.hip); > 0 = not covered
        const int rc = launch_conv_pw_fwd(k, K8, st);
        if (rc <= 0) return rc;
    }
    if (ksize == 1 && mode == 1 && stride == 1) {
        const int rc = launch_conv_pw_dgrad(k, K8, st);
        if (rc <= 0) return rc;
    }
    {   // levels 3..5: K chunks side by side, weights straight from L2 (conv_deep.hip); > 0 = not covered: the brick kernels below
        const int rc = launch_conv_deep(k, ksize, stride, mode, st);
        if (rc <= 0) return rc;
    }
    if (ksize == 1) return dgrad ? launch_lean_geo<1, 1>(k, K8, cmax, mtiles, st) : launch_lean_geo<1, 0>(k, K8, cmax, mtiles, st);
    if (mode == 2 && stride == 2) return launch_lean_geo<3, 2, 2>(k, K8, cmax, mtiles, st);
    if (mode == 2) return launch_lean_geo<3, 2>(k, K8, cmax, mtiles, st);
    return dgrad ? launch_lean_geo<3, 1>(k, K8, cmax, mtiles, st) : launch_lean_geo<3, 0>(k, K8, cmax, mtiles, st);
}

}  // namespace pconv
