// Deep-level bf16 implicit-GEMM conv: forward and input gradient of the stride-1 convs whose K axis is made of 32-channel chunks
// (levels 3..5 of the encoders / decoders: 32 ... 128 channels over 32^3 ... 8^3 voxels; reference code/models/mmformer.py:40-64,
// 66-114, 116-189 and the fusion blocks code/models/blocks.py:495-542).
//
// These launches move a few MB each and were 4.9 of 10.5 ms of a training step's kernel time on conv_lean.hip at 1-4 % of any
// roofline (profiles/r03e_launch_table.txt): a lean workgroup owns a 4x8x8 brick and 16/32 output channels, keeps its weights in
// LDS (28-55 KB on top of a 48 KB brick: ONE workgroup per CU) and walks the 32-channel chunks of the K axis one after the other
// (fetch -> normalise -> barrier -> 27 x 4 MFMAs -> next chunk): 16-64 workgroups x a serial chain of up to 4 chunks at level 5.
// Here the K chunks are worked on SIDE BY SIDE by the four waves of a workgroup and the launch is cut finer:
//   * a workgroup owns one flat run of FT = 64 / 128 outputs of ONE (H, W) plane of one sample and NMW 16-row tiles of the M axis;
//     its waves are (K chunk kq) x (row tile mq) x (column half nq), NKW * NMW * NNW = 4.  A 128 -> 128 conv over 4 x 8^3
//     voxels is 256 workgroups of 4 waves x 108 MFMAs instead of 64 workgroups x 4 chunks x 108;
//   * weights never pass through LDS: the A fragment of (tap, chunk, row tile) is 16 bytes per lane straight from the packed
//     image (L2-resident: every workgroup of a sample reads the same 0.9 MB), nine taps requested ahead of the nine in use;
//   * the staged input (planes d-1 .. d+1, the rows the run touches +- 1, full width + 2) holds all NKW chunks at once; one
//     barrier, then every wave runs its 27 taps without another;
//   * the partial sums of the K chunks meet in LDS (fixed order: bit-reproducible), the kq = 0 waves run the epilogue.
// Flat runs instead of 8x8 tiles: the padded-domain input gradient (MODE 2, see conv_lean.hip) works on (S+2)^3 = 10^3 / 18^3 /
// 34^3 outputs and the reference's default crop gives 5^3 / 10^3 / 20^3 -- no tile shape fits them all, a run of a plane does.
// Same operand layouts, packed weight images, scratch / sums contract and epilogue arithmetic as conv_lean.hip (finish kernels
// and fused statistics unchanged).
#include <cstdlib>
#include "conv_common.h"
#include "prof.h"
#include "wgrad_lean_body.h"

namespace pconv {
extern int g_conv_dbg;
int g_conv_deep = 1;
int g_deep_pair = 0;          // option "deep_pair" / PASSION_DEEP_PAIR=1: input gradient + weight gradient of a deep-level 3x3x3 conv as one launch.
                              // OFF: measured 1.0-1.3 % SLOWER on the plain and the PASSION step (6 interleaved runs each): the
                              // weight gradient can then no longer clear the InstanceNorm-backward sums in its own launch (a memset node per conv
                              // takes the launch's place), and the launches it removes already ran beside other streams
PendingWgrad& pending_wgrad() { static thread_local PendingWgrad pw{}; return pw; }
bool& defer_next_wgrad() { static thread_local bool f = false; return f; }
}
using namespace pconv;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int VS = 80;            // bytes per staged voxel of one 32-channel chunk (64 + 16: spreads the 16-byte slots over the banks)
constexpr int MAXCH = 16;         // K <= 512 channels
constexpr int NT = 4;             // 16-voxel column tiles per wave

struct DeepP {
    int Od, Oh, Ow, PF;           // output domain of the GEMM (forward: the volume; MODE 2: the padded volume) and Oh * Ow
    int FT, nfx;                  // outputs per workgroup, workgroups per plane
    int NRmax, RW;                // staged rows (most) and row width in voxels (KS 3: rows + 2, Ow + 2; KS 1: the run itself)
    int SH;                       // source coordinate of staged index z along an axis: output origin - SH + z
    int nchunk, K8;               // 32-channel chunks / 8-channel blocks of the K axis
    float inv_rw, inv_ow;         // 1 / RW, 1 / Ow (index splitting without integer division)
    int red_off;                  // byte offset of the chunk descriptors behind the staged region
};

struct alignas(16) ChunkDesc {    // one 32-channel chunk of the K axis, resolved for this workgroup's sample
    const bf16_t* src;            // sample base + channel offset
    int C, on, has_norm, pad_;    // voxel stride in elements; present (modality mask); InstanceNorm + LeakyReLU in the loader
    float rstd[32], shift[32];    // normalised value = x * rstd + shift
};
static_assert(sizeof(ChunkDesc) == 288, "ChunkDesc layout");
struct alignas(16) EpiD {         // input-gradient epilogue: the 16 output rows (= forward input channels) of one row tile
    const bf16_t* x;              // forward input of those channels (sample base + channel)
    bf16_t* dx;
    int C, ok, has_stats;
    float mval;
    float mean[16], rstd[16];
};

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T> __device__ __forceinline__ T gload(const void* ptr) {
    return *reinterpret_cast<const T __attribute__((address_space(1)))*>(reinterpret_cast<unsigned long long>(ptr));
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
// v / d for 0 <= v < 2^20 and small d through the reciprocal (exact: (v + 0.5) / d is at least 0.5 / d away from an integer)
__device__ __forceinline__ int fdiv(int v, float inv) { return (int)(((float)v + 0.5f) * inv); }

// the 32-channel chunk c of the K axis for sample b, from the launch descriptor (uniform: scalar code)
struct ChunkHdr { const bf16_t* src; int C; bool on, has_norm; };
template <bool DG>
__device__ __forceinline__ ChunkHdr chunk_hdr(const ConvK& p, int b, int c, long vol) {
    ChunkHdr h{nullptr, 8, false, false};
    if (DG) {
        h.src = reinterpret_cast<const bf16_t*>(p.dy) + (long)b * vol * p.Cout + c * 32;
        h.C = p.Cout; h.on = true;
        return h;
    }
    int r = c * 32;
    bool found = false;
#pragma unroll
    for (int s2 = 0; s2 < PASSION_MAX_SEG; ++s2) {
        if (s2 < p.nseg && !found) {
            const SegD& sd = p.seg[s2];
            if (r < sd.C) {
                found = true;
                h.src = reinterpret_cast<const bf16_t*>(sd.x) + (long)sd.bsrc(b) * sd.bstride + r;
                h.C = sd.C; h.has_norm = sd.stats != nullptr;
                h.on = !sd.mask || sd.mask[(long)b * sd.mask_ld + sd.mask_col] != 0;
            }
            r -= sd.C;
        }
    }
    return h;
}

// MODE 0 forward; 1 input gradient written in place (KS 1 only); 2 input gradient over the padded domain (KS 3)
// MAXU: staged voxels per thread and chunk (64 voxel slots x 4 blocks per pass over the staged region)
template <int KS, int MODE, int NKW, int NMW, int NNW, int MAXU>
__device__ __forceinline__ void conv_deep_body(const ConvK& p, const DeepP& q, const int bxi, const int byi, const int gxn) {
    static_assert(NKW * NMW * NNW == 4, "four waves");
    static_assert(MODE == 0 || (MODE == 1 && KS == 1) || (MODE == 2 && KS == 3), "modes");
    constexpr bool DG = MODE >= 1;
    constexpr bool PADOUT = MODE == 2;
    constexpr int NTAP = KS * KS * KS;
    constexpr int NPL = KS == 3 ? 3 : 1;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    __shared__ float lsum[4 * 16 * 2 + NMW * 16 * 2];
    __shared__ EpiD edesc[DG ? NMW : 1];

    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int wv = rfl(tid >> 6);
    const int kq = wv % NKW, mq = (wv / NKW) % NMW, nq = wv / (NKW * NMW);
    ChunkDesc* desc = reinterpret_cast<ChunkDesc*>(lds + q.red_off);

    // ---- which outputs: sample b, plane od, flat run [f0, f0 + FT) of the plane
    const int bx = xcd_swizzle(bxi, gxn);
    const int fx = bx % q.nfx, od = (bx / q.nfx) % q.Od, b = bx / (q.nfx * q.Od);
    const int f0 = fx * q.FT, f1 = min(f0 + q.FT, q.PF) - 1;
    const int h_lo = fdiv(f0, q.inv_ow), h_hi = fdiv(f1, q.inv_ow);
    const int NRc = KS == 3 ? h_hi - h_lo + 3 : 1;               // staged rows of this workgroup
    const int RS = q.RW * VS, PS = NRc * RS, CS = NPL * PS;      // row / plane / chunk strides of the staged region
    const int mtile = byi * NMW + mq;                     // 16-row tile of the M axis
    const int Sd = p.Di, Sh = p.Hi, Sw = p.Wi;
    const long vol = (long)Sd * Sh * Sw;
    const int K8 = q.K8;
    const bf16x8* wp = reinterpret_cast<const bf16x8*>(p.wpack) + wvec_of(p, b);
    const int niter = (q.nchunk + NKW - 1) / NKW;
    const bool rows_ok = mtile * 16 < p.Mpad;

    // ---- A fragments: (tap, chunk, row tile) = 16 bytes per lane straight from the packed image; nine taps (one depth
    // tap) are requested while the nine before them are in use
    constexpr int AG = KS == 3 ? 9 : 1, NG = NTAP / AG;
    const long wtap = (long)K8 * p.Mpad;                          // vectors between consecutive taps
    const bf16x8* wa0 = wp + ((long)g * p.Mpad + mtile * 16 + n);     // + chunk * 4 * Mpad
    constexpr int NBUF = NG > 1 ? 2 : 1;                          // group tg lives in buffer tg & 1; the next iteration's first group is
    bf16x8 abuf[NBUF][AG];                                        // requested after this iteration's last (the commit phase covers it)
    auto a_load = [&](int it, int tg, bf16x8 (&dst)[AG]) {
        const int cw = min(it * NKW + kq, q.nchunk - 1);
        const bf16x8* wa = wa0 + (long)cw * 4 * p.Mpad + (long)tg * AG * wtap;
#pragma unroll
        for (int t = 0; t < AG; ++t) dst[t] = (rows_ok && !(p.dbg & 8)) ? gload<bf16x8>(wa + t * wtap) : zero8();
    };
    a_load(0, 0, abuf[0]);

    // ---- staging geometry of this thread, once: voxel slot v0 + 64 u of the staged region -> source voxel (clamped), in-range bit
    const int NV = KS == 3 ? NPL * NRc * q.RW : (f1 - f0 + 1);    // staged voxels per chunk
    const float inv_nr = 1.0f / (float)NRc;
    const int jj = tid & 3, v0 = tid >> 2;
    int voxoff[MAXU];
    unsigned okgeo = 0u;
#pragma unroll
    for (int u = 0; u < MAXU; ++u) {
        const int v = min(v0 + 64 * u, NV - 1);
        bool ok = v0 + 64 * u < NV;
        if (KS == 3) {
            const int t = fdiv(v, q.inv_rw), zw = v - t * q.RW;
            const int zd = fdiv(t, inv_nr), zr = t - zd * NRc;
            int sd_ = od - q.SH + zd, sh_ = h_lo - q.SH + zr, sw_ = zw - q.SH;
            if (!DG && p.pad_mode == PASSION_PAD_REFLECT) {
                sd_ = reflect_idx(sd_, Sd); sh_ = reflect_idx(sh_, Sh); sw_ = reflect_idx(sw_, Sw);
            } else {
                ok = ok && sd_ >= 0 && sd_ < Sd && sh_ >= 0 && sh_ < Sh && sw_ >= 0 && sw_ < Sw;
            }
            sd_ = min(max(sd_, 0), Sd - 1); sh_ = min(max(sh_, 0), Sh - 1); sw_ = min(max(sw_, 0), Sw - 1);
            voxoff[u] = (sd_ * Sh + sh_) * Sw + sw_;
        } else {
            voxoff[u] = od * q.PF + f0 + v;
        }
        if (ok) okgeo |= 1u << u;
    }
    u32x4 raw[NKW * MAXU];
    unsigned chunk_on = 0u, chunk_norm = 0u;                      // of the iteration in `raw`, bit kc
    auto fetch = [&](int it) {
        chunk_on = 0u; chunk_norm = 0u;
#pragma unroll
        for (int kc = 0; kc < NKW; ++kc) {
            const int c = it * NKW + kc;
            const ChunkHdr h = chunk_hdr<DG>(p, b, min(c, q.nchunk - 1), vol);
            const bool on = h.on && c < q.nchunk;
            if (on) chunk_on |= 1u << kc;
            if (on && h.has_norm) chunk_norm |= 1u << kc;
            const bf16_t* src = h.src + jj * 8;
#pragma unroll
            for (int u = 0; u < MAXU; ++u)                        // unconditional, clamped: nothing waits on a branch
                raw[kc * MAXU + u] = (p.dbg & 16) ? (u32x4){0u, 0u, 0u, 0u} : gload<u32x4>(src + (unsigned)voxoff[u] * (unsigned)h.C);
        }
    };
    fetch(0);

    // ---- once per workgroup: InstanceNorm statistics of the K axis (one thread per channel) -> LDS
    if (!DG) {
        for (int i = tid; i < q.nchunk * 32; i += 256) {
            const int c = i >> 5, ch = i & 31;
            int r = c * 32;
            bool found = false;
            float rs = 1.f, sh = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < PASSION_MAX_SEG; ++s2) {
                if (s2 < p.nseg && !found) {
                    const SegD& sd = p.seg[s2];
                    if (r < sd.C) {
                        found = true;
                        if (sd.stats) {
                            const int bs = sd.bsrc(b);
                            rs = sd.stats[((long)bs * sd.C + r + ch) * 2 + 1];
                            sh = -sd.stats[((long)bs * sd.C + r + ch) * 2 + 0] * rs;
                        }
                    }
                    r -= sd.C;
                }
            }
            desc[c].rstd[ch] = rs; desc[c].shift[ch] = sh;
        }
    }
    if (DG && tid >= 192 && tid < 192 + NMW * 16) {              // epilogue descriptors: one thread per output row
        const int e = (tid - 192) >> 4, r = (tid - 192) & 15;
        const int cg = (byi * NMW + e) * 16 + r;
        EpiD& d = edesc[e];
        if (r == 0) { d.x = nullptr; d.dx = nullptr; d.C = 16; d.ok = 0; d.has_stats = 0; d.mval = 1.f; }
        d.mean[r] = 0.f; d.rstd[r] = 1.f;
#pragma unroll
        for (int s2 = 0; s2 < PASSION_MAX_SEG; ++s2) {
            if (s2 < p.nseg && cg < p.Cin) {
                const SegD& sd = p.seg[s2];
                if (cg >= sd.cbase && cg < sd.cbase + sd.C && sd.dx != nullptr) {
                    const int cl = cg - sd.cbase, bs = sd.bsrc(b);
                    if (r == 0) {
                        d.ok = 1; d.C = sd.C; d.has_stats = sd.stats != nullptr;
                        d.x = reinterpret_cast<const bf16_t*>(sd.x) + (long)bs * sd.bstride + cl;
                        d.dx = PADOUT ? reinterpret_cast<bf16_t*>(p.y) + (long)b * (Sd + 2) * (Sh + 2) * pad8(Sw + 2) * p.Cin + cg
                                      : reinterpret_cast<bf16_t*>(sd.dx) + (long)b * sd.bstride + cl;
                        if (sd.mask) d.mval = sd.mask[(long)b * sd.mask_ld + sd.mask_col] ? 1.f : 0.f;
                    }
                    if (sd.stats) {
                        d.mean[r] = sd.stats[((long)bs * sd.C + cl) * 2 + 0];
                        d.rstd[r] = sd.stats[((long)bs * sd.C + cl) * 2 + 1];
                    }
                }
            }
        }
    }

    // ---- per-lane geometry of the wave's column tiles: voxel f of the plane -> (h, w), staged offset, validity
    int fv[NT], boff[NT];
    bool fok[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int f = f0 + (nq * NT + nt) * 16 + n;
        fok[nt] = f <= f1;
        fv[nt] = min(f, f1);
        const int h = fdiv(fv[nt], q.inv_ow), w = fv[nt] - h * q.Ow;
        boff[nt] = (KS == 3 ? (h - h_lo) * RS + w * VS : (fv[nt] - f0) * VS) + g * 16 + kq * CS;
    }
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x2 slope2 = {p.slope, p.slope};
    __syncthreads();                                              // statistics visible

    for (int it = 0; it < niter; ++it) {
        // ---- commit: the fetched chunks -> normalise + LeakyReLU -> staged region
        const unsigned c_on = chunk_on, c_norm = chunk_norm;
#pragma unroll
        for (int kc = 0; kc < NKW; ++kc) {
            if (p.dbg & 1) break;
            const ChunkDesc& d = desc[min(it * NKW + kc, q.nchunk - 1)];
            const bool on = (c_on >> kc) & 1u, norm = (c_norm >> kc) & 1u;
            f32x2 sc[4], sh[4];
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) {
                sc[k2] = (f32x2){d.rstd[jj * 8 + 2 * k2], d.rstd[jj * 8 + 2 * k2 + 1]};
                sh[k2] = (f32x2){d.shift[jj * 8 + 2 * k2], d.shift[jj * 8 + 2 * k2 + 1]};
            }
            char* dst = lds + kc * CS + jj * 16;
#pragma unroll
            for (int u = 0; u < MAXU; ++u) {
                const int v = v0 + 64 * u;
                if (v >= NV) continue;
                const u32x4 r4 = raw[kc * MAXU + u];
                u32x4 o = (u32x4){0u, 0u, 0u, 0u};
                if (on && ((okgeo >> u) & 1u)) {
                    o = r4;
                    if (norm) {
                        bf16x8 ob;
#pragma unroll
                        for (int k2 = 0; k2 < 4; ++k2) {
                            const f32x2 xv = {__uint_as_float(r4[k2] << 16), __uint_as_float(r4[k2] & 0xffff0000u)};
                            f32x2 t = __builtin_elementwise_fma(xv, sc[k2], sh[k2]);
                            const f32x2 ts = t * slope2;
                            t.x = fmaxf(t.x, ts.x);
                            t.y = fmaxf(t.y, ts.y);
                            ob[2 * k2] = (__bf16)t.x; ob[2 * k2 + 1] = (__bf16)t.y;
                        }
                        o = __builtin_bit_cast(u32x4, ob);
                    }
                }
                *reinterpret_cast<u32x4*>(dst + v * VS) = o;
            }
        }
        const bool wlive = ((c_on >> kq) & 1u) && rows_ok && !(p.dbg & 2);
        __syncthreads();
        if (it + 1 < niter) fetch(it + 1);

        // ---- this wave's chunk: NTAP taps x NT column tiles, A fragments two groups ahead
#pragma unroll
        for (int tg = 0; tg < NG; ++tg) {
            if (tg + 1 < NG) a_load(it, tg + 1, abuf[(tg + 1) & (NBUF - 1)]);      // one group ahead of use
            if (wlive) {
#pragma unroll
                for (int t = 0; t < AG; ++t) {
                    int toff = 0;
                    if (KS == 3) {
                        int td = tg, th = t / 3, tw = t % 3;
                        if (DG) { td = 2 - td; th = 2 - th; tw = 2 - tw; }     // gather form: output o reads u = o - t
                        toff = td * PS + th * RS + tw * VS;
                    }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const bf16x8 bv = *reinterpret_cast<const bf16x8*>(lds + boff[nt] + toff);
                        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(abuf[tg & (NBUF - 1)][t], bv, acc[nt], 0, 0, 0);
                    }
                }
            }
        }
        if (it + 1 < niter) a_load(it + 1, 0, abuf[0]);
        if (it + 1 < niter) __syncthreads();                      // staged data consumed before the next commit
    }

    // ---- the K chunks' partial sums meet in LDS (fixed order), the kq = 0 waves go on
    if (NKW > 1) {
        __syncthreads();                                          // staged data dead
        f32x4* red = reinterpret_cast<f32x4*>(lds);
        if (kq > 0) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) red[(((kq - 1) * NMW * NNW + mq * NNW + nq) * NT + nt) * 64 + lane] = acc[nt];
        }
        __syncthreads();
        if (kq == 0) {
#pragma unroll
            for (int k2 = 1; k2 < NKW; ++k2)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] += red[(((k2 - 1) * NMW * NNW + mq * NNW + nq) * NT + nt) * 64 + lane];
        }
    }

    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.dbg & 4) return;
    if (!DG) {
        const bool fstats = p.fst.stats != nullptr;
        const int co = mtile * 16 + g * 4;
        if (kq == 0 && rows_ok && co < p.Cout) {
            const long sample_el = ((long)b * q.Od + od) * q.PF * p.Cout;
            bf16_t* y = reinterpret_cast<bf16_t*>(p.y) + sample_el;
            const bf16_t* res = p.residual ? reinterpret_cast<const bf16_t*>(p.residual) + sample_el : nullptr;
            const float* bias = bias_of(p, b);
            const f32x4 bv = bias ? load4(bias + co) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (!fok[nt]) continue;
                const unsigned e = (unsigned)fv[nt] * (unsigned)p.Cout + (unsigned)co;
                f32x4 v = acc[nt] + bv;
                if (res) v += load4(res + e);
                unsigned pk[2];
                float rb[4];
                round4_bf16(v, pk, rb);
                *reinterpret_cast<u32x2*>(y + e) = (u32x2){pk[0], pk[1]};
#pragma unroll
                for (int j = 0; j < 4; ++j) { s1[j] += rb[j]; s2[j] = __builtin_fmaf(rb[j], rb[j], s2[j]); }
            }
        }
        if (!fstats) return;                                      // workgroup-uniform
    } else {
        // back through LeakyReLU / the modality mask; x_hat from the forward input at the (reflected) position; the
        // InstanceNorm-backward sums of the workgroup go to one replica of the global sums (conv_lean.hip: same contract)
        if (kq == 0 && rows_ok) {
            const EpiD& e = edesc[mq];
            const bool ok = e.ok != 0, hs = e.has_stats != 0;
            f32x4 mean, rstd;
#pragma unroll
            for (int j = 0; j < 4; ++j) { mean[j] = e.mean[g * 4 + j]; rstd[j] = e.rstd[g * 4 + j]; }
            const float f_pos = e.mval, f_neg = hs ? e.mval * p.slope : e.mval;
            const int Wp = pad8(Sw + 2);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const bool valid = ok && fok[nt];
                long xvox, dvox;
                if (PADOUT) {
                    const int oh = fdiv(fv[nt], q.inv_ow), ow = fv[nt] - oh * q.Ow;
                    int rd = od - 1, rh = oh - 1, rw = ow - 1;
                    rd = rd < 0 ? -rd : (rd >= Sd ? 2 * Sd - 2 - rd : rd);
                    rh = rh < 0 ? -rh : (rh >= Sh ? 2 * Sh - 2 - rh : rh);
                    rw = rw < 0 ? -rw : (rw >= Sw ? 2 * Sw - 2 - rw : rw);
                    xvox = ((long)rd * Sh + rh) * Sw + rw;
                    dvox = ((long)od * (Sh + 2) + oh) * Wp + ow;
                } else {
                    xvox = (long)od * q.PF + fv[nt];
                    dvox = xvox;
                }
                u32x2 xr = {0u, 0u};
                if (valid && hs) xr = gload<u32x2>(e.x + xvox * e.C + g * 4);
                const float xv[4] = {__uint_as_float(xr.x << 16), __uint_as_float(xr.x & 0xffff0000u),
                                     __uint_as_float(xr.y << 16), __uint_as_float(xr.y & 0xffff0000u)};
                const float vf = (valid && hs) ? 1.f : 0.f;
                f32x4 gv = acc[nt];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xh = (xv[j] - mean[j]) * rstd[j];
                    const float gg = gv[j] * (xh > 0.f ? f_pos : f_neg);
                    gv[j] = gg;
                    const float gs = gg * vf;
                    s1[j] += gs;
                    s2[j] = __builtin_fmaf(gs, xh, s2[j]);
                }
                if (valid) gstore4(e.dx + dvox * (PADOUT ? p.Cin : e.C) + g * 4, gv);
            }
        }
    }

    // ---- per-row sums of the workgroup: 16 voxel lanes (shuffles) -> waves with the same row tile (LDS, fixed order)
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a1 = group16_sum(s1[j]), a2 = group16_sum(s2[j]);
        if (n == 0) {
            lsum[(wv * 16 + g * 4 + j) * 2 + 0] = a1;
            lsum[(wv * 16 + g * 4 + j) * 2 + 1] = a2;
        }
    }
    __syncthreads();
    float* wsum = lsum + 4 * 16 * 2;
    for (int i = tid; i < NMW * 16 * 2; i += 256) {
        const int m = i / 32, r2 = i % 32;
        float t = 0.f;
#pragma unroll
        for (int nq2 = 0; nq2 < NNW; ++nq2) t += lsum[(((nq2 * NMW + m) * NKW) * 16) * 2 + r2];      // the kq = 0 wave of (m, nq2)
        wsum[i] = t;
    }
    __syncthreads();
    if (!DG) {
        fwd_stats_commit(p.fst, p.Cout, b, od * q.nfx + fx, byi * NMW * 16, NMW * 16, wsum, tid);
    } else {
        for (int i = tid; i < NMW * 16 * 2; i += 256) {
            const int row = i >> 1, k2 = i & 1;
            const int cg = byi * NMW * 16 + row;
            const float t = wsum[i];
            if (cg < p.Cin && t != 0.f) {
                const long idx = (long)(bxi % p.sums_rep) * p.B * p.Cin * 2 + ((long)b * p.Cin + cg) * 2 + k2;
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

template <int KS, int MODE, int NKW, int NMW, int NNW, int MAXU>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void conv_deep_kernel(ConvK p, DeepP q) {
    conv_deep_body<KS, MODE, NKW, NMW, NNW, MAXU>(p, q, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x);
}

// Input gradient (conv_deep_body, MODE 2) and weight gradient (wgrad_lean_body) of ONE 3x3x3 deep-level conv as two workgroup ranges of
// one grid: workgroups [0, nd) are the input gradient's (gxd x gyd), the rest the weight gradient's (gxw x gyw).  Both read (x, dY), neither
// depends on the other, each is 64..256 workgroups bound by its own latency chain (descriptor -> loads -> MFMA -> epilogue): side by side
// they cost what the longer one costs, and a conv backward at levels 3..5 is two launches (this one, the finish) instead of three.
template <int NKW, int NMW, int NNW, int MAXU, int MT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void deep_pair_kernel(ConvK pd, DeepP q, ConvK pw, float* __restrict__ dW,
                                                                                                 float* __restrict__ dbias, int gxd, int gyd, int gxw) {
    const int bid = (int)blockIdx.x, nd = gxd * gyd;
    if (bid < nd) {
        conv_deep_body<3, 2, NKW, NMW, NNW, MAXU>(pd, q, bid % gxd, bid / gxd, gxd);
    } else {
        const int w = bid - nd;
        pconv_wl::wgrad_lean_body<3, MT>(pw, dW, dbias, w % gxw, w / gxw, gxw);
    }
}

struct DeepPlan { int nkw, nmw, nnw, maxu; long wgs; };

// geometry of a launch with waves (nkw, nmw, nnw): fills q, returns the dynamic LDS bytes (0 = does not fit) and the staged voxels
template <int KS>
long deep_geometry(DeepP& q, int nkw, int nmw, int nnw, int* nv_max) {
    q.FT = nnw * NT * 16;
    q.nfx = (q.PF + q.FT - 1) / q.FT;
    int span = 1;                                                  // the most rows a run touches
    for (int fx = 0; fx < q.nfx; ++fx) {
        const int a = fx * q.FT, b2 = (int)hmin(a + q.FT, q.PF) - 1;
        span = (int)hmax(span, b2 / q.Ow - a / q.Ow + 1);
    }
    q.NRmax = KS == 3 ? span + 2 : 1;
    q.RW = KS == 3 ? q.Ow + 2 : q.FT;
    q.inv_rw = 1.0f / (float)q.RW;
    q.inv_ow = 1.0f / (float)q.Ow;
    const long nv = (long)(KS == 3 ? 3 : 1) * q.NRmax * q.RW;
    *nv_max = (int)nv;
    const long stage = (long)nkw * nv * VS;
    const long red = nkw > 1 ? (long)(nkw - 1) * nmw * nnw * NT * 64 * 16 : 0;
    q.red_off = (int)((hmax(stage, red) + 15) / 16 * 16);
    const long bytes = q.red_off + (long)q.nchunk * (long)sizeof(ChunkDesc);
    return bytes <= 160 * 1024 - 4096 ? bytes : 0;
}

// the pair launch: 0 = launched (the parked weight gradient is consumed), 1 = not for this instantiation
template <int NKW, int NMW, int NNW, int MAXU>
int launch_deep_pair(ConvK& k, DeepP& q, long bytes, int gx, int gy, hipStream_t st) {
    PendingWgrad& pw = pending_wgrad();
    constexpr long WG_STATIC = 33 * 1024;                 // static LDS of the weight-gradient body (activation, dY and indicator images) + the input gradient's own
    if (!pw.on || pw.MT != 2 || !g_deep_pair || bytes + WG_STATIC > 160 * 1024) return 1;
    static int attr_bytes = 0;
    if (bytes > attr_bytes) {
        HIP_RET(hipFuncSetAttribute(reinterpret_cast<const void*>(&deep_pair_kernel<NKW, NMW, NNW, MAXU, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        attr_bytes = (int)bytes;
    }
    passion_prof_tag("deep_pair_kernel<%d, %d, %d, %d> (conv_deep_kernel<3, 2, ...> + wgrad_lean_kernel<3, 2>)", NKW, NMW, NNW, MAXU);
    pw.on = false;
    hipLaunchKernelGGL((deep_pair_kernel<NKW, NMW, NNW, MAXU, 2>), dim3(gx * gy + pw.gx * pw.gy), dim3(256), (size_t)bytes, st, k, q, pw.k, pw.dW, pw.db, gx, gy, pw.gx);
    LAUNCH_RET();
    return 0;
}

template <int KS, int MODE, int NKW, int NMW, int NNW, int MAXU>
int launch_deep_one(ConvK& k, DeepP& q, int mtiles, hipStream_t st) {
    int nv = 0;
    const long bytes = deep_geometry<KS>(q, NKW, NMW, NNW, &nv);
    if (!bytes || (nv + 63) / 64 > MAXU) return 1;
    const int gx = k.B * q.Od * q.nfx, gy = (mtiles + NMW - 1) / NMW;
    if constexpr (KS == 3 && MODE == 2 && ((NKW == 1 && MAXU == 12) || (NKW == 2 && NMW == 2))) {      // the instantiations the steps use (tools/launch_table.py)
        const int rc = launch_deep_pair<NKW, NMW, NNW, MAXU>(k, q, bytes, gx, gy, st);
        if (rc <= 0) return rc;
    }
    k.fst.stats = nullptr;
    if (MODE == 0 && k.fst_want) {
        const int rc = fwd_stats_setup(k, q.Od * q.nfx, gy, st);
        if (rc) return rc < 0 ? rc : -rc;
    }
    static int attr_bytes = 0;
    if (bytes > attr_bytes) {
        HIP_RET(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_deep_kernel<KS, MODE, NKW, NMW, NNW, MAXU>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        attr_bytes = (int)bytes;
    }
    passion_prof_tag("conv_deep_kernel<%d, %d, %d, %d, %d, %d>", KS, MODE, NKW, NMW, NNW, MAXU);
    hipLaunchKernelGGL((conv_deep_kernel<KS, MODE, NKW, NMW, NNW, MAXU>), dim3(gx, gy), dim3(256), (size_t)bytes, st, k, q);
    LAUNCH_RET();
    return 0;
}

// Which waves do what.  Every workgroup stages its inputs for ALL of its row tiles, so the M axis goes to the waves first (no
// staging repeated per row tile up to 64 rows); the K chunks are then walked one after the other with the next chunk's loads in
// flight.  Only when that leaves the chip short of workgroups do waves move to the K axis (chunks side by side: a shorter chain
// per workgroup, staging repeated per row-tile group) -- level 5, batch-1 decoder launches.
template <int KS, int MODE>
int launch_deep_geo(ConvK& k, DeepP& q, int mtiles, hipStream_t st) {
    static const long want = []() { const char* e = getenv("PASSION_DEEP_WGS"); return e ? atol(e) : 256L; }();
    const int cand[6][3] = {{1, 4, 1}, {1, 2, 2}, {1, 1, 4}, {2, 2, 1}, {2, 1, 2}, {4, 1, 1}};
    int best = -1;
    long best_w = -1;
    for (int i = 0; i < 6; ++i) {
        const int nkw = cand[i][0], nmw = cand[i][1], nnw = cand[i][2];
        if (nmw > 1 && nmw > mtiles) continue;                    // waves without a row tile
        if (nkw > q.nchunk) continue;
        if (i < 3 && nmw < 4 && nmw * 2 <= mtiles) continue;      // K not split: take all the row tiles four waves can hold
        DeepP t = q;
        int nv = 0;
        if (!deep_geometry<KS>(t, nkw, nmw, nnw, &nv)) continue;
        if ((nv + 63) / 64 > (nkw == 4 ? 6 : 12)) continue;
        const long wgs = (long)k.B * t.Od * t.nfx * ((mtiles + nmw - 1) / nmw);
        if (best < 0 || (best_w < want && wgs > best_w)) { best = i; best_w = wgs; }
    }
    if (best < 0) return 1;
    int nv = 0;
    DeepP t = q;
    deep_geometry<KS>(t, cand[best][0], cand[best][1], cand[best][2], &nv);
    const bool small = (nv + 63) / 64 <= 6;
    switch (best) {
        case 0: return launch_deep_one<KS, MODE, 1, 4, 1, 12>(k, q, mtiles, st);
        case 1: return launch_deep_one<KS, MODE, 1, 2, 2, 12>(k, q, mtiles, st);
        case 2: return launch_deep_one<KS, MODE, 1, 1, 4, 12>(k, q, mtiles, st);
        case 3: return small ? launch_deep_one<KS, MODE, 2, 2, 1, 6>(k, q, mtiles, st) : launch_deep_one<KS, MODE, 2, 2, 1, 12>(k, q, mtiles, st);
        case 4: return small ? launch_deep_one<KS, MODE, 2, 1, 2, 6>(k, q, mtiles, st) : launch_deep_one<KS, MODE, 2, 1, 2, 12>(k, q, mtiles, st);
        default: return launch_deep_one<KS, MODE, 4, 1, 1, 6>(k, q, mtiles, st);
    }
}

}  // namespace

namespace pconv {

// 0 = launched, < 0 = HIP error, 1 = shape not covered.  Called from launch_conv_lean with k.Mpad / k.wpack / k.sums_rep set.
int launch_conv_deep(ConvK& k, int ksize, int stride, int mode, hipStream_t st) {
    if (!g_conv_deep || stride != 1 || k.y_f32 || k.bd_rows || k.pwm) return 1;
    if (!(k.slope >= 0.f && k.slope <= 1.f)) return 1;
    if (!((ksize == 3 && (mode == 0 || mode == 2)) || (ksize == 1 && (mode == 0 || mode == 1)))) return 1;
    const bool dg = mode >= 1;
    DeepP q;
    int K;
    if (dg) {
        K = k.Cout;
        if (K % 32) return 1;
        for (int s = 0; s < k.nseg; ++s) if (k.seg[s].C % 16) return 1;         // a 16-row tile lies inside one segment
        if (k.Cin % 16) return 1;
    } else {
        K = k.Cin;
        for (int s = 0; s < k.nseg; ++s) if (k.seg[s].C % 32) return 1;         // a 32-channel chunk lies inside one segment
        if (k.Cout % 16) return 1;
    }
    if (K < 32 || K / 32 > MAXCH) return 1;
    q.nchunk = K / 32;
    q.K8 = K / 8;
    const int pd = mode == 2 ? 2 : 0;
    q.Od = k.Di + pd; q.Oh = k.Hi + pd; q.Ow = k.Wi + pd;
    q.PF = q.Oh * q.Ow;
    q.SH = mode == 2 ? 2 : (ksize == 3 ? 1 : 0);
    // the deep levels only: wide volumes keep their kernels (marching / streaming pointwise / lean bricks are sized for them)
    const long vox = (long)k.Di * k.Hi * k.Wi;
    if (vox > 40000 || q.PF < 16) return 1;
    if (vox * (long)(k.Cin > k.Cout ? k.Cin : k.Cout) >= (1L << 31)) return 1;      // 32-bit element offsets inside a sample
    const int mtiles = k.Mpad / 16;
    if (ksize == 1) return mode == 0 ? launch_deep_geo<1, 0>(k, q, mtiles, st) : launch_deep_geo<1, 1>(k, q, mtiles, st);
    return mode == 0 ? launch_deep_geo<3, 0>(k, q, mtiles, st) : launch_deep_geo<3, 2>(k, q, mtiles, st);
}

}  // namespace pconv
