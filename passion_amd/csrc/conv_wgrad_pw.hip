// Weight gradient of the 1x1x1 convs as a STREAMING kernel (reference general_conv3d_prenorm / general_conv3d with k_size = 1,
// code/models/blocks.py:312-316,366-370; the seg layers and the fusion blocks' 1x1x1 convs, code/models/mmformer.py:90,136-140,337-340):
//   dW[co][ci] = sum_v dY[v][co] * act(x)[v][ci]      dbias[co] = sum_v dY[v][co]      (+ spos[co][ci] = sum_v dY[v][co] [x_hat[v][ci] > 0])
// wgrad_lean_kernel<1, *> (conv_wgrad_lean.hip) does this with the 3x3x3 kernel's machinery: a workgroup per 8-channel block of the
// input (dY is read once per block), one voxel per thread and brick, two workgroup barriers per 256 voxels -- 2.9-4.0 TB/s of
// algorithmic bytes, the kernel RFNet's bench line names (30 launches, 3.6 ms per step) and 1.5 ms of a PASSION step.
// A pointwise weight gradient has no neighbourhood: a sample is V consecutive voxels.  Here a WAVEFRONT owns 64 consecutive voxels
// at a time: its lanes load every input block and the dY rows of those voxels (16-byte loads, consecutive lanes on consecutive
// addresses when the input is one tensor), normalise + LeakyReLU in registers, put the bf16 images into a wavefront-private
// piece of LDS and read both MFMA operands back transposed (ds_read_b64_tr_b16: the K axis of this GEMM is the voxel, the memory
// order is channel-fastest).  No workgroup barrier in the loop -- LDS operations of one wavefront execute in order -- and the loads
// of the next 64 voxels are in flight during the MFMAs.  All input blocks of a voxel are handled by the same wavefront: x and dY
// are read exactly once.  The four wavefronts' partial sums are added in a fixed order at the end; the slot / fold contract of
// the partial slab is the lean kernel's (slots in sample order, spare slots zero), so conv.hip's folds do not change.
#include "conv_common.h"
#include "prof.h"
#include <algorithm>

using namespace pconv;

namespace pconv { int g_wgrad_pw = 1, g_wgrad_pw_minvox = 1 << 20; }      // option "wgrad_pw" / PASSION_WGRAD_PW=0: back to wgrad_lean_kernel<1, *> (A/B)

namespace {

#ifndef PW_U
#define PW_U 1
#endif

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 tr_read8(const char* p0, const char* p1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p1));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// NB = 8-channel blocks of the virtual concat (1, 2, 4, 8), DRN = 16-byte blocks of a dY row that hold channels (1: Cout <= 8, 2: <= 16,
// 4: <= 32), IDENT = the indicator sums too.  A wavefront has PW_U chunks of 64 voxels in flight.  One is the measured optimum: with
// two and four (4-8 KB per wavefront on its way) the registers of the staged chunks cost one to two wavefronts per SIMD and the
// 32 -> 8 layers went from 157 to 218 us over 5 x 128^3 -- occupancy hides more latency here than a deeper prefetch (-DPW_U=n: A/B).
template <int NB, int DRN, bool IDENT>
__global__ __launch_bounds__(256) void wgrad_pw_kernel(ConvK p, float* __restrict__ dW, float* __restrict__ dbias, int wgs_per_b, long vpw) {
    wgrad_clear(p);
    constexpr int MT = DRN == 4 ? 2 : 1;              // 16-row tiles of dY
    constexpr int MC = MT * 16;
    // bytes per voxel row of the dY image: its 16-byte blocks + 16 against bank conflicts (as wgrad_lean_kernel); <= 8 output channels:
    // dense 16-byte rows, and the lanes of rows 8..15 of the tile read a line of zeros that is not part of the image
    constexpr int DYS = DRN == 1 ? 16 : DRN * 16 + 16;
    constexpr int DYZ = DRN == 1 ? 64 : 0;            // the zeros
    constexpr int U = PW_U;                           // chunks in flight per wavefront
    constexpr int NT = IDENT ? NB : (NB + 1) / 2;     // 16-column tiles: (block | its indicator) or (block 2 nt | block 2 nt + 1)
    constexpr int IMG = 64 * 16;                      // one block's image of 64 voxels
    constexpr int NIMG = IDENT ? 2 * NB : NB;
    constexpr int WAVE_LDS = NIMG * IMG + 64 * DYS + DYZ;
    constexpr int RED_TILES = MT * NT + MT;           // accumulator tiles a wavefront hands over at the end (+ the bias tiles)
    static_assert(RED_TILES * 1024 <= WAVE_LDS, "the cross-wave sum reuses the staging memory");
    __shared__ __attribute__((aligned(16))) char lds[4 * WAVE_LDS];

    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = n >> 2, pq = n & 3;
    char* lact = lds + wv * WAVE_LDS;
    char* lind = lact + NB * IMG;                      // IDENT only
    char* ldy = lact + NIMG * IMG;

    const int bx = blockIdx.x;
    const int b = bx / wgs_per_b;                      // >= B: spare workgroup, flushes a zero slot
    const long V = (long)p.Di * p.Hi * p.Wi;
    const long v0 = (long)(bx % wgs_per_b) * vpw;
    const long v1 = (b < p.B) ? (v0 + vpw < V ? v0 + vpw : V) : 0;

    // ---- this lane's input block: block jb of the virtual concat, voxels lane / NB + (64 / NB) r of a chunk
    const int jb = lane % NB, vl0 = lane / NB;
    int C = 8, c0 = 0, sg = 0;
    {
        int r = jb;
        bool found = false;
#pragma unroll
        for (int s2 = 0; s2 < PASSION_MAX_SEG; ++s2) {
            if (s2 < p.nseg && !found) {
                const int nb = p.seg[s2].C / 8;
                if (r < nb) { found = true; sg = s2; c0 = r * 8; C = p.seg[s2].C; }
                r -= nb;
            }
        }
    }
    const bf16_t* src = nullptr;
    bool present = false, has_norm = false;
    float mean[8], rstd[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { mean[c] = 0.f; rstd[c] = 1.f; }
    if (v1 > v0) {
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
                    for (int c = 0; c < 8; ++c) {
                        mean[c] = sd.stats[((long)bs * sd.C + c0 + c) * 2 + 0];
                        rstd[c] = sd.stats[((long)bs * sd.C + c0 + c) * 2 + 1];
                    }
                }
            }
        }
    }
    // ---- this lane's dY block: 16-byte block dl of the rows lane / DRN + (64 / DRN) r
    const int dl = lane % DRN, dvl0 = lane / DRN, dco = dl * 8;
    const long bo = (long)(b < p.B ? b : 0) * V * p.Cout;
    const bf16_t* dyb = reinterpret_cast<const bf16_t*>(p.dy) + bo;
    const float* dyf = reinterpret_cast<const float*>(p.dy) + bo;      // p.y_f32: the f32 logit gradients (Cout = 4 or 8)
    auto load_dy = [&](long vox) -> u32x4 {
        if (p.y_f32) {
            const float* f = dyf + vox * p.Cout + dco;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(f);
            f32x4 hi = {0.f, 0.f, 0.f, 0.f};
            if (dco + 8 <= p.Cout) hi = *reinterpret_cast<const f32x4*>(f + 4);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) { o[j] = (__bf16)lo[j]; o[4 + j] = (__bf16)hi[j]; }
            return __builtin_bit_cast(u32x4, o);
        }
        return *reinterpret_cast<const u32x4*>(dyb + vox * p.Cout + dco);
    };

    // ---- per-lane constants of the MFMA loop (K step ks: voxels 32 ks + 8 g + 4 half + q, as the lean kernel's brick rows)
    const int alane = (g * 8 + q) * DYS + pq * 8;
    const bool azero = DRN == 1 && pq >= 2;            // rows 8..15 of a <= 8-channel dY tile
    int boff[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        int img;
        if (IDENT) img = (pq >> 1) ? NB + j : j;                      // columns 8..15: the indicator image of block j
        else { img = 2 * j + (pq >> 1); if (img >= NB) img = NB - 1; }     // NB = 1: padding half-tile, dropped at the flush
        boff[j] = img * IMG + (g * 8 + q) * 16 + (pq & 1) * 8;
    }
    f32x4 acc[MT][NT], accb[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        accb[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;

    // the dY rows' columns past Cout (and the 16 bytes of padding) are never written: zero them once
    for (int i = lane; i < (64 * DYS + DYZ) / 16; i += 64) *reinterpret_cast<u32x4*>(ldy + i * 16) = (u32x4){0u, 0u, 0u, 0u};

    u32x4 ra[U][NB], rd[U][DRN];
    unsigned okbits = 0u;                              // bit u * NB + r: ra[u][r] holds a voxel of a present, in-range row
    auto fetch = [&](long cv) {                        // cv = first voxel of the U chunks
        okbits = 0u;
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                const long vox = cv + 64 * u + vl0 + (64 / NB) * r;
                ra[u][r] = (u32x4){0u, 0u, 0u, 0u};
                if (present && vox < v1) {
                    ra[u][r] = *reinterpret_cast<const u32x4*>(src + vox * C);
                    okbits |= 1u << (u * NB + r);
                }
            }
#pragma unroll
            for (int r = 0; r < DRN; ++r) {
                const long vox = cv + 64 * u + dvl0 + (64 / DRN) * r;
                rd[u][r] = (u32x4){0u, 0u, 0u, 0u};
                if (vox < v1 && dco < p.Cout) rd[u][r] = load_dy(vox);
            }
        }
    };
    auto commit = [&](const u32x4 (&xa)[NB], const u32x4 (&xd)[DRN], unsigned ok) {
#pragma unroll
        for (int r = 0; r < NB; ++r) {
            const int vl = vl0 + (64 / NB) * r;
            u32x4 o = xa[r], iv = (u32x4){0u, 0u, 0u, 0u};
            if (has_norm && ((ok >> r) & 1u)) {
                bf16x8 ob;
#pragma unroll
                for (int c = 0; c < 4; ++c) {      // bf16 pairs -> f32: low half = even channel
                    const float x0 = __uint_as_float(xa[r][c] << 16), x1 = __uint_as_float(xa[r][c] & 0xffff0000u);
                    float t0 = (x0 - mean[2 * c]) * rstd[2 * c], t1 = (x1 - mean[2 * c + 1]) * rstd[2 * c + 1];
                    t0 = fmaxf(t0, t0 * p.slope);      // LeakyReLU for 0 <= slope <= 1
                    t1 = fmaxf(t1, t1 * p.slope);
                    ob[2 * c] = (__bf16)t0; ob[2 * c + 1] = (__bf16)t1;
                }
                o = __builtin_bit_cast(u32x4, ob);
                if (IDENT) {
                    // [x_hat > 0] from the bf16 images: LeakyReLU and the rounding keep the sign, a positive bf16 is a positive int16.
                    // (inline assembly: clang 19 turns the vector builtins into compare / select / permute per half -- and
                    // __builtin_bit_cast of a vector ELEMENT reads element 0 whatever the index)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const unsigned wd = o[c];
                        unsigned pos, one, ind;
                        asm("v_pk_max_i16 %0, %1, 0" : "=v"(pos) : "v"(wd));
                        asm("v_pk_min_u16 %0, %1, %2" : "=v"(one) : "v"(pos), "v"(0x00010001u));
                        asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(ind) : "v"(one), "v"(0x3f803f80u));      // bf16 1.0
                        iv[c] = ind;
                    }
                }
            }
            *reinterpret_cast<u32x4*>(lact + jb * IMG + vl * 16) = o;
            if (IDENT) *reinterpret_cast<u32x4*>(lind + jb * IMG + vl * 16) = iv;
        }
#pragma unroll
        for (int r = 0; r < DRN; ++r) *reinterpret_cast<u32x4*>(ldy + (dvl0 + (64 / DRN) * r) * DYS + dl * 16) = xd[r];
    };

    // ---- runs wv, wv + 4, ... of 64 U voxels of this workgroup's range
    long cv = v0 + 64 * U * wv;
    if (cv < v1) fetch(cv);
    for (; cv < v1; cv += 256 * U) {
        u32x4 ca[U][NB], cd[U][DRN];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int r = 0; r < NB; ++r) ca[u][r] = ra[u][r];
#pragma unroll
            for (int r = 0; r < DRN; ++r) cd[u][r] = rd[u][r];
        }
        const unsigned cok = okbits;
        if (cv + 256 * U < v1) fetch(cv + 256 * U);    // in flight during the U chunks below
#pragma unroll
        for (int u = 0; u < U; ++u) {
            commit(ca[u], cd[u], cok >> (u * NB));     // (the reads of the previous chunk were issued before these writes: in order)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const char* ap = ldy + alane + ks * 32 * DYS;
                bf16x8 a[MT];
                if constexpr (DRN == 1) {
                    const char* az = ldy + 64 * DYS + (lane & 7) * 8;
                    a[0] = tr_read8(azero ? az : ap, azero ? az : ap + 4 * DYS);
                } else {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) a[mt] = tr_read8(ap + mt * 32, ap + mt * 32 + 4 * DYS);
                }
                const char* bp = lact + ks * 32 * 16;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const bf16x8 bv = tr_read8(bp + boff[j], bp + boff[j] + 4 * 16);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], bv, acc[mt][j], 0, 0, 0);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) accb[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], ones, accb[mt], 0, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }

    // ---- the four wavefronts hold partial sums over disjoint voxels of the SAME outputs: add them through LDS in a fixed order
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);      // [4 waves][RED_TILES][64 lanes][4]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int j = 0; j < NT; ++j) *reinterpret_cast<f32x4*>(red + ((wv * RED_TILES + mt * NT + j) * 64 + lane) * 4) = acc[mt][j];
        *reinterpret_cast<f32x4*>(red + ((wv * RED_TILES + MT * NT + mt) * 64 + lane) * 4) = accb[mt];
    }
    __syncthreads();
    // wavefront w sums and flushes the tiles t = w, w + 4, ...
    const long nWf = (long)p.Cout * p.Cin;
    float* slot = dW ? dW + (long)bx * p.wslot : nullptr;
    for (int t = wv; t < RED_TILES; t += 4) {
        f32x4 s = *reinterpret_cast<const f32x4*>(red + ((0 * RED_TILES + t) * 64 + lane) * 4);
#pragma unroll
        for (int w2 = 1; w2 < 4; ++w2) {
            const f32x4 o = *reinterpret_cast<const f32x4*>(red + ((w2 * RED_TILES + t) * 64 + lane) * 4);
            s[0] += o[0]; s[1] += o[1]; s[2] += o[2]; s[3] += o[3];
        }
        if (t >= MT * NT) {                             // bias tile: every column holds the sums
            const int mt = t - MT * NT;
            if (dbias && n == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = mt * 16 + g * 4 + r;
                    if (co < p.Cout) dbias[(long)bx * p.wslot + co] = s[r];
                }
            }
            continue;
        }
        const int mt = t / NT, j = t % NT;
        const int blk = IDENT ? j : 2 * j + (n >> 3);
        if (blk >= NB) continue;                        // padding half-tile (NB = 1)
        int ci = 0;                                     // channel of the virtual concat behind (block blk, column n & 7)
        {
            int r = blk;
            bool found = false;
#pragma unroll
            for (int s2 = 0; s2 < PASSION_MAX_SEG; ++s2) {
                if (s2 < p.nseg && !found) {
                    const int nb = p.seg[s2].C / 8;
                    if (r < nb) { found = true; ci = p.seg[s2].cbase + r * 8 + (n & 7); }
                    r -= nb;
                }
            }
        }
        float* dst = slot;
        if (IDENT && (n >> 3)) dst = p.spos + (long)bx * nWf;
        if (dst) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = mt * 16 + g * 4 + r;
                if (co < p.Cout) dst[(long)co * p.Cin + ci] = s[r];
            }
        }
    }
}

// Workgroups of a sample that sum voxels: as many as are resident at once (a second, partial round of workgroups was a third of the
// launch time at one sample: 1024 workgroups on 768 places); the slots behind them are flushed as zeros by workgroups that return at once.
template <int NB, int DRN, bool IDENT>
int launch_one(ConvK& k, int gx, float* dW, float* db, int wgs_per_b, hipStream_t st) {
    static int resident = 0;
    if (!resident) {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, wgrad_pw_kernel<NB, DRN, IDENT>, 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        resident = per_cu * (cus > 0 ? cus : 256);
    }
    static const int pct = getenv("PASSION_WPW_ACT") ? atoi(getenv("PASSION_WPW_ACT")) : 100;      // percent of the resident places (A/B)
    const int act = std::max(1, std::min(wgs_per_b, (int)((long)resident * pct / 100) / k.B));
    const long V = (long)k.Di * k.Hi * k.Wi;
    const long vpw = ((V + act - 1) / act + 255) / 256 * 256;
    hipLaunchKernelGGL((wgrad_pw_kernel<NB, DRN, IDENT>), dim3(gx), dim3(256), 0, st, k, dW, db, wgs_per_b, vpw);
    LAUNCH_RET();
    return 0;
}

template <int NB, int DRN>
int launch_nb(ConvK& k, bool ident, int gx, float* dW, float* db, int wgs_per_b, hipStream_t st) {
    return ident ? launch_one<NB, DRN, true>(k, gx, dW, db, wgs_per_b, st) : launch_one<NB, DRN, false>(k, gx, dW, db, wgs_per_b, st);
}

template <int DRN>
int launch_dr(ConvK& k, int K8, bool ident, int gx, float* dW, float* db, int wgs_per_b, hipStream_t st) {
    if (K8 == 1) return launch_nb<1, DRN>(k, ident, gx, dW, db, wgs_per_b, st);
    if (K8 == 2) return launch_nb<2, DRN>(k, ident, gx, dW, db, wgs_per_b, st);
    if (K8 == 4) return launch_nb<4, DRN>(k, ident, gx, dW, db, wgs_per_b, st);
    return launch_nb<8, DRN>(k, ident, gx, dW, db, wgs_per_b, st);
}

}  // namespace

namespace pconv {

// the shapes the streaming kernel takes (the planner asks too: it gives these launches 1024 slots whatever the block count)
bool wgrad_pw_covers(const ConvK& k, bool y_f32) {
    if (!g_wgrad_pw) return false;
    if (k.Cout % 8 && !(y_f32 && k.Cout % 4 == 0 && k.Cout <= 8)) return false;
    if (pad16(k.Cout) > 32 || !(k.slope >= 0.f && k.slope <= 1.f)) return false;
    int K8 = 0;
    for (int s = 0; s < k.nseg; ++s) {
        if (k.seg[s].C % 8) return false;
        K8 += k.seg[s].C / 8;
    }
    // wide volumes only: below ~1 M voxels (levels 3..5) and at 64 input channels over less than 128^3 x 2 the lean kernel's split over
    // the input blocks (more, smaller workgroups) is ahead: 64 -> 16 over 4 x 64^3 21 us against 30, 16.8 MB launches 11 against 13.5
    // (option "wgrad_pw_minvox": the unit tests set 0)
    const long vox = (long)k.B * k.Di * k.Hi * k.Wi;
    if (vox < g_wgrad_pw_minvox) return false;
    return K8 == 1 || K8 == 2 || K8 == 4 || (K8 == 8 && vox >= 4L * g_wgrad_pw_minvox);
}

// 0 = launched, < 0 = HIP error, 1 = not this shape.  (gx, gy) is the lean kernel's plan: gx slots, sample b owns wgs_per_b of them
int launch_wgrad_pw(ConvK& k, int MT, int gx, int gy, float* dW, float* db, hipStream_t st) {
    (void)gy;
    if (!wgrad_pw_covers(k, k.y_f32 != 0) || MT != pad16(k.Cout) / 16) return 1;
    int K8 = 0;
    for (int s = 0; s < k.nseg; ++s) K8 += k.seg[s].C / 8;
    k.Mpad = pad16(k.Cout);
    const int tiles_per_b = k.tilesD * k.tilesH * k.tilesW;
    const int wgs = gx / k.B;
    k.per = (tiles_per_b + wgs - 1) / wgs;             // conv.hip derives the slots per sample from these two, as for the lean kernel
    const int wgs_per_b = (tiles_per_b + k.per - 1) / k.per;
    const bool ident = k.spos != nullptr;
    const int drn = k.Cout <= 8 ? 1 : (k.Cout <= 16 ? 2 : 4);
    passion_prof_tag("wgrad_pw_kernel<%d, %d, %d>", K8, drn, (int)ident);
    if (drn == 1) return launch_dr<1>(k, K8, ident, gx, dW, db, wgs_per_b, st);
    if (drn == 2) return launch_dr<2>(k, K8, ident, gx, dW, db, wgs_per_b, st);
    return launch_dr<4>(k, K8, ident, gx, dW, db, wgs_per_b, st);
}

}  // namespace pconv
