// bf16 matrix-core version of the fused (InstanceNorm -> LeakyReLU ->) Conv3d forward / dgrad.
// Same implicit GEMM as conv.hip (D[channel][voxel] += W[channel][K] * act[K][voxel], K = (tap, cin)),
// but on v_mfma_f32_16x16x32_bf16: one MFMA eats K = 32 = four (tap, 8-channel block) pairs.
//
//  * LDS brick is voxel-major [voxel][CKB channels] bf16: the 8 consecutive k of a lane's B fragment are
//    the 8 channels of one voxel at one tap = ONE ds_read_b128, no transposes anywhere.
//  * lane group g = lane >> 4 owns pair q = 4*step + g -> (tap = q / (CKB/8), block = q % (CKB/8)); every
//    lane computes its own tap offset (the LDS address is per-lane anyway), so a 3x3x3 conv over 8
//    channels is 7 MFMA steps per column tile instead of 54 f32 steps.
//  * weights stream from L2 as [tap][cin/8][coutPad][8] bf16: one 16-byte load per lane per step,
//    1 KiB contiguous per wave, prefetched one step ahead.
//  * the loader normalises / activates / masks / pads each input element once and writes it back to LDS
//    with one ds_write_b128 per 8 channels.
// The accumulator layout equals the f32 kernel's, so the epilogues (conv_common.h) are shared.
#include "conv_common.h"
#include "prof.h"

namespace pconv { extern int g_conv_merge, g_wgrad_wide, g_conv_lean; }
using namespace pconv;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KS, int S, bool DG, int CKB> struct GeoB {
    static constexpr int HALO = (KS == 3) ? 1 : 0;
    static constexpr int ID = DG ? TD + 2 * HALO : (TD - 1) * S + KS;
    static constexpr int IH = DG ? TH + 2 * HALO : (TH - 1) * S + KS;
    static constexpr int IW = DG ? TW + 2 * HALO : (TW - 1) * S + KS;
    static constexpr int VOL = ID * IH * IW;
    static constexpr int NC8 = CKB / 8;
    static constexpr int VS = CKB * 2 + (CKB > 8 ? 16 : 0);  // bytes per voxel (padded: spreads 16-B slots over banks)
    static constexpr int NTAP = KS * KS * KS;
    static constexpr int Q = NTAP * NC8;
    static constexpr int NSTEP = (Q + 3) / 4;
};

// weight fragments of one chunk in LDS: NSTEP * 4 lane groups * MT*16 rows * 16 B; kept when <= 56 KB
__host__ __device__ constexpr bool wlds_fits(int ckb, int mt) { return ckb == 8 || mt <= 2; }
template <int KS, int S, bool DG, int CKB> __host__ __device__ constexpr int brick_bytes() {
    return (GeoB<KS, S, DG, CKB>::VOL * GeoB<KS, S, DG, CKB>::VS + 15) / 16 * 16;
}
template <int KS, int S, bool DG, int CKB, int MT> constexpr int conv_lds_bytes() {
    return brick_bytes<KS, S, DG, CKB>() + (wlds_fits(CKB, MT) ? GeoB<KS, S, DG, CKB>::NSTEP * 4 * MT * 16 * 16 : 0);
}

__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (__bf16)0.f;
    return z;
}

// Stage one brick into lds as [voxel][CKB] bf16.  Thread t fills 8-channel block j = t % NC8 of every voxel it
// visits; the caller says where that block comes from (tensor of the virtual concat, channel offset cb, how many
// of the 8 channels exist, its InstanceNorm statistics) -- nch = 0 stages zeros (absent modality / past the end).
template <int KS, int S, bool DG, int CKB>
__device__ __forceinline__ void stage_tile_b(char* lds, const bf16_t* __restrict__ src, int C, int cb, int nch,
                                             const float* __restrict__ stats, float slope, int pad_mode, int b,
                                             int o0d, int o0h, int o0w, int Sd, int Sh, int Sw, int tid) {
    using G = GeoB<KS, S, DG, CKB>;
    const int j = tid % G::NC8;                 // 8-channel block of this thread (constant: NC8 divides 256)
    const bool has_norm = (stats != nullptr);
    float mean[8], rstd[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        mean[c] = 0.f; rstd[c] = 1.f;
        if (has_norm && c < nch) {
            mean[c] = stats[((long)b * C + cb + c) * 2 + 0];
            rstd[c] = stats[((long)b * C + cb + c) * 2 + 1];
        }
    }
    for (int it = tid; it < G::VOL * G::NC8; it += 256) {
        const int tv = it / G::NC8;
        const int zw = tv % G::IW, zh = (tv / G::IW) % G::IH, zd = tv / (G::IW * G::IH);
        int sd, sh, sw;
        bool ok = true;
        if (!DG) {
            sd = o0d * S - G::HALO + zd; sh = o0h * S - G::HALO + zh; sw = o0w * S - G::HALO + zw;
            if (pad_mode == PASSION_PAD_REFLECT) {
                sd = reflect_idx(sd, Sd); sh = reflect_idx(sh, Sh); sw = reflect_idx(sw, Sw);
                sd = min(max(sd, 0), Sd - 1); sh = min(max(sh, 0), Sh - 1); sw = min(max(sw, 0), Sw - 1);
            } else {
                ok = (sd >= 0 && sd < Sd && sh >= 0 && sh < Sh && sw >= 0 && sw < Sw);
            }
        } else {
            const int ud = o0d - G::HALO + zd, uh = o0h - G::HALO + zh, uw = o0w - G::HALO + zw;
            ok = (ud >= 0 && uh >= 0 && uw >= 0 && (ud % S) == 0 && (uh % S) == 0 && (uw % S) == 0);
            sd = ud / S; sh = uh / S; sw = uw / S;
            ok = ok && (sd < Sd && sh < Sh && sw < Sw);
        }
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = 0.f;
        if (ok && nch > 0) {
            const bf16_t* ptr = src + (((long)sd * Sh + sh) * Sw + sw) * C + cb;
            if (nch == 8) {
                const f32x4 lo = load4(ptr), hi = load4(ptr + 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) { v[c] = lo[c]; v[4 + c] = hi[c]; }
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) if (c < nch) v[c] = to_f32(ptr[c]);
            }
            if (has_norm) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    float t = (v[c] - mean[c]) * rstd[c];
                    t = t > 0.f ? t : t * slope;
                    v[c] = (c < nch) ? t : 0.f;
                }
            }
        }
        bf16x8 o;
#pragma unroll
        for (int c = 0; c < 8; ++c) o[c] = (__bf16)v[c];
        *reinterpret_cast<bf16x8*>(lds + tv * G::VS + j * 16) = o;
    }
}

template <int KS, int S, int MT, int MODE, int CKB>
__global__ __launch_bounds__(256) void conv_bf16_kernel(ConvK p) {
    constexpr bool DG = (MODE == 1);
    using G = GeoB<KS, S, DG, CKB>;
    // dynamic LDS: [input brick | this chunk's weight fragments (when they fit: WLDS)]
    extern __shared__ __attribute__((aligned(16))) char dyn_lds[];
    char* lds = dyn_lds;
    constexpr bool WLDS = wlds_fits(CKB, MT);
    constexpr int WV = G::NSTEP * 4 * MT * 16;          // 16-byte weight vectors per chunk
    bf16x8* wl = reinterpret_cast<bf16x8*>(dyn_lds + brick_bytes<KS, S, DG, CKB>());
    __shared__ float lsum[DG ? 4 * MT * 16 * 2 : 5 * MT * 16 * 2];     // forward: [wave][row][2] + the workgroup's totals (fused statistics)

    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, n = lane & 15, g = lane >> 4;
    int bid = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tw_ = bid % p.tilesW; bid /= p.tilesW;
    const int th_ = bid % p.tilesH; bid /= p.tilesH;
    const int td_ = bid % p.tilesD; bid /= p.tilesD;
    const int b = bid;
    const int o0d = td_ * TD, o0h = th_ * TH, o0w = tw_ * TW;
    const int mg = blockIdx.y;

    const int Od = DG ? p.Di : p.Do, Oh = DG ? p.Hi : p.Ho, Ow = DG ? p.Wi : p.Wo;
    const int Sd = DG ? p.Do : p.Di, Sh = DG ? p.Ho : p.Hi, Sw = DG ? p.Wo : p.Wi;
    int K8 = pad8(p.Cout) / 8;                                      // 8-channel blocks along K in the packed image
    if (!DG) { K8 = 0; for (int s2 = 0; s2 < p.nseg; ++s2) K8 += (p.seg[s2].C + 7) / 8; }
    const bf16x8* wp = reinterpret_cast<const bf16x8*>(p.wpack) + wvec_of(p, b);      // grouped launch: this sample's weight set

    f32x4 acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int lh[4] = {0 + (n >> 3), 2 + (n >> 3), 4 + (n >> 3), 6 + (n >> 3)};
    const int lw = n & 7, ld_ = wv;
    // byte offset of this lane's four voxels (tap 0) inside the brick
    int pos[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
        pos[nt] = DG ? ((ld_ * G::IH + lh[nt]) * G::IW + lw) * G::VS
                     : (((ld_ * S) * G::IH + lh[nt] * S) * G::IW + lw * S) * G::VS;

    const bool reflect_dg = DG && (KS == 3) && (p.pad_mode == PASSION_PAD_REFLECT);
    const bool lowD = reflect_dg && (o0d == 0) && (Od >= 2), lowH = reflect_dg && (o0h == 0) && (Oh >= 2),
               lowW = reflect_dg && (o0w == 0) && (Ow >= 2);
    const bool hiD = reflect_dg && p.hi_t[0] >= 0 && (o0d <= Od - 2) && (Od - 2 < o0d + TD);
    const bool hiH = reflect_dg && p.hi_t[1] >= 0 && (o0h <= Oh - 2) && (Oh - 2 < o0h + TH);
    const bool hiW = reflect_dg && p.hi_t[2] >= 0 && (o0w <= Ow - 2) && (Ow - 2 < o0w + TW);
    const bool any_mirror = lowD || lowH || lowW || hiD || hiH || hiW;
    // list of this brick's mirror tap combos, in a fixed order (ballot + prefix count over waves 0 and 1)
    __shared__ unsigned short mlist[128];
    __shared__ int mcnt[4];
    __shared__ unsigned char mtab[3][5][8];   // [axis d,h,w][option][local coordinate] = tap | valid << 2 | brick position << 3
    int nmirror = 0;
    if (DG && KS == 3 && any_mirror) {       // workgroup-uniform
        if (tid < 3 * 5 * 8) {
            const int ax = tid / 40, a = (tid / 8) % 5, il = tid % 8;
            const int i0 = ax == 0 ? o0d : (ax == 1 ? o0h : o0w), Sn = ax == 0 ? Od : (ax == 1 ? Oh : Ow);
            int t, pos; bool v;
            axis_tap<KS, G::HALO, DG, S>(a, il, i0, Sn, p.hi_t[ax], p.hi_u[ax], t, pos, v);
            mtab[ax][a][il] = (unsigned char)((t & 3) | (v ? 4 : 0) | ((pos & 31) << 3));
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

    // K runs over the 8-channel blocks of the virtual concat (dgrad: of dY); a chunk = NC8 consecutive blocks, which
    // may come from different tensors (decoder skip concat [8|8], the four modality maps of a fusion conv [8|8|8|8])
    auto locate = [&](int gb, int& sg, int& c0) -> bool {      // global block -> (segment, channel offset); false: past the end
        if (DG) { sg = 0; c0 = gb * 8; return gb < K8; }
        for (int s2 = 0; s2 < p.nseg; ++s2) {
            const int nb = (p.seg[s2].C + 7) / 8;
            if (gb < nb) { sg = s2; c0 = gb * 8; return true; }
            gb -= nb;
        }
        sg = 0; c0 = 0;
        return false;
    };
    {
        for (int gb0 = 0; gb0 < K8; gb0 += G::NC8) {
            const int cb8 = gb0;
            unsigned vmask = 0;                      // which of this chunk's blocks exist and are present
#pragma unroll
            for (int j2 = 0; j2 < G::NC8; ++j2) {
                int sg2, c02;
                bool ok = locate(gb0 + j2, sg2, c02);
                if (ok && !DG && p.seg[sg2].mask && !p.seg[sg2].mask[(long)b * p.seg[sg2].mask_ld + p.seg[sg2].mask_col]) ok = false;
                vmask |= (ok ? 1u : 0u) << j2;
            }
            if (vmask == 0) continue;               // e.g. an absent modality: contributes exactly zero
            // this thread's source block
            int sgj, c0j;
            const int jj = tid % G::NC8;
            const bool okj = locate(gb0 + jj, sgj, c0j) && ((vmask >> jj) & 1u);
            const int Cj = DG ? p.Cout : p.seg[sgj].C;
            const bf16_t* srcj = DG ? reinterpret_cast<const bf16_t*>(p.dy) + (long)b * Sd * Sh * Sw * Cj
                                    : reinterpret_cast<const bf16_t*>(p.seg[sgj].x) + (long)p.seg[sgj].bsrc(b) * p.seg[sgj].bstride;
            const float* statsj = DG ? nullptr : p.seg[sgj].stats;
            const int nchj = okj ? min(8, Cj - c0j) : 0;
            __syncthreads();
            stage_tile_b<KS, S, DG, CKB>(lds, srcj, Cj, c0j, nchj, statsj, p.slope, p.pad_mode, DG ? b : p.seg[sgj].bsrc(b), o0d, o0h, o0w, Sd, Sh, Sw, tid);
            if (WLDS) {
                for (int v = tid; v < WV; v += 256) {
                    const int row = v % (MT * 16), gg = (v / (MT * 16)) % 4, s = v / (MT * 16 * 4);
                    const int q = 4 * s + gg, tap = q / G::NC8, blk = q % G::NC8;
                    const bool ok = (q < G::Q) && ((vmask >> blk) & 1u);
                    wl[v] = ok ? wp[((long)tap * K8 + cb8 + blk) * p.Mpad + (mg * MT) * 16 + row] : zero8();
                }
            }
            __syncthreads();

            // ---- regular taps: 4 (tap, block) pairs per MFMA step, one per lane group
            auto decode = [&](int s, int& toff, long& wrow) -> bool {
                const int q = 4 * s + g;
                const int tap = q / G::NC8, blk = q % G::NC8;
                const bool v = (q < G::Q) && ((vmask >> blk) & 1u);
                int td = 0, th = 0, tw = 0;
                if (KS == 3) { td = tap / 9; th = (tap / 3) % 3; tw = tap % 3; }
                if (DG && KS == 3) { td = 2 - td; th = 2 - th; tw = 2 - tw; }   // gather form: voxel i reads u = i + 1 - t
                toff = v ? ((td * G::IH + th) * G::IW + tw) * G::VS + blk * 16 : 0;
                wrow = ((long)(v ? tap : 0) * K8 + cb8 + (v ? blk : 0)) * p.Mpad + (mg * MT) * 16 + n;
                return v;
            };
            if (WLDS) {
                // A fragments come from LDS: the whole chunk's weights were fetched by all 256 threads at once
                // (many loads in flight) instead of one dependent L2 round trip per MFMA step
                for (int s = 0; s < G::NSTEP; ++s) {
                    int toff; long wrow;
                    decode(s, toff, wrow);
                    bf16x8 a[MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) a[mt] = wl[((s * 4 + g) * MT + mt) * 16 + n];
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const bf16x8 bv = *reinterpret_cast<const bf16x8*>(lds + pos[nt] + toff);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], bv, acc[mt][nt], 0, 0, 0);
                    }
                }
            } else {
                int toff; long wrow;
                bool v = decode(0, toff, wrow);
                bf16x8 a[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a[mt] = v ? wp[wrow + mt * 16] : zero8();
                for (int s = 0; s < G::NSTEP; ++s) {
                    int toff_n = 0; long wrow_n = 0; bool v_n = false;
                    bf16x8 a_n[MT];
                    if (s + 1 < G::NSTEP) {
                        v_n = decode(s + 1, toff_n, wrow_n);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) a_n[mt] = v_n ? wp[wrow_n + mt * 16] : zero8();
                    }
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const bf16x8 bv = *reinterpret_cast<const bf16x8*>(lds + pos[nt] + toff);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], bv, acc[mt][nt], 0, 0, 0);
                    }
                    toff = toff_n;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) a[mt] = a_n[mt];
                }
            }

            // ---- mirror taps (adjoint of reflect padding); only bricks touching planes 1 / Sin-2.  The brick's
            // list of (tap, block) pairs is packed four per MFMA step exactly like the regular taps.
            if (nmirror > 0) {
                const int P = nmirror * G::NC8;
                // decode pair 4*s2 + g -> (mirror combo, 8-channel block): weight fragment, per-voxel LDS offsets, validity
                auto prep = [&](int s2, bf16x8 (&am)[MT], int (&mpos)[4], bool (&mval)[4]) {
                    const int pr = s2 * 4 + g;
                    const int ent = pr / G::NC8, blk = pr % G::NC8;
                    const bool pv = (ent < nmirror) && ((vmask >> blk) & 1u);
                    const int code = mlist[pv ? ent : 0];
                    // per-axis (tap, brick position, valid) come from the tables built once per workgroup
                    const int cd = mtab[0][code & 7][ld_], cw = mtab[2][(code >> 6) & 7][lw];
                    const int ah = (code >> 3) & 7;
                    int thp = 0;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const int ch = mtab[1][ah][lh[nt]];
                        thp = ch & 3;
                        mval[nt] = pv && (cd & 4) && (ch & 4) && (cw & 4);
                        mpos[nt] = mval[nt] ? (((cd >> 3) * G::IH + (ch >> 3)) * G::IW + (cw >> 3)) * G::VS + blk * 16 : 0;
                    }
                    const int tap = ((cd & 3) * KS + thp) * KS + (cw & 3);
                    // the mirror terms reuse the weights of regular taps: with WLDS they are already in LDS
                    // (pair q = tap * NC8 + blk sits at wl[(q * MT + mt) * 16 + row])
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        if (WLDS) am[mt] = pv ? wl[((tap * G::NC8 + blk) * MT + mt) * 16 + n] : zero8();
                        else am[mt] = pv ? wp[((long)tap * K8 + cb8 + blk) * p.Mpad + (mg * MT + mt) * 16 + n] : zero8();
                    }
                };
                bf16x8 am[MT]; int mpos[4]; bool mval[4];
                prep(0, am, mpos, mval);
                for (int s2 = 0; s2 * 4 < P; ++s2) {
                    bf16x8 am_n[MT]; int mpos_n[4]; bool mval_n[4];
                    const bool more = (s2 + 1) * 4 < P;
                    if (more) prep(s2 + 1, am_n, mpos_n, mval_n);       // weight loads of the next step fly during these MFMAs
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        bf16x8 bv = zero8();
                        if (mval[nt]) bv = *reinterpret_cast<const bf16x8*>(lds + mpos[nt]);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[mt], bv, acc[mt][nt], 0, 0, 0);
                    }
                    if (more) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) am[mt] = am_n[mt];
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) { mpos[nt] = mpos_n[nt]; mval[nt] = mval_n[nt]; }
                    }
                }
            }
        }
    }

    const int oh4[4] = {o0h + lh[0], o0h + lh[1], o0h + lh[2], o0h + lh[3]};
    if (!DG) {
        if (p.fst.stats) {        // workgroup-uniform: statistics of y from this epilogue (one brick = one slot)
            __syncthreads();
            epilogue_fwd_stats<MT>(p, acc, lsum, b, (td_ * p.tilesH + th_) * p.tilesW + tw_, mg, o0d + ld_, o0w + lw, oh4, g, n, wv, tid);
        } else {
            epilogue_fwd<bf16_t, MT>(p, acc, b, mg, o0d + ld_, o0w + lw, oh4, g);
        }
    } else epilogue_dgrad<bf16_t, MT>(p, acc, lsum, b, mg, o0d + ld_, o0w + lw, oh4, g, n, tid);
}

// ------------------------------------------------------------------------------------------------
// wgrad on bf16 matrix cores:  dW[co][ci][tap] += sum_vox dY[vox][co] * act(x)[vox*s + tap][ci]
// D[m = co][n = (tap, ci)] with K = 32 voxels per MFMA.  Both operands are k-strided in their voxel-major
// LDS images ([vox][co] and [vox][ci]), so both fragments come from ds_read_b64_tr_b16: a 16-lane group
// reads a block of 4 voxels x 16 columns and every lane receives ITS column for the 4 voxels; lane 4q+p of
// the group supplies the address of (voxel q, columns 4p..4p+3), which lets one 16-column tile mix two taps
// of an 8-channel chunk (columns 0-7: tap 2t, columns 8-15: tap 2t+1) with no copies.
// A workgroup owns (a strided set of bricks) x (one 8-channel chunk of one segment) x (MT row tiles); partial
// sums stay in registers across bricks and are flushed once with f32 atomics.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 tr_read8(const char* p0, const char* p1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p1));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

template <int KS, int S, int MT, int CKW>
__global__ __launch_bounds__(256) void conv_wgrad_bf16_kernel(ConvK p, float* __restrict__ dW, float* __restrict__ dbias,
                                                              int tiles_per_b) {
    wgrad_clear(p);
    using G = GeoB<KS, S, false, CKW>;
    constexpr int NC8 = CKW / 8;
    constexpr int NTAP = G::NTAP;
    // 16 columns = 2 taps x 8 channels (CKW 8) or 1 tap x 16 channels (CKW 16; the two 8-channel blocks of a
    // chunk may belong to different tensors of the virtual concat, e.g. the decoder's [upsampled | skip] pair)
    constexpr int NTILES = (CKW == 8) ? (NTAP + 1) / 2 : NTAP;
    constexpr int KSPLIT = (NTILES >= 4) ? 1 : 4;     // few column tiles: waves split the voxels instead
    constexpr int NTW = (KSPLIT == 1) ? (NTILES + 3) / 4 : NTILES;
    constexpr int MC = MT * 16;
    constexpr int DYS = MC * 2 + 16;                  // bytes per voxel row of the dY image
    __shared__ __attribute__((aligned(16))) char lact[G::VOL * G::VS];
    __shared__ __attribute__((aligned(16))) char ldy[TD * TH * TW * DYS];

    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int q = n >> 2, pq = n & 3;
    const int mgroups = p.Mpad / 16 / MT;
    const int mg = blockIdx.y % mgroups;
    const int gb0 = (blockIdx.y / mgroups) * NC8;     // first global 8-channel block of this workgroup's chunk
    auto locate = [&](int gb, int& sg, int& c0) -> bool {
        for (int s2 = 0; s2 < p.nseg; ++s2) {
            const int nb = (p.seg[s2].C + 7) / 8;
            if (gb < nb) { sg = s2; c0 = gb * 8; return true; }
            gb -= nb;
        }
        sg = 0; c0 = 0;
        return false;
    };
    // this thread stages block jj of every voxel it visits
    const int jj = tid % NC8;
    int sgj, c0j;
    const bool existj = locate(gb0 + jj, sgj, c0j);
    const SegD& sd = p.seg[sgj];
    const int nch = existj ? min(8, sd.C - c0j) : 0;
    const bool has_norm = existj && (sd.stats != nullptr);

    // per column tile of this wave: byte offset of this lane's (tap, channel quad) inside a voxel neighbourhood
    int boff[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int nt = (KSPLIT == 1) ? wv + 4 * j : j;
        int tap = (CKW == 8) ? 2 * nt + (pq >> 1) : nt;
        if (tap >= NTAP) tap = 0;                      // padded column: reads finite data, dropped at the flush
        const int tw = tap % KS, th = (tap / KS) % KS, td = tap / (KS * KS);
        boff[j] = ((td * G::IH + th) * G::IW + tw) * G::VS + ((CKW == 8) ? (pq & 1) : pq) * 8;
    }
    f32x4 acc[MT][NTW], accb[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        accb[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const bool bias_block = (dbias != nullptr) && (blockIdx.y / mgroups == 0);
    const bool do_bias = bias_block && (KSPLIT == 4 || wv == 0);
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;

    // Software pipeline over this workgroup's bricks: the global loads of brick t+1 (raw activations + dY) are issued
    // into registers before the MFMA loop of brick t and only normalised / written to LDS after it, so the HBM
    // round trip hides under the matrix work instead of serialising with it.
    constexpr int AR = (G::VOL * NC8 + 255) / 256;   // 16-byte activation blocks per thread
    constexpr int DR = MC / 8;                       // dY 16-byte blocks per thread
    struct Regs { bf16x8 a[AR]; bf16x8 d[DR]; unsigned ok; int b; bool present; };
    auto fetch = [&](int tile, Regs& R) {
        int t2 = tile;
        const int tw_ = t2 % p.tilesW; t2 /= p.tilesW;
        const int th_ = t2 % p.tilesH; t2 /= p.tilesH;
        const int td_ = t2 % p.tilesD; t2 /= p.tilesD;
        const int b = t2;
        const int o0d = td_ * TD, o0h = th_ * TH, o0w = tw_ * TW;
        R.b = b;
        R.present = existj && (sd.mask ? (sd.mask[(long)b * sd.mask_ld + sd.mask_col] != 0) : true);
        R.ok = 0u;
        const bf16_t* src = reinterpret_cast<const bf16_t*>(sd.x) + (long)sd.bsrc(b) * sd.bstride;
#pragma unroll
        for (int r = 0; r < AR; ++r) {
            const int tv = (tid + 256 * r) / NC8;
            R.a[r] = zero8();
            if (tv >= G::VOL || !R.present) continue;
            const int zw = tv % G::IW, zh = (tv / G::IW) % G::IH, zd = tv / (G::IW * G::IH);
            int s_d = o0d * S - G::HALO + zd, s_h = o0h * S - G::HALO + zh, s_w = o0w * S - G::HALO + zw;
            bool ok = true;
            if (p.pad_mode == PASSION_PAD_REFLECT) {
                s_d = reflect_idx(s_d, p.Di); s_h = reflect_idx(s_h, p.Hi); s_w = reflect_idx(s_w, p.Wi);
                s_d = min(max(s_d, 0), p.Di - 1); s_h = min(max(s_h, 0), p.Hi - 1); s_w = min(max(s_w, 0), p.Wi - 1);
            } else {
                ok = (s_d >= 0 && s_d < p.Di && s_h >= 0 && s_h < p.Hi && s_w >= 0 && s_w < p.Wi);
            }
            if (!ok) continue;
            const bf16_t* ptr = src + (((long)s_d * p.Hi + s_h) * p.Wi + s_w) * sd.C + c0j;
            if (nch == 8) {
                R.a[r] = *reinterpret_cast<const bf16x8*>(ptr);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (e < nch) R.a[r][e] = ptr[e];
            }
            R.ok |= (1u << r);
        }
        const bf16_t* dy = reinterpret_cast<const bf16_t*>(p.dy) + (long)b * p.Do * p.Ho * p.Wo * p.Cout;
#pragma unroll
        for (int r = 0; r < DR; ++r) {
            const int i = tid + 256 * r;
            const int c8 = i % DR, v = i / DR;
            const int w = v % TW, h = (v / TW) % TH, d = v / (TW * TH);
            const int od = o0d + d, oh = o0h + h, ow = o0w + w;
            const int co = mg * MC + c8 * 8;
            R.d[r] = zero8();
            if (od < p.Do && oh < p.Ho && ow < p.Wo && co < p.Cout) {
                const bf16_t* ptr = dy + ((((long)od) * p.Ho + oh) * p.Wo + ow) * p.Cout + co;
                if (co + 8 <= p.Cout) {
                    R.d[r] = *reinterpret_cast<const bf16x8*>(ptr);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (co + e < p.Cout) R.d[r][e] = ptr[e];
                }
            }
        }
    };
    auto commit = [&](const Regs& R) {
        float mean[8], rstd[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            mean[c] = 0.f; rstd[c] = 1.f;
            if (has_norm && c < nch) {
                mean[c] = sd.stats[((long)sd.bsrc(R.b) * sd.C + c0j + c) * 2 + 0];
                rstd[c] = sd.stats[((long)sd.bsrc(R.b) * sd.C + c0j + c) * 2 + 1];
            }
        }
#pragma unroll
        for (int r = 0; r < AR; ++r) {
            const int tv = (tid + 256 * r) / NC8;
            if (tv >= G::VOL) continue;
            bf16x8 o = zero8();
            if ((R.ok >> r) & 1u) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    float t = (float)R.a[r][c];
                    if (has_norm) {
                        t = (t - mean[c]) * rstd[c];
                        t = t > 0.f ? t : t * p.slope;
                    }
                    o[c] = (__bf16)((c < nch) ? t : 0.f);
                }
            }
            *reinterpret_cast<bf16x8*>(lact + tv * G::VS + jj * 16) = o;
        }
#pragma unroll
        for (int r = 0; r < DR; ++r) {
            const int i = tid + 256 * r;
            *reinterpret_cast<bf16x8*>(ldy + (i / DR) * DYS + (i % DR) * 16) = R.d[r];
        }
    };

    // Grouped launch (p.bgroup > 0: G = B / bgroup weight sets, the stride-2 layers of the four modality encoders as ONE launch): the
    // gridDim.x slots are dealt to the sets in equal runs (slots [g * per, (g + 1) * per) sum bricks of set g's samples only), so the
    // grouped fold finds every set's partial sums back to back.  The launcher makes gridDim.x a multiple of G.
    int total_tiles = p.B * tiles_per_b, tstep = (int)gridDim.x;
    int tile = xcd_swizzle(blockIdx.x, gridDim.x);   // per round of gridDim.x bricks every XCD works on one contiguous run
    if (p.bgroup > 0) {
        const int ngrp = p.B / p.bgroup, per = (int)gridDim.x / ngrp, grp = (int)blockIdx.x / per;
        tile = grp * p.bgroup * tiles_per_b + (int)blockIdx.x % per;
        total_tiles = (grp + 1) * p.bgroup * tiles_per_b;
        tstep = per;
    }
    bool have = tile < total_tiles;
    Regs R;
    if (have) fetch(tile, R);
    while (have) {
        __syncthreads();
        commit(R);
        __syncthreads();
        const bool work = true;   // absent blocks were staged as zeros: their columns flush zeros
        const int next = tile + tstep;
        const bool have_n = next < total_tiles;
        Regs Rn;
        if (have_n) fetch(next, Rn);
        // K loop: 32 voxels per MFMA = one (d, 4 h-rows) slab; lane group g owns h-row g, its two tr-reads
        // cover w = 0..3 and 4..7
        for (int ks = (KSPLIT == 4 ? wv : 0); work && ks < TD * TH * TW / 32; ks += KSPLIT) {
            const int d = ks >> 1, h = (ks & 1) * 4 + g, w = q;
            const int v = (d * TH + h) * TW + w;
            const char* ap = ldy + v * DYS + pq * 8;
            bf16x8 a[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[mt] = tr_read8(ap + mt * 32, ap + mt * 32 + 4 * DYS);
            const char* bp = lact + (((d * S) * G::IH + h * S) * G::IW + w * S) * G::VS;
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                const bf16x8 bv = tr_read8(bp + boff[j], bp + boff[j] + 4 * S * G::VS);
                if (KSPLIT == 1 && wv + 4 * j >= NTILES) continue;   // wave-uniform
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], bv, acc[mt][j], 0, 0, 0);
            }
            if (do_bias) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    accb[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], ones, accb[mt], 0, 0, 0);
            }
        }
        tile = next; have = have_n;
        if (have_n) R = Rn;
    }
    if (KSPLIT == 4) {
        // the four waves hold partial sums over disjoint voxels of the SAME outputs: combine through LDS (fixed order)
        __syncthreads();
        float* red = reinterpret_cast<float*>(ldy);          // [4 waves][MT][NTW + 1][64 lanes][4]
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int j = 0; j < NTW; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[(((wv * MT + mt) * (NTW + 1) + j) * 64 + lane) * 4 + r] = acc[mt][j][r];
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(((wv * MT + mt) * (NTW + 1) + NTW) * 64 + lane) * 4 + r] = accb[mt][r];
        }
        __syncthreads();
        if (wv != 0) return;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int j = 0; j <= NTW; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = 0.f;
#pragma unroll
                    for (int w2 = 0; w2 < 4; ++w2) t += red[(((w2 * MT + mt) * (NTW + 1) + j) * 64 + lane) * 4 + r];
                    if (j < NTW) acc[mt][j][r] = t; else accb[mt][r] = t;
                }
        }
    }
    if (dW) {
        dW += (long)blockIdx.x * p.wslot;                    // this workgroup's slot of the partial-sum slab
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int nt = (KSPLIT == 1) ? wv + 4 * j : j;
            const int tap = (CKW == 8) ? 2 * nt + (n >> 3) : nt;
            const int blk = (CKW == 8) ? 0 : (n >> 3), c = n & 7;
            int sgb, c0b;
            if (nt >= NTILES || tap >= NTAP || !locate(gb0 + blk, sgb, c0b) || c0b + c >= p.seg[sgb].C) continue;
            const long ci = p.seg[sgb].cbase + c0b + c;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = (mg * MT + mt) * 16 + g * 4 + r;
                    const float v = acc[mt][j][r];
                    if (co < p.Cout) dW[((long)co * p.Cin + ci) * NTAP + tap] = v;
                }
        }
    }
    if (bias_block && wv == 0 && n == 0) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = (mg * MT + mt) * 16 + g * 4 + r;
                if (co < p.Cout) dbias[(long)blockIdx.x * p.wslot + co] = accb[mt][r];
            }
    }
}

template <int KS, int S, int CKW>
int launch_wgrad_b(ConvK& k, int MT, int gx, int gy, float* dW, float* db, hipStream_t st) {
    const int tiles_per_b = k.tilesD * k.tilesH * k.tilesW;
    dim3 grid(gx, gy);
    passion_prof_tag("conv_wgrad_bf16_kernel<%d, %d, %d, %d>", KS, S, MT == 2 ? 2 : 1, CKW);
    if (MT == 2) hipLaunchKernelGGL((conv_wgrad_bf16_kernel<KS, S, 2, CKW>), grid, dim3(256), 0, st, k, dW, db, tiles_per_b);
    else hipLaunchKernelGGL((conv_wgrad_bf16_kernel<KS, S, 1, CKW>), grid, dim3(256), 0, st, k, dW, db, tiles_per_b);
    LAUNCH_RET();
    return 0;
}

// weight f32 [Cout][Cin][ntap] -> bf16 fwd image [tap][CinPad8/8][CoutPad16][8] and dgrad image
// [tap][CoutPad8/8][CinPad16][8]
__global__ void pack_weight_bf16_kernel(const float* __restrict__ w, int Cin, int Cout, int ntap, bf16_t* __restrict__ fwd,
                                        bf16_t* __restrict__ dg) {
    const int ci8 = pad8(Cin), co16 = pad16(Cout), co8 = pad8(Cout), ci16 = pad16(Cin);
    const long nf = (long)ntap * ci8 * co16, nd = (long)ntap * co8 * ci16;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nf + nd; i += (long)gridDim.x * blockDim.x) {
        if (i < nf) {
            const int e = (int)(i % 8);
            const int co = (int)((i / 8) % co16);
            const int cb = (int)((i / (8L * co16)) % (ci8 / 8));
            const int tap = (int)(i / (8L * co16 * (ci8 / 8)));
            const int ci = cb * 8 + e;
            fwd[i] = (bf16_t)((co < Cout && ci < Cin) ? w[((long)co * Cin + ci) * ntap + tap] : 0.f);
        } else {
            const long k = i - nf;
            const int e = (int)(k % 8);
            const int ci = (int)((k / 8) % ci16);
            const int cb = (int)((k / (8L * ci16)) % (co8 / 8));
            const int tap = (int)(k / (8L * ci16 * (co8 / 8)));
            const int co = cb * 8 + e;
            dg[k] = (bf16_t)((co < Cout && ci < Cin) ? w[((long)co * Cin + ci) * ntap + tap] : 0.f);
        }
    }
}

template <int KS, int S, int MT, int MODE, int CKB>
int launch_one(ConvK& k, dim3 grid, hipStream_t st) {
    constexpr int bytes = conv_lds_bytes<KS, S, MODE == 1, CKB, MT>();
    static bool attr_set = false;     // > 64 KB of dynamic LDS has to be granted once per kernel
    if (!attr_set) {
        HIP_RET(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_kernel<KS, S, MT, MODE, CKB>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        attr_set = true;
    }
    passion_prof_tag("conv_bf16_kernel<%d, %d, %d, %d, %d>", KS, S, MT, MODE, CKB);
    k.fst.stats = nullptr;
    if (MODE == 0 && k.fst_want) {          // one brick per workgroup: tiles per sample slots, grid.y row-tile groups
        const int rc = fwd_stats_setup(k, k.tilesD * k.tilesH * k.tilesW, (int)grid.y, st);
        if (rc) return rc < 0 ? rc : -rc;
    }
    hipLaunchKernelGGL((conv_bf16_kernel<KS, S, MT, MODE, CKB>), grid, dim3(256), bytes, st, k);
    LAUNCH_RET();
    return 0;
}

template <int KS, int S, int MODE, int CKB>
int launch_mt(ConvK& k, int mtiles, hipStream_t st) {
    const int tiles = k.B * k.tilesD * k.tilesH * k.tilesW;
    int MT = mtiles >= 4 ? 4 : mtiles;
    if (mtiles % MT) MT = (mtiles % 2 == 0) ? 2 : 1;
    while (MT > 1 && (long)tiles * (mtiles / MT) < 512) MT >>= 1;
    dim3 grid(tiles, mtiles / MT);
    if (MT == 4) return launch_one<KS, S, 4, MODE, CKB>(k, grid, st);
    if (MT == 2) return launch_one<KS, S, 2, MODE, CKB>(k, grid, st);
    return launch_one<KS, S, 1, MODE, CKB>(k, grid, st);
}

using pconv::g_conv_merge;

template <int KS, int S, int MODE>
int launch_ckb(ConvK& k, int ktot, int mtiles, hipStream_t st) {
    // chunk width = as much of K as exists, up to 32 channels (one brick staging + one barrier pair per chunk)
    if (ktot >= 32) return launch_mt<KS, S, MODE, 32>(k, mtiles, st);
    if (ktot >= 16 && g_conv_merge) return launch_mt<KS, S, MODE, 16>(k, mtiles, st);
    return launch_mt<KS, S, MODE, 8>(k, mtiles, st);
}

template <int MODE>
int launch_geo(ConvK& k, int ksize, int stride, int ktot, int mtiles, hipStream_t st) {
    if (ksize == 1) return launch_ckb<1, 1, MODE>(k, ktot, mtiles, st);
    if (stride == 2) {
        // forward stride-2 brick is 9x17x17 voxels: only the 8-channel chunk fits LDS; dgrad stages a zero-stuffed 6x10x10 brick
        if constexpr (MODE == 0) return launch_mt<3, 2, MODE, 8>(k, mtiles, st);
        else return launch_ckb<3, 2, MODE>(k, ktot, mtiles, st);
    }
    return launch_ckb<3, 1, MODE>(k, ktot, mtiles, st);
}

}  // namespace

namespace pconv {

extern int g_conv_merge, g_wgrad_wide;

long bf16_pack_elems(int Cin, int Cout, int ntap) {
    return (long)ntap * pad8(Cin) * pad16(Cout) + (long)ntap * pad8(Cout) * pad16(Cin);
}

int pack_weight_bf16(const float* w, int Cin, int Cout, int ntap, void* out, hipStream_t st) {
    const long nf = (long)ntap * pad8(Cin) * pad16(Cout);
    const long total = bf16_pack_elems(Cin, Cout, ntap);
    const int blocks = (int)hmin((total + 255) / 256, 1024);
    hipLaunchKernelGGL(pack_weight_bf16_kernel, dim3(blocks), dim3(256), 0, st, w, Cin, Cout, ntap,
                       reinterpret_cast<bf16_t*>(out), reinterpret_cast<bf16_t*>(out) + nf);
    LAUNCH_RET();
    return 0;
}

// All-bf16 launches only; every segment must have C % 8 == 0 unless it is the only one.
bool bf16_eligible(const ConvK& k, bool dgrad) {
    if (k.nseg == 1) return dgrad ? true : true;
    for (int s = 0; s < k.nseg; ++s)
        if (k.seg[s].C % 8) return false;
    return true;
}

int launch_conv_bf16(ConvK& k, int ksize, int stride, bool dgrad, const void* packed_bf16, hipStream_t st) {
    if (!dgrad) {                            // first encoder layer: one input channel, eight outputs (conv_c1_mfma.hip; other shapes: conv_c1_fwd.hip)
        int rc = launch_conv_c1m_fwd(k, ksize, stride, packed_bf16, st);
        if (rc <= 0) return rc;
        rc = launch_conv_c1_fwd(k, ksize, stride, packed_bf16, st);
        if (rc <= 0) return rc;
        rc = launch_conv_s2f(k, ksize, stride, packed_bf16, st);      // first stride-2 layer (8 -> 16): marching kernel (conv_s2_fwd.hip)
        if (rc <= 0) return rc;
    }
    if (g_conv_lean) {                       // lean persistent kernel for the stride-1, 8-channel-aligned shapes
        const int rc = launch_conv_lean(k, ksize, stride, dgrad ? 1 : 0, packed_bf16, st);
        if (rc <= 0) return rc;              // > 0: shape not covered, use the general kernel below
    }
    const int ntap = ksize * ksize * ksize;
    int cmax = 0;
    if (dgrad) {
        cmax = k.Cout;
        k.Mpad = pad16(k.Cin);
        k.wpack = reinterpret_cast<const bf16_t*>(packed_bf16) + (long)ntap * pad8(k.Cin) * pad16(k.Cout);
    } else {
        // chunk width: measured on MI355X, merging several 8-channel tensors into one 16/32-channel chunk LOSES
        // (3x the LDS per workgroup -> fewer resident workgroups -> less latency hiding: [8|8]->8 @128^3 86 -> 129 us),
        // so the width follows the widest single tensor; PASSION_CONV_MERGE=1 switches to the total for A/B runs
        for (int s = 0; s < k.nseg; ++s) {
            if (g_conv_merge) cmax += pad8(k.seg[s].C);
            else cmax = k.seg[s].C > cmax ? k.seg[s].C : cmax;
        }
        k.Mpad = pad16(k.Cout);
        k.wpack = packed_bf16;
    }
    const int mtiles = k.Mpad / 16;
    return dgrad ? launch_geo<1>(k, ksize, stride, cmax, mtiles, st) : launch_geo<0>(k, ksize, stride, cmax, mtiles, st);
}

int g_conv_dbg = 0;       // developer timing switch (PASSION_CONV_DBG): bit0 skip staging, bit1 skip MFMA, bit2 skip epilogue
int g_conv_target = 1024; // persistent conv workgroups aimed at per launch (PASSION_CONV_TARGET)
int g_pack_f32 = 1;       // option "pack_f32": 0 = weight packs write the bf16 operand images only (conv.hip: pack_multi_kernel, launch_conv_geo)
int g_conv_lean = 1;      // PASSION_CONV_LEAN=0: general kernel everywhere (A/B)
int g_conv_merge = 0, g_wgrad_wide = 0;   // A/B switches (environment: PASSION_CONV_MERGE, PASSION_WGRAD_WIDE)

int wgrad_bf16_chunk(const ConvK& k, int ksize, int stride) {
    int ktot = 0;
    for (int s = 0; s < k.nseg; ++s) ktot += pad8(k.seg[s].C);
    return (g_wgrad_wide && ktot >= 16 && stride == 1) ? 16 : 8;     // stride-2 bricks (9x17x17) only fit LDS 8 channels at a time
}

int launch_wgrad_bf16(ConvK& k, int ksize, int stride, int MT, int gx, int gy, float* dW, float* db, hipStream_t st) {
    k.Mpad = pad16(k.Cout);
    const bool wide = wgrad_bf16_chunk(k, ksize, stride) == 16;
    if (g_conv_lean && !wide) {
        const int rc = launch_wgrad_lean(k, ksize, stride, MT, gx, gy, dW, db, st);
        if (rc <= 0) return rc;
    }
    if (ksize == 1) return wide ? launch_wgrad_b<1, 1, 16>(k, MT, gx, gy, dW, db, st) : launch_wgrad_b<1, 1, 8>(k, MT, gx, gy, dW, db, st);
    if (stride == 1) return wide ? launch_wgrad_b<3, 1, 16>(k, MT, gx, gy, dW, db, st) : launch_wgrad_b<3, 1, 8>(k, MT, gx, gy, dW, db, st);
    return launch_wgrad_b<3, 2, 8>(k, MT, gx, gy, dW, db, st);
}

}  // namespace pconv
