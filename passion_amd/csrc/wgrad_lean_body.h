// The body of wgrad_lean_kernel (conv_wgrad_lean.hip) as a device function of (workgroup x, workgroup y, workgroups in x), so that a
// launch can hold it beside ANOTHER kernel's body: conv_deep.hip runs the weight gradient and the input gradient of a deep-level conv
// (levels 3..5: a few MB, 64..256 workgroups each, both bound by their latency chain, neither fills the chip) as two workgroup ranges of
// ONE grid -- they read the same (x, dY) and do not depend on each other (reference code/models/mmformer.py:40-64,66-114,116-189).
#pragma once
#include "conv_common.h"

namespace pconv_wl {
using namespace pconv;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 tr_read8(const char* p0, const char* p1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p1));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

template <int KS, int MT>
__device__ __forceinline__ void wgrad_lean_body(const ConvK& p, float* __restrict__ dW, float* __restrict__ dbias, const int bxi, const int byi, const int gxn) {
    constexpr int HALO = (KS == 3) ? 1 : 0;
    constexpr int ID = TD + 2 * HALO, IH = TH + 2 * HALO, IW = TW + 2 * HALO, VOL = ID * IH * IW;
    constexpr int VS = 16;                            // bytes per voxel of the activation image (8 channels)
    constexpr int NTAP = KS * KS * KS;
    constexpr int NTILES = (NTAP + 1) / 2;            // 16 columns = 2 taps x 8 channels
    constexpr int KSPLIT = (NTILES >= 4) ? 1 : 4;     // few column tiles: the waves split the voxels instead
    constexpr int NTW = (KSPLIT == 1) ? (NTILES + 3) / 4 : NTILES;
    constexpr int MC = MT * 16;
    constexpr int DYS = MC * 2 + 16;                  // bytes per voxel row of the dY image
    constexpr int AR = (VOL + 255) / 256;             // activation voxels per thread
    constexpr int DR = MC / 8;                        // dY 16-byte blocks per thread
    constexpr int NKS = TD * TH * TW / 32;            // K steps per brick
    __shared__ __attribute__((aligned(16))) char lact[VOL * VS];
    __shared__ __attribute__((aligned(16))) char ldy[TD * TH * TW * DYS];
    // KS == 1 with p.spos: the 0/1 image of x_hat > 0 (present blocks only) -- the B operand of columns 8..15, whose sums give
    // sum_v dY[v][co] [x_hat[v][ci] > 0]; the bias sums then come from one more MFMA per K step against a tile of ones
    __shared__ __attribute__((aligned(16))) char lind[KS == 1 ? VOL * VS : 16];
    const bool ident = KS == 1 && p.spos != nullptr;

    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: branches on it stay scalar
    const int q = n >> 2, pq = n & 3;
    const int mgroups = p.Mpad / 16 / MT;
    const int mg = byi % mgroups;
    const int gb0 = byi / mgroups;             // this workgroup's 8-channel block of the virtual concat
    const int Sd = p.Di, Sh = p.Hi, Sw = p.Wi;        // stride 1: input and output volumes coincide

    const int tiles_per_b = p.tilesD * p.tilesH * p.tilesW;
    const int wgs_per_b = (tiles_per_b + p.per - 1) / p.per;
    const int bx = xcd_swizzle(bxi, gxn);
    const int b = bx / wgs_per_b;                     // >= B: spare workgroup, flushes a zero slot
    const int t0 = (bx % wgs_per_b) * p.per;
    const int nbricks = (b < p.B) ? max(0, min(p.per, tiles_per_b - t0)) : 0;

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
    float mean[8], rstd[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { mean[c] = 0.f; rstd[c] = 1.f; }
    if (exist && nbricks > 0) {
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
    const bf16_t* dyb = reinterpret_cast<const bf16_t*>(p.dy) + (long)(b < p.B ? b : 0) * Sd * Sh * Sw * p.Cout;
    const float* dyf = reinterpret_cast<const float*>(p.dy) + (long)(b < p.B ? b : 0) * Sd * Sh * Sw * p.Cout;   // p.y_f32
    // dY block of 8 channels at (voxel, co): bf16 16-byte load, or (f32 logit heads, Cout <= 8) converted on the fly
    auto load_dy = [&](long vox, int co) -> u32x4 {
        if (p.y_f32) {
            const float* f = dyf + vox * p.Cout + co;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(f);
            f32x4 hi = {0.f, 0.f, 0.f, 0.f};
            if (co + 8 <= p.Cout) hi = *reinterpret_cast<const f32x4*>(f + 4);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) { o[j] = (__bf16)lo[j]; o[4 + j] = (__bf16)hi[j]; }
            return __builtin_bit_cast(u32x4, o);
        }
        return *reinterpret_cast<const u32x4*>(dyb + vox * p.Cout + co);
    };

    // ---- per-thread constants of the staging
    int voxoff[AR];                                   // activation voxel of iteration r relative to the brick's first voxel
#pragma unroll
    for (int r = 0; r < AR; ++r) {
        const int tv = tid + 256 * r;
        const int zw = tv % IW, zh = (tv / IW) % IH, zd = tv / (IW * IH);
        voxoff[r] = (zd * Sh + zh) * Sw + zw;
    }
    int dvoff[DR], dco[DR], dlds[DR];                 // dY: voxel offset, channel offset, LDS byte offset
#pragma unroll
    for (int r = 0; r < DR; ++r) {
        const int i = tid + 256 * r;
        const int c8 = i % DR, v = i / DR;
        const int w = v % TW, h = (v / TW) % TH, d = v / (TW * TH);
        dvoff[r] = (d * Sh + h) * Sw + w;
        dco[r] = mg * MC + c8 * 8;
        dlds[r] = v * DYS + c8 * 16;
    }
    // ---- per-lane constants of the MFMA loop (K step ks: voxels d = ks >> 1, h = (ks & 1) * 4 + g, w = 4 * half + q)
    const int alane = (g * TW + q) * DYS + pq * 8;
    const int blane = (g * IW + q) * VS;
    int boff[NTW];                                    // column tile j: this lane's (tap, channel quad) inside a neighbourhood
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int nt = (KSPLIT == 1) ? wv + 4 * j : j;
        int tap = 2 * nt + (pq >> 1);
        if (tap >= NTAP) tap = 0;                      // padded column: reads finite data, dropped at the flush
        const int tw = tap % KS, th = (tap / KS) % KS, td = tap / (KS * KS);
        boff[j] = blane + ((td * IH + th) * IW + tw) * VS + (pq & 1) * 8;
        if (ident && (pq >> 1)) boff[j] = (int)(lind - lact) + blane + (pq & 1) * 8;      // second half of the tile: the indicator image
    }
    f32x4 accb[MT];                                   // ident: the bias sums (tile of ones)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) accb[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 acc[MT][NTW];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // dbias[co] = sum_vox dY[vox][co] rides along for free: NTAP is odd, so columns 8..15 of the last column tile are
    // padding -- their B operand is set to ones and the accumulator columns 8..15 of that tile come out as the bias sums.
    static_assert(NTAP % 2 == 1, "bias trick needs a padding half-tile");
    constexpr int JB = (KSPLIT == 1) ? (NTILES - 1) / 4 : NTILES - 1;   // tile slot and ...
    constexpr int WB = (KSPLIT == 1) ? (NTILES - 1) % 4 : -1;           // ... wave that own the last tile (-1: every wave)
    const bool bias_block = (dbias != nullptr) && (gb0 == 0);
    const bool bias_wave = (WB < 0 || wv == WB);
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;

    // ---- staging: FETCH (global -> registers) / COMMIT (normalise -> LDS)
    u32x4 ra[AR], rd[DR];
    unsigned okbits = 0u;
    auto fetch = [&](int o0d, int o0h, int o0w) {
        okbits = 0u;
        const bool interior = o0d >= HALO && o0h >= HALO && o0w >= HALO && o0d + TD + HALO <= Sd && o0h + TH + HALO <= Sh &&
                              o0w + TW + HALO <= Sw;
        if (interior) {                               // workgroup-uniform: no padding, no partial brick
            const int basevox = ((o0d - HALO) * Sh + (o0h - HALO)) * Sw + (o0w - HALO);
#pragma unroll
            for (int r = 0; r < AR; ++r) {
                ra[r] = (u32x4){0u, 0u, 0u, 0u};
                if (present && tid + 256 * r < VOL) {
                    ra[r] = *reinterpret_cast<const u32x4*>(src + (long)(basevox + voxoff[r]) * C);
                    okbits |= 1u << r;
                }
            }
            const int baseo = (o0d * Sh + o0h) * Sw + o0w;
#pragma unroll
            for (int r = 0; r < DR; ++r) {
                rd[r] = (u32x4){0u, 0u, 0u, 0u};
                if (dco[r] < p.Cout) rd[r] = load_dy((long)(baseo + dvoff[r]), dco[r]);
            }
        } else {
#pragma unroll
            for (int r = 0; r < AR; ++r) {
                const int tv = tid + 256 * r;
                const int zw = tv % IW, zh = (tv / IW) % IH, zd = tv / (IW * IH);
                int s_d = o0d - HALO + zd, s_h = o0h - HALO + zh, s_w = o0w - HALO + zw;
                bool ok = present && tv < VOL;
                if (p.pad_mode == PASSION_PAD_REFLECT) {
                    s_d = reflect_idx(s_d, Sd); s_h = reflect_idx(s_h, Sh); s_w = reflect_idx(s_w, Sw);
                    s_d = min(max(s_d, 0), Sd - 1); s_h = min(max(s_h, 0), Sh - 1); s_w = min(max(s_w, 0), Sw - 1);
                } else {
                    ok = ok && (s_d >= 0 && s_d < Sd && s_h >= 0 && s_h < Sh && s_w >= 0 && s_w < Sw);
                }
                ra[r] = (u32x4){0u, 0u, 0u, 0u};
                if (ok) {
                    ra[r] = *reinterpret_cast<const u32x4*>(src + (((long)s_d * Sh + s_h) * Sw + s_w) * C);
                    okbits |= 1u << r;
                }
            }
#pragma unroll
            for (int r = 0; r < DR; ++r) {
                const int v = (tid + 256 * r) / DR;
                const int w = v % TW, h = (v / TW) % TH, d = v / (TW * TH);
                const int od = o0d + d, oh = o0h + h, ow = o0w + w;
                rd[r] = (u32x4){0u, 0u, 0u, 0u};
                if (od < Sd && oh < Sh && ow < Sw && dco[r] < p.Cout)
                    rd[r] = load_dy(((long)od * Sh + oh) * Sw + ow, dco[r]);
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int r = 0; r < AR; ++r) {
            if (tid + 256 * r >= VOL) continue;
            u32x4 o = ra[r];
            if (has_norm && ((okbits >> r) & 1u)) {
                bf16x8 ob;
#pragma unroll
                for (int c = 0; c < 4; ++c) {      // bf16 pairs -> f32: low half = even channel
                    const float v0 = __uint_as_float(ra[r][c] << 16), v1 = __uint_as_float(ra[r][c] & 0xffff0000u);
                    float t0_ = (v0 - mean[2 * c]) * rstd[2 * c], t1_ = (v1 - mean[2 * c + 1]) * rstd[2 * c + 1];
                    t0_ = fmaxf(t0_, t0_ * p.slope);      // LeakyReLU for 0 <= slope <= 1: one max instead of compare + select
                    t1_ = fmaxf(t1_, t1_ * p.slope);
                    ob[2 * c] = (__bf16)t0_; ob[2 * c + 1] = (__bf16)t1_;
                }
                o = __builtin_bit_cast(u32x4, ob);
            }
            *reinterpret_cast<u32x4*>(lact + (tid + 256 * r) * VS) = o;
            if (KS == 1 && ident) {
                u32x4 iv = (u32x4){0u, 0u, 0u, 0u};
                if (has_norm && ((okbits >> r) & 1u)) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float v0 = __uint_as_float(ra[r][c] << 16), v1 = __uint_as_float(ra[r][c] & 0xffff0000u);
                        const unsigned i0 = (v0 - mean[2 * c]) * rstd[2 * c] > 0.f ? 0x3f80u : 0u;            // bf16 1.0
                        const unsigned i1 = (v1 - mean[2 * c + 1]) * rstd[2 * c + 1] > 0.f ? 0x3f80u : 0u;
                        iv[c] = i0 | (i1 << 16);
                    }
                }
                *reinterpret_cast<u32x4*>(lind + (tid + 256 * r) * VS) = iv;
            }
        }
#pragma unroll
        for (int r = 0; r < DR; ++r) *reinterpret_cast<u32x4*>(ldy + dlds[r]) = rd[r];
    };

    // ---- bricks t0 .. t0 + nbricks - 1 of sample b, coordinates advanced incrementally
    int tw_ = t0 % p.tilesW, th_ = (t0 / p.tilesW) % p.tilesH, td_ = t0 / (p.tilesW * p.tilesH);
    if (nbricks > 0) fetch(td_ * TD, th_ * TH, tw_ * TW);
    for (int i = 0; i < nbricks; ++i) {
        __syncthreads();
        commit();
        __syncthreads();
        if (++tw_ == p.tilesW) { tw_ = 0; if (++th_ == p.tilesH) { th_ = 0; ++td_; } }
        if (i + 1 < nbricks) fetch(td_ * TD, th_ * TH, tw_ * TW);    // loads of the next brick fly during the MFMAs below
#pragma unroll
        for (int kk = 0; kk < NKS / KSPLIT; ++kk) {
            // KSPLIT 1: ks = kk (compile-time offsets); KSPLIT 4: ks = wv + 4 kk (per-wave offsets)
            const int ks = (KSPLIT == 1) ? kk : wv + 4 * kk;
            const int kd = ks >> 1, kh = (ks & 1) * 4;
            const char* ap = ldy + alane + ((kd * TH + kh) * TW) * DYS;
            bf16x8 a[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[mt] = tr_read8(ap + mt * 32, ap + mt * 32 + 4 * DYS);
            const char* bp = lact + ((kd * IH + kh) * IW) * VS;
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                bf16x8 bv = tr_read8(bp + boff[j], bp + boff[j] + 4 * VS);
                if (j == JB && bias_wave && n >= 8 && !ident) bv = ones;        // padding half of the last tile (the lane RECEIVES column n)
                // (tiles past NTILES on waves 2/3 are computed and dropped at the flush: no branch around an MFMA)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], bv, acc[mt][j], 0, 0, 0);
            }
            if (KS == 1 && ident) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) accb[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], ones, accb[mt], 0, 0, 0);
            }
        }
    }

    if (KSPLIT == 4) {
        // the four waves hold partial sums over disjoint voxels of the SAME outputs: combine through LDS (fixed order)
        __syncthreads();
        float* red = reinterpret_cast<float*>(ldy);          // [4 waves][MT][NTW][64 lanes][4]
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int j = 0; j < NTW; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[(((wv * MT + mt) * NTW + j) * 64 + lane) * 4 + r] = acc[mt][j][r];
        }
        float* redb = red + 4 * MT * NTW * 64 * 4;           // ident: [4 waves][MT][64 lanes][4] bias partials
        if (KS == 1 && ident) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) redb[((wv * MT + mt) * 64 + lane) * 4 + r] = accb[mt][r];
        }
        __syncthreads();
        if (wv != 0) return;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int j = 0; j < NTW; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = 0.f;
#pragma unroll
                    for (int w2 = 0; w2 < 4; ++w2) t += red[(((w2 * MT + mt) * NTW + j) * 64 + lane) * 4 + r];
                    acc[mt][j][r] = t;
                }
            if (KS == 1 && ident) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = 0.f;
#pragma unroll
                    for (int w2 = 0; w2 < 4; ++w2) t += redb[((w2 * MT + mt) * 64 + lane) * 4 + r];
                    accb[mt][r] = t;
                }
            }
        }
    }
    // ---- this workgroup's slot of the partial-sum slab (plain stores; reduce_parts_kernel folds the slots)
    if (dW && exist) {
        float* slot = dW + (long)bx * p.wslot;          // slots in sample order (a grouped launch folds them per weight set)
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int nt = (KSPLIT == 1) ? wv + 4 * j : j;
            const int tap = 2 * nt + (n >> 3), c = n & 7;
            const long ci = cbase + c0 + c;
            if (KS == 1 && ident && nt == 0 && n >= 8) {          // columns 8..15: the indicator sums of channels c0 .. c0 + 7
                float* sp = p.spos + (long)bx * ((long)p.Cout * p.Cin);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = (mg * MT + mt) * 16 + g * 4 + r;
                        if (co < p.Cout) sp[(long)co * p.Cin + ci] = acc[mt][j][r];
                    }
                continue;
            }
            if (nt >= NTILES || tap >= NTAP) continue;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = (mg * MT + mt) * 16 + g * 4 + r;
                    if (co < p.Cout) slot[((long)co * p.Cin + ci) * NTAP + tap] = acc[mt][j][r];
                }
        }
    }
    if (KS == 1 && ident) {
        if (bias_block && n == 0) {                               // every column of the ones tile holds the bias sums
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = (mg * MT + mt) * 16 + g * 4 + r;
                    if (co < p.Cout) dbias[(long)bx * p.wslot + co] = accb[mt][r];
                }
        }
    } else
    if (bias_block && (KSPLIT == 4 || wv == WB) && n == 8) {       // column 8 of the last tile = bias sums
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = (mg * MT + mt) * 16 + g * 4 + r;
                if (co < p.Cout) dbias[(long)bx * p.wslot + co] = acc[mt][JB][r];
            }
    }
}

}  // namespace pconv_wl
