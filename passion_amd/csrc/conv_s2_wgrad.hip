// Marching weight gradient of the FIRST stride-2 conv of an encoder (e2_c1: 3x3x3, stride 2, 8 -> 16 channels; reference
// code/models/mmformer.py:40-45 via general_conv3d_prenorm, blocks.py:300-316):
//   dW[co][ci][tap] = sum_v dY[v][co] * act(x_hat)[2 v + tap - 1][ci],     dbias[co] = sum_v dY[v][co]        (v: output voxels)
// The general brick kernel (conv_wgrad_bf16_kernel<3, 2, 1, 8>) ran this launch at 96 us (4 x 128^3 -> 64^3; 241 MB of counter
// traffic for 201 MB of operands).  The schedule of wgrad_march16_kernel (conv_wgrad_march.hip: sixteen dY rows in the M tile, the
// K axis = output voxels through transposing LDS reads, 27 taps x 8 channels = 14 column tiles of two taps) with the stride-2 staging
// of conv_s2_fwd.hip: a workgroup owns 8 x 16 tiles of the OUTPUT (H, W) plane and marches along D; an iteration stages the input
// planes 2 od, 2 od + 1 (2 od - 1 is there from the iteration before; eight ring slots: one barrier per output plane), rows split by
// column parity so that the sixteen output columns of a tap are sixteen consecutive 16-byte voxels; and the dY plane od.
// Tap 27 does not exist: the second half of tile 13 multiplies dY by ONES instead and yields the bias gradient.
// Same operand images, slab layout (one slot per workgroup, sample order: a grouped launch folds per weight set) and fixed-order
// fold as the other weight-gradient kernels.
#include <cstdlib>
#include "conv_common.h"
#include "prof.h"

namespace pconv { int g_wgrad_s2m = 1; }      // option "wgrad_s2m" / PASSION_WGRAD_S2M=0: back to the brick kernel (A/B)
using namespace pconv;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int TOH = 8, TOW = 16;                  // output tile (rows: two per wave = one K step of 32 voxels)
constexpr int IR = 2 * TOH + 1, IC = 2 * TOW + 1; // staged input rows / columns of a plane: 17 x 33
constexpr int NODD = TOW + 1;                     // odd-parity columns (2 ow0 - 1 + 2 j) in front of the even ones
constexpr int RS = IC * 16;                       // 528 bytes per staged row (no padding voxel: ring + dY buffers = 80128 bytes, two workgroups per CU)
constexpr int XSLOT = IR * RS;                    // 8976
constexpr int NXS = 8;                            // ring slots of X planes: plane i in slot i & 7
constexpr int YSLOT = TOH * TOW * 32 + 64;        // one dY plane of the tile (16 channels), double-buffered
constexpr int XB = NXS * XSLOT, YB = 2 * YSLOT;
constexpr int PV = IR * IC;                       // 561
constexpr int NIT = (2 * PV + 255) / 256;         // 5 X loads per thread and output plane
constexpr int NT = 14;                            // column tiles: taps 2 j, 2 j + 1
constexpr int RED_BYTES = 4 * (NT / 2) * 64 * 16; // the waves' partial sums meet in two halves of seven tiles
constexpr int LDS_BYTES = XB + YB > RED_BYTES ? XB + YB : RED_BYTES;

struct S2WP {
    int wgs_per_b;                 // workgroups (= slots of the partial slab) per sample
    int ncols, tilesW;             // (H, W) tiles per sample
    int nseg_d, planes_per_seg;    // D segments per tile; a unit = (tile, segment), dealt round-robin to the sample's workgroups
};

template <typename T> __device__ __forceinline__ T gload(const void* ptr) {
    return *reinterpret_cast<const T __attribute__((address_space(1)))*>(reinterpret_cast<unsigned long long>(ptr));
}
__device__ __forceinline__ float uniform(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ bf16x8 tr_read8(const char* p0, const char* p1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p1));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad_s2m_kernel(ConvK p, S2WP mp, float* __restrict__ dW, float* __restrict__ dbias) {
    wgrad_clear(p);
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const lx = lds;
    char* const ly = lds + XB;

    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = n >> 2, pq = n & 3;
    const int Sd = p.Di, Sh = p.Hi, Sw = p.Wi, Od = p.Do, Oh = p.Ho, Ow = p.Wo;

    const int bx = xcd_swizzle(blockIdx.x, gridDim.x);
    const int b = bx / mp.wgs_per_b;                  // >= B: spare workgroup, flushes a zero slot
    const int i0 = bx % mp.wgs_per_b;
    const int nunits = (b < p.B) ? mp.ncols * mp.nseg_d : 0;

    // ---- the input tensor (one 8-channel segment), its modality mask and statistics
    const SegD& sd = p.seg[0];
    const int bb = b < p.B ? b : 0, bs = sd.bsrc(bb);
    const bool present = !sd.mask || sd.mask[(long)bb * sd.mask_ld + sd.mask_col] != 0;
    const bf16_t* src = reinterpret_cast<const bf16_t*>(sd.x) + (long)bs * sd.bstride;
    const bool has_norm = sd.stats != nullptr;
    f32x2 sc[4], sh[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { sc[c] = (f32x2){1.f, 1.f}; sh[c] = (f32x2){0.f, 0.f}; }
    if (has_norm) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 st4 = *reinterpret_cast<const f32x4*>(sd.stats + ((long)bs * 8 + 2 * c) * 2);      // mean, rstd, mean, rstd
            sc[c] = (f32x2){uniform(st4[1]), uniform(st4[3])};
            sh[c] = (f32x2){uniform(-st4[0] * st4[1]), uniform(-st4[2] * st4[3])};
        }
    }
    const bf16_t* dyb = reinterpret_cast<const bf16_t*>(p.dy) + (long)bb * Od * Oh * Ow * 16;      // Cout == 16
    const bool reflect = p.pad_mode == PASSION_PAD_REFLECT;

    // ---- per-lane constants of the MFMA loop (wgrad_march16_kernel): wave wv owns output rows 2 wv, 2 wv + 1 of the tile; K index g * 8 + e
    // of its step is the voxel (row 2 wv + (g >> 1), column (g & 1) * 8 + e); lane (q, pq) LOADS 8 bytes of voxel q (+ 4 in the second
    // read): A = channels 4 pq .. of dY, B = channels (pq & 1) * 4 .. of tap 2 j + (pq >> 1) of tile j
    const int row = 2 * wv + (g >> 1), col = (g & 1) * 8 + q;
    const int yoff = (row * TOW + col) * 32 + pq * 8;
    const int h = pq >> 1;
    int toff[NT], tkd[NT];                            // in-slot byte offset of the lane's tap of tile j, and its depth tap
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int t = min(2 * j + h, 26), kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
        tkd[j] = kd;
        toff[j] = (2 * row + kh) * RS + (col + (kw == 1 ? NODD : (kw >> 1))) * 16 + (pq & 1) * 8;
    }
    const bool ones13 = (n >> 3) == 1;                // tile 13, columns 8 .. 15 of the RESULT (the transposing read hands lane n column n): tap 27
                                                      // does not exist -> ones (the bias gradient)
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.f;
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x2 slope2 = {p.slope, p.slope};

    // ---- staging roles: every thread NIT voxels of the two X planes of an iteration, and one 16-byte half of a dY voxel
    const int yv = tid >> 1, yhalf = tid & 1, vh = yv >> 4, vw = yv & 15;
    const int ylds = yv * 32 + yhalf * 16;
    const unsigned xplane_el = (unsigned)(Sh * Sw) * 8u, yplane_el = (unsigned)(Oh * Ow) * 16u;

    for (int unit = i0; unit < nunits; unit += mp.wgs_per_b) {
        const int colidx = unit % mp.ncols, dseg = unit / mp.ncols;
        const int oh0 = (colidx / mp.tilesW) * TOH, ow0 = (colidx % mp.tilesW) * TOW;
        const int olo = dseg * mp.planes_per_seg, ohi = min(olo + mp.planes_per_seg, Od);
        if (ohi <= olo) continue;
        unsigned goff[NIT];
        int loff[NIT];
        unsigned okst = 0u, okin = 0u, psel = 0u;      // bit u: has a voxel / inside the volume (zero padding) / second plane
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            int idx = tid + 256 * u;
            if (idx < 2 * PV) okst |= 1u << u;
            idx = min(idx, 2 * PV - 1);
            const int ps = idx >= PV ? 1 : 0, w = idx - PV * ps;
            const int zr = w / IC, zc = w - zr * IC;
            int ih = 2 * oh0 - 1 + zr, iw = 2 * ow0 - 1 + zc;
            bool in = true;
            if (reflect) { ih = reflect_idx(ih, Sh); iw = reflect_idx(iw, Sw); }
            else in = ih >= 0 && ih < Sh && iw >= 0 && iw < Sw;
            ih = min(max(ih, 0), Sh - 1); iw = min(max(iw, 0), Sw - 1);
            goff[u] = (unsigned)(ih * Sw + iw) * 8u;
            loff[u] = zr * RS + ((zc & 1) ? (NODD + (zc >> 1)) : (zc >> 1)) * 16;
            if (in) okin |= 1u << u;
            if (ps) psel |= 1u << u;
        }
        const unsigned yel = (unsigned)((oh0 + vh) * Ow + ow0 + vw) * 16u + (unsigned)yhalf * 8u;

        // every load of the plane loop is UNCONDITIONAL (clamped address, data dropped at commit time) -- conv_march.hip
        u32x4 ra[NIT], rd;
        auto fetch = [&](int od) {                     // X planes 2 od, 2 od + 1 and the dY plane od
#pragma unroll
            for (int u = 0; u < NIT; ++u) {
                int pl = 2 * od + (int)((psel >> u) & 1u);
                if (reflect) pl = reflect_idx(pl, Sd);
                pl = min(max(pl, 0), Sd - 1);
                ra[u] = gload<u32x4>(src + ((unsigned)pl * xplane_el + goff[u]));
            }
            rd = gload<u32x4>(dyb + ((unsigned)min(max(od, olo), ohi - 1) * yplane_el + yel));
        };
        auto commit = [&](int od) {
#pragma unroll
            for (int u = 0; u < NIT; ++u) {
                const int pl = 2 * od + (int)((psel >> u) & 1u);
                const bool ok = present && ((okin >> u) & 1u) && (reflect || (pl >= 0 && pl < Sd));
                u32x4 o = (u32x4){0u, 0u, 0u, 0u};     // zero padding is applied AFTER norm + activation
                if (ok) {
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
                if ((okst >> u) & 1u) *reinterpret_cast<u32x4*>(lx + (pl & (NXS - 1)) * XSLOT + loff[u]) = o;
            }
            if (od >= olo) *reinterpret_cast<u32x4*>(ly + (od & 1) * YSLOT + ylds) = rd;
        };

        // ---- pipeline.  iteration od: commit X planes 2 od, 2 od + 1 and dY plane od; request those of od + 1; barrier; multiply.
        // The slots an iteration overwrites (X: planes 2 od - 8, 2 od - 7; dY: plane od - 2) were last read two iterations ago.
        __syncthreads();                               // the previous unit's last readers are done with the rings
        fetch(olo - 1);
        commit(olo - 1);                               // (X planes 2 olo - 2, 2 olo - 1: the first is never read; no dY plane)
        fetch(olo);
        for (int od = olo; od < ohi; ++od) {
            commit(od);
            fetch(min(od + 1, ohi - 1));
            __syncthreads();
            const char* ya = ly + (od & 1) * YSLOT + yoff;
            const bf16x8 a = tr_read8(ya, ya + 128);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const char* bp = lx + ((2 * od - 1 + tkd[j]) & (NXS - 1)) * XSLOT + toff[j];
                bf16x8 bv = tr_read8(bp, bp + 64);
                if (j == NT - 1 && ones13) bv = ones;
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bv, acc[j], 0, 0, 0);
            }
        }
    }

    // ---- the four waves hold partial sums over disjoint voxels of the same outputs: combine through LDS in fixed order (two halves of
    // seven tiles), write this workgroup's slot of the partial-sum slab
    f32x4* red = reinterpret_cast<f32x4*>(lds);
    constexpr int HT = NT / 2;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < HT; ++j) red[(wv * HT + j) * 64 + lane] = acc[half * HT + j];
        __syncthreads();
        float* slot = dW ? dW + (long)bx * p.wslot : nullptr;
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
            if (tap > 26) {                            // the ones column: sum_v dY[v][co], eight copies
                if (dbias && ci == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) dbias[(long)bx * p.wslot + g2 * 4 + r] = v[r];
                }
                continue;
            }
            if (slot) {
#pragma unroll
                for (int r = 0; r < 4; ++r) slot[((long)(g2 * 4 + r) * 8 + ci) * 27 + tap] = v[r];
            }
        }
    }
}

}  // namespace

namespace pconv {

// the launches this kernel takes: the slab has gx slots (a whole number per sample)
bool wgrad_s2m_covers(const ConvK& k, int ksize, int stride, int gx) {
    static const bool env_on = []() { const char* e = getenv("PASSION_WGRAD_S2M"); return !(e && e[0] == '0'); }();
    if (!g_wgrad_s2m || !env_on || ksize != 3 || stride != 2 || k.nseg != 1 || k.seg[0].C != 8 || k.Cin != 8 || k.Cout != 16 || k.y_f32) return false;
    if (k.seg[0].bmod > 0 || ((k.Di | k.Hi | k.Wi) & 1) || k.Ho % TOH || k.Wo % TOW || k.Do < 4) return false;
    if (k.pad_mode == PASSION_PAD_REFLECT && (k.Di < 4 || k.Hi < 4 || k.Wi < 4)) return false;
    if (gx % k.B || gx / k.B < 1) return false;
    if ((long)k.B * k.Do * k.Ho * k.Wo < (1L << 18)) return false;              // small launches stay on the brick kernel
    return (long)k.Di * k.Hi * k.Wi * 16 < (1L << 32) && (long)k.Do * k.Ho * k.Wo * 32 < (1L << 32);
}

int launch_wgrad_s2m(ConvK& k, int ksize, int stride, int gx, float* dW, float* db, hipStream_t st) {
    if (!wgrad_s2m_covers(k, ksize, stride, gx)) return 1;
    S2WP mp;
    mp.wgs_per_b = gx / k.B;
    mp.tilesW = k.Wo / TOW;
    mp.ncols = (k.Ho / TOH) * mp.tilesW;
    // D segments: the split that keeps most of the sample's workgroups busy in every round; ties go to the longer march
    int best = 1;
    double best_eff = -1.0;
    for (int ns = 1; ns <= 64; ns *= 2) {
        const int pps = (k.Do + ns - 1) / ns;
        if (ns > 1 && pps < 4) break;
        const long units = (long)mp.ncols * ((k.Do + pps - 1) / pps);
        const long rounds = (units + mp.wgs_per_b - 1) / mp.wgs_per_b;
        const double eff = (double)units / (double)(rounds * mp.wgs_per_b);
        if (eff > best_eff + 0.02) { best_eff = eff; best = ns; }
    }
    mp.planes_per_seg = (k.Do + best - 1) / best;
    mp.nseg_d = (k.Do + mp.planes_per_seg - 1) / mp.planes_per_seg;
    static bool granted = false;
    if (!granted) {
        HIP_RET(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_s2m_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        granted = true;
    }
    passion_prof_tag("wgrad_s2m_kernel");
    hipLaunchKernelGGL(wgrad_s2m_kernel, dim3(gx), dim3(256), LDS_BYTES, st, k, mp, dW, db);
    LAUNCH_RET();
    return 0;
}

}  // namespace pconv
