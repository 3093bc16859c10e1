// Marching forward of the FIRST stride-2 conv of an encoder (e2_c1: 3x3x3, stride 2, 8 -> 16 channels over the 128^3 level;
// reference code/models/mmformer.py:40-45 via general_conv3d_prenorm, blocks.py:300-316): y = Conv3d(LeakyReLU(InstanceNorm(x))) + bias.
//
// The general brick kernel (conv_bf16_kernel<3, 2, 1, 0, 8>) ran this launch at 108 us for 201 MB (4 x 128^3 -> 64^3, the four
// modality encoders as one grouped launch): a 4x8x8 output brick stages a 9x17x17 input brick -- 2.5x its share of the volume as halo --
// and walks 27 taps with a barrier pair per brick.  Same recipe as the stride-1 marching kernels (conv_march.hip), stride-2 geometry:
//   * a workgroup owns an 8 x 16 tile of the OUTPUT (H, W) plane and marches along D: output plane od needs the input planes
//     2 od - 1 .. 2 od + 1, so an iteration stages TWO new planes (2 od, 2 od + 1; halo 17/16 x 33/32 per plane instead of
//     9/8 x 17/16 x 17/16 per brick) and keeps 2 od - 1 from the iteration before.  Eight ring slots: plane i lives in slot i & 7, the
//     slots an iteration overwrites were last read two iterations ago -- ONE barrier per output plane;
//   * a staged row is split by the parity of the input column: the taps of one output voxel read columns 2 ow - 1, 2 ow, 2 ow + 1, so
//     with the odd columns [17 voxels] in front of the even ones [16] the sixteen voxel lanes of a B fragment read sixteen consecutive
//     16-byte slots for every tap (stride-2 addressing would put two lanes on every bank pair);
//   * K = 27 taps x 8 channels = seven v_mfma_f32_16x16x32_bf16 steps (lane group g of step s holds tap 4 s + g; tap 27 is zero
//     weights), the sixteen output channels fill the M tile, weights of the whole K axis in registers, one ds_read_b128 per MFMA;
//   * all index arithmetic (padding, parity split, LDS offsets) once per workgroup; the loads of the next output plane are in flight
//     while the current one is multiplied; normalisation + activation on the way into LDS; bias, bf16 rounding and the InstanceNorm
//     statistics of y (for the next pre-norm conv) in the epilogue, through the same slots / tickets as the other forward kernels.
// Same operand images, weight pack and statistics contract as conv_march.hip; grouped weights (sample b uses set b / bgroup).
#include <cstdlib>
#include "conv_common.h"
#include "prof.h"

namespace pconv { int g_conv_s2f = 1; extern int g_conv_target; }      // option "conv_s2f" / PASSION_CONV_S2F=0: back to the brick kernel (A/B)
using namespace pconv;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int TOH = 8, TOW = 16;                  // output tile (rows: two per wave)
constexpr int IR = 2 * TOH + 1, IC = 2 * TOW + 1; // staged input rows / columns of a plane: 17 x 33
constexpr int NODD = TOW + 1;                     // odd-parity columns of a staged row (2 ow0 - 1 + 2 j, j = 0 .. 16) sit in front of the even ones
constexpr int RS = (IC + 1) * 16;                 // 544 bytes per staged row (one voxel of padding: 32 mod 256, consecutive rows start 8 banks apart)
constexpr int SLOT = IR * RS;                     // 9248
constexpr int NSLOT = 8;
constexpr int RING = NSLOT * SLOT;                // 73984: two workgroups per CU
constexpr int PV = IR * IC;                       // 561 voxels per staged plane
constexpr int NIT = (2 * PV + 255) / 256;         // 5 loads per thread and output plane
constexpr int NSTEP = 7;

struct S2FP {
    int tilesH, tilesW;
    int nseg_d, planes_per_seg;    // D segments per tile and OUTPUT planes per segment
    int nwg_b;                     // workgroups per sample
};

template <typename T> __device__ __forceinline__ T gload(const void* ptr) {      // a GLOBAL load, whatever the pointer's history
    return *reinterpret_cast<const T __attribute__((address_space(1)))*>(reinterpret_cast<unsigned long long>(ptr));
}
__device__ __forceinline__ float uniform(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_s2f_kernel(ConvK p, S2FP mp) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    __shared__ float lsum[4 * 16 * 2 + 32];

    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Sd = p.Di, Sh = p.Hi, Sw = p.Wi, Od = p.Do, Oh = p.Ho, Ow = p.Wo;

    const int bx = xcd_swizzle(blockIdx.x, gridDim.x);
    const int b = bx / mp.nwg_b;
    int r = bx % mp.nwg_b;
    const int tw_i = r % mp.tilesW; r /= mp.tilesW;
    const int th_i = r % mp.tilesH; r /= mp.tilesH;
    const int dseg = r;
    const int oh0 = th_i * TOH, ow0 = tw_i * TOW;
    const int olo = dseg * mp.planes_per_seg, ohi = min(olo + mp.planes_per_seg, Od);
    if (ohi <= olo) return;

    // ---- the input tensor (one 8-channel segment), its modality mask and statistics
    const SegD& sd = p.seg[0];
    const int bs = sd.bsrc(b);
    const bool present = !sd.mask || sd.mask[(long)b * sd.mask_ld + sd.mask_col] != 0;
    const bf16_t* src = reinterpret_cast<const bf16_t*>(sd.x) + (long)bs * sd.bstride;
    const bool has_norm = sd.stats != nullptr;
    f32x2 sc[4], sh[4];                                // x_hat = x * rstd + shift, channel pairs
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
    const bool reflect = p.pad_mode == PASSION_PAD_REFLECT;

    // ---- staging geometry of this thread, once: voxel t + 256 u of the two planes an iteration stages
    unsigned goff[NIT];                                // element offset inside an input plane (clamped)
    int loff[NIT];                                     // byte offset inside a ring slot
    unsigned okst = 0u, okin = 0u, psel = 0u;          // bit u: has a voxel / the voxel is inside the volume (zero padding) / second plane
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
    const unsigned plane_el = (unsigned)(Sh * Sw) * 8u;

    // every load of the plane loop is UNCONDITIONAL (clamped address, data dropped at commit time): a load inside a branch makes the
    // compiler wait with vmcnt(0) and drains the prefetch (conv_march.hip)
    u32x4 raw[NIT];
    auto fetch = [&](int od) {                         // input planes 2 od and 2 od + 1
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            int pl = 2 * od + (int)((psel >> u) & 1u);
            if (reflect) pl = reflect_idx(pl, Sd);
            pl = min(max(pl, 0), Sd - 1);
            raw[u] = gload<u32x4>(src + ((unsigned)pl * plane_el + goff[u]));
        }
    };
    const f32x2 slope2 = {p.slope, p.slope};
    auto commit = [&](int od) {
        if (p.dbg & 1) return;
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int pl = 2 * od + (int)((psel >> u) & 1u);
            const bool ok = present && ((okin >> u) & 1u) && (reflect || (pl >= 0 && pl < Sd));
            u32x4 o = (u32x4){0u, 0u, 0u, 0u};         // zero padding is applied AFTER norm + activation
            if (ok) {
                o = raw[u];
                if (has_norm) {
                    bf16x8 ob;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {      // bf16 pairs -> f32: low half = even channel
                        const f32x2 xv = {__uint_as_float(raw[u][c] << 16), __uint_as_float(raw[u][c] & 0xffff0000u)};
                        f32x2 t = __builtin_elementwise_fma(xv, sc[c], sh[c]);
                        const f32x2 ts = t * slope2;   // LeakyReLU for 0 <= slope <= 1: max(t, slope t)
                        t.x = fmaxf(t.x, ts.x);
                        t.y = fmaxf(t.y, ts.y);
                        ob[2 * c] = (__bf16)t.x; ob[2 * c + 1] = (__bf16)t.y;
                    }
                    o = __builtin_bit_cast(u32x4, ob);
                }
            }
            if ((okst >> u) & 1u) *reinterpret_cast<u32x4*>(lds + (pl & (NSLOT - 1)) * SLOT + loff[u]) = o;
        }
    };
    fetch(olo - 1);                                    // (planes 2 olo - 2, 2 olo - 1: the first is never read)

    // ---- weights: all seven A fragments of the K axis; per-lane tap geometry of the B fragments
    bf16x8 wfr[NSTEP];
    int tkd[NSTEP], toff[NSTEP];
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        const int tap = min(4 * s + g, 26);
        wfr[s] = gload<bf16x8>(reinterpret_cast<const bf16x8*>(p.wpack) + wvec_of(p, b) + (long)tap * p.Mpad + n);
        if (4 * s + g > 26) {
#pragma unroll
            for (int j = 0; j < 8; ++j) wfr[s][j] = (__bf16)0.f;
        }
        const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        tkd[s] = kd;
        toff[s] = kh * RS + (kw == 1 ? NODD : (kw >> 1)) * 16;      // kw 0: odd[n], 1: even[n], 2: odd[n + 1]
    }
    const int lanebase = (4 * wv) * RS + n * 16;       // output row 2 wv + j of the tile: staged rows 2 (2 wv + j) + kh
    f32x4 biasv = {0.f, 0.f, 0.f, 0.f};
    if (bias_of(p, b)) biasv = load4(bias_of(p, b) + g * 4);
    bf16_t* yb = reinterpret_cast<bf16_t*>(p.y) + (long)b * Od * Oh * Ow * 16;
    const unsigned oel = (unsigned)((oh0 + 2 * wv) * Ow + ow0 + n) * 16u + (unsigned)(g * 4);
    const unsigned oplane = (unsigned)(Oh * Ow) * 16u, orow = (unsigned)Ow * 16u;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const bool fstats = p.fst.stats != nullptr;

    // ---- pipeline.  iteration od: commit the planes 2 od, 2 od + 1; request those of od + 1; barrier; multiply output plane od
    // (planes 2 od - 1 .. 2 od + 1) and store it.  The first iteration (od = olo - 1) only stages.
    commit(olo - 1);
    fetch(olo);
    for (int od = olo; od < ohi; ++od) {
        commit(od);
        fetch(min(od + 1, ohi - 1));
        __syncthreads();
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if (!(p.dbg & 2)) {
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                const char* bp = lds + ((2 * od - 1 + tkd[s]) & (NSLOT - 1)) * SLOT + toff[s] + lanebase;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const bf16x8 bv = *reinterpret_cast<const bf16x8*>(bp + j * 2 * RS);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[s], bv, acc[j], 0, 0, 0);
                }
            }
        }
        if (p.dbg & 4) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f32x4 v = acc[j] + biasv;
            unsigned pk[2];
            float rb[4];
            round4_bf16(v, pk, rb);
            *reinterpret_cast<u32x2 __attribute__((address_space(1)))*>(reinterpret_cast<unsigned long long>(yb + ((unsigned)od * oplane + oel + (unsigned)j * orow))) = (u32x2){pk[0], pk[1]};
            if (fstats) {
#pragma unroll
                for (int c = 0; c < 4; ++c) { s1[c] += rb[c]; s2[c] = __builtin_fmaf(rb[c], rb[c], s2[c]); }
            }
        }
    }

    if (fstats) {
        // statistics of y: 16 voxel lanes (shuffles) -> 4 waves (LDS, fixed order) -> this workgroup's slot (conv_march.hip)
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
        if (tid < 32) {
            const int c = tid >> 1, k2 = tid & 1;
            lsum[4 * 16 * 2 + tid] = ((lsum[(0 * 16 + c) * 2 + k2] + lsum[(1 * 16 + c) * 2 + k2]) + lsum[(2 * 16 + c) * 2 + k2]) + lsum[(3 * 16 + c) * 2 + k2];
        }
        __syncthreads();
        fwd_stats_commit(p.fst, p.Cout, b, bx % mp.nwg_b, 0, 16, lsum + 4 * 16 * 2, tid);
    }
}

}  // namespace

namespace pconv {

// 0 = launched, < 0 = HIP error, 1 = not this shape (the caller goes on to the general kernel)
int launch_conv_s2f(ConvK& k, int ksize, int stride, const void* packed_bf16, hipStream_t st) {
    static const bool env_on = []() { const char* e = getenv("PASSION_CONV_S2F"); return !(e && e[0] == '0'); }();
    if (!g_conv_s2f || !env_on || ksize != 3 || stride != 2 || k.nseg != 1 || k.seg[0].C != 8 || k.Cin != 8 || k.Cout != 16 || k.residual || k.y_f32) return 1;
    if (k.seg[0].bmod > 0 || (k.Di | k.Hi | k.Wi) & 1 || k.Ho % TOH || k.Wo % TOW || k.Do < 4) return 1;
    if (k.pad_mode == PASSION_PAD_REFLECT && (k.Di < 4 || k.Hi < 4 || k.Wi < 4)) return 1;
    if ((long)k.Di * k.Hi * k.Wi * 16 >= (1L << 32)) return 1;                 // 32-bit byte offsets inside a sample
    S2FP mp;
    mp.tilesH = k.Ho / TOH; mp.tilesW = k.Wo / TOW;
    const int tiles = mp.tilesH * mp.tilesW;
    // D segments: about g_conv_target workgroups per launch, at least 8 output planes per segment (an extra staging iteration each)
    int nseg = (int)hmax(1, (g_conv_target + (long)k.B * tiles - 1) / ((long)k.B * tiles));
    nseg = (int)hmin(nseg, hmax(1, k.Do / 8));
    mp.planes_per_seg = (k.Do + nseg - 1) / nseg;
    mp.nseg_d = (k.Do + mp.planes_per_seg - 1) / mp.planes_per_seg;
    mp.nwg_b = mp.nseg_d * tiles;
    if ((long)k.B * mp.nwg_b < 256) return 1;        // (4 x 64^3: 128 workgroups, 19 us against 16 on the brick kernel; 4 x 128^3: 44 against 79)
    k.Mpad = 16;
    k.wpack = reinterpret_cast<const bf16_t*>(packed_bf16);
    k.fst.stats = nullptr;
    if (k.fst_want) {
        const int rc = fwd_stats_setup(k, mp.nwg_b, 1, st);
        if (rc) return rc < 0 ? rc : -rc;
    }
    static bool granted = false;
    if (!granted) {
        HIP_RET(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_s2f_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, RING));
        granted = true;
    }
    passion_prof_tag("conv_s2f_kernel");
    hipLaunchKernelGGL(conv_s2f_kernel, dim3(k.B * mp.nwg_b), dim3(256), RING, st, k, mp);
    LAUNCH_RET();
    return 0;
}

}  // namespace pconv
