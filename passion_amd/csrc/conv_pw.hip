// Pointwise (1x1x1, stride 1) bf16 forward conv of general_conv3d_prenorm / general_conv3d (reference code/models/blocks.py:300-316,
// 354-370) for the wide volumes: Cin <= 128 over up to PASSION_MAX_SEG concatenated (masked, normalised) segments, Cout <= 32.
// A 1x1x1 conv over a channels-last tensor is a streaming GEMM whose MFMA B operand IS the memory layout: for
// v_mfma_f32_16x16x32_bf16, lane (n, g) wants 8 consecutive K values of column n -- the 16 bytes of channels 8 g .. 8 g + 7 of
// voxel n.  So there is no LDS stage, no halo, no barrier: every lane loads its own operand straight from HBM (InstanceNorm +
// LeakyReLU applied in registers on the way), the weights of the whole K axis live in registers as A fragments, four 16-voxel
// tiles per wavefront are in flight per trip, and the epilogue stores 8 bytes per lane (16 voxels x 4 rows).  The lean brick
// kernel (conv_lean.hip, KS = 1) staged bricks through LDS with two barriers per brick for these layers and reached 2.6-2.9 TB/s;
// they are 7 % of a mmformer step and 19 % of an RFNet step.
// Same descriptors, masks (absent modalities read as zeros, several masked passes over the same features: SegD.bmod), grouped
// weight sets and fused output statistics (FwdStats) as the other forward kernels.
#include <cstdlib>
#include "conv_common.h"
#include "prof.h"

namespace pconv { extern int g_conv_target; int g_conv_pw = 1, g_conv_pw8 = 1, g_conv_pwfinal = 1; }      // option "conv_pw" / PASSION_CONV_PW=0: back to the lean kernel (A/B)
using namespace pconv;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <typename T> __device__ __forceinline__ T gload(const void* ptr) {
    return *reinterpret_cast<const T __attribute__((address_space(1)))*>(reinterpret_cast<unsigned long long>(ptr));
}
__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (__bf16)0.f;
    return z;
}


// grid (nwg_b, B): workgroup x of sample b walks the 256-voxel chunks x, x + nwg_b, ...
#ifndef PW_TT1
#define PW_TT1 4
#endif
#define PW_TT(NS) ((NS) == 1 ? PW_TT1 : 4)
// TT 16-voxel tiles per wavefront and trip: 16 loads of 16 bytes in flight per lane whatever the K depth (with 8 input channels
// only a quarter of the lanes load data: bytes in flight, not the memory pipes, bound the first version at 3 TB/s)
template <int NSTEP, int MT>
__global__ __launch_bounds__(256) void conv_pw_fwd_kernel(ConvK p, int K8, long V) {
    constexpr int TT = PW_TT(NSTEP);
    __shared__ float lsum[4 * MT * 16 * 2 + MT * 16 * 2];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4, wv = tid >> 6;
    const int b = blockIdx.y;

    // ---- this lane's K blocks: q = 4 s + g -> segment, source pointer, mask, statistics
    const bf16_t* src[NSTEP];
    unsigned cst[NSTEP];
    bool live[NSTEP], hn[NSTEP];
    f32x2 sc[NSTEP][4], sh[NSTEP][4];
    bf16x8 wfr[NSTEP][MT];
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        const int q = 4 * s + g;
        src[s] = reinterpret_cast<const bf16_t*>(p.seg[0].x);
        cst[s] = 0u; live[s] = false; hn[s] = false;
        const uint8_t* mptr = nullptr;
        const float* sptr = nullptr;
        int rr = q;
        bool found = false;
#pragma unroll
        for (int s2 = 0; s2 < PASSION_MAX_SEG; ++s2) {
            if (s2 < p.nseg && !found) {
                const SegD& sd = p.seg[s2];
                const int nb = sd.C / 8;
                if (rr < nb) {
                    found = true;
                    const int bs = sd.bsrc(b), c0 = rr * 8;
                    src[s] = reinterpret_cast<const bf16_t*>(sd.x) + (long)bs * sd.bstride + c0;
                    cst[s] = (unsigned)sd.C;
                    if (sd.mask) mptr = sd.mask + (long)b * sd.mask_ld + sd.mask_col;
                    if (sd.stats) sptr = sd.stats + ((long)bs * sd.C + c0) * 2;
                }
                rr -= nb;
            }
        }
        live[s] = found && q < K8;
        if (mptr && gload<uint8_t>(mptr) == 0) live[s] = false;
        hn[s] = sptr != nullptr;
#pragma unroll
        for (int c = 0; c < 4; ++c) { sc[s][c] = (f32x2){1.f, 1.f}; sh[s][c] = (f32x2){0.f, 0.f}; }
        if (hn[s]) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x4 st4 = gload<f32x4>(sptr + 4 * c);          // (mean, rstd) of channels 2c, 2c + 1
                sc[s][c] = (f32x2){st4[1], st4[3]};
                sh[s][c] = (f32x2){-st4[0] * st4[1], -st4[2] * st4[3]};
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            wfr[s][mt] = zero8();
            if (q < K8) wfr[s][mt] = gload<bf16x8>(reinterpret_cast<const bf16x8*>(p.wpack) + wvec_of(p, b) + ((long)q * p.Mpad + mt * 16 + n));
        }
    }
    f32x4 biasv[MT];
    bool row_ok[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        row_ok[mt] = mt * 16 + g * 4 < p.Cout;
        biasv[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (bias_of(p, b) && row_ok[mt]) biasv[mt] = load4(bias_of(p, b) + mt * 16 + g * 4);
    }
    const f32x2 slope2 = {p.slope, p.slope};
    const bool fstats = p.fst.stats != nullptr;
    float s1[MT][4], s2v[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) { s1[mt][j] = 0.f; s2v[mt][j] = 0.f; }
    char* yb = reinterpret_cast<char*>(p.y) + (long)b * V * p.Cout * (p.y_f32 ? 4 : 2);

    for (long v0 = ((long)blockIdx.x * 4 + wv) * (16 * TT); v0 < V; v0 += (long)gridDim.x * 4 * (16 * TT)) {
        // every load of the trip before the first use
        u32x4 raw[NSTEP][TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const long v = v0 + t * 16 + n;
            const long vc = v < V ? v : V - 1;
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) raw[s][t] = gload<u32x4>(src[s] + vc * cst[s]);
        }
        f32x4 acc[MT][TT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < TT; ++t) acc[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                u32x4 o = raw[s][t];
                if (hn[s]) {
                    bf16x8 ob;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const f32x2 xv = {__uint_as_float(o[c] << 16), __uint_as_float(o[c] & 0xffff0000u)};
                        f32x2 tv = __builtin_elementwise_fma(xv, sc[s][c], sh[s][c]);
                        const f32x2 ts = tv * slope2;
                        tv.x = fmaxf(tv.x, ts.x);
                        tv.y = fmaxf(tv.y, ts.y);
                        ob[2 * c] = (__bf16)tv.x; ob[2 * c + 1] = (__bf16)tv.y;
                    }
                    o = __builtin_bit_cast(u32x4, ob);
                }
                if (!live[s]) o = (u32x4){0u, 0u, 0u, 0u};
                const bf16x8 bfr = __builtin_bit_cast(bf16x8, o);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[s][mt], bfr, acc[mt][t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const long v = v0 + t * 16 + n;
            const bool valid = v < V;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const f32x4 val = acc[mt][t] + biasv[mt];
                const long el = v * p.Cout + mt * 16 + g * 4;
                if (p.y_f32) {
                    if (valid && row_ok[mt]) *reinterpret_cast<f32x4*>(yb + el * 4) = val;
                } else {
                    unsigned pk[2];
                    float rb[4];
                    round4_bf16(val, pk, rb);
                    if (valid && row_ok[mt]) *reinterpret_cast<u32x2*>(yb + el * 2) = (u32x2){pk[0], pk[1]};
                    if (fstats) {
                        const float vf = valid ? 1.f : 0.f;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float r = rb[j] * vf;
                            s1[mt][j] += r;
                            s2v[mt][j] = __builtin_fmaf(r, r, s2v[mt][j]);
                        }
                    }
                }
            }
        }
    }

    if (fstats) {
        // 16 voxel lanes (shuffles) -> 4 waves (LDS, fixed order) -> this workgroup's slot
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
        float* wsum = lsum + 4 * MT * 16 * 2;
        if (tid < MT * 16 * 2) {
            const int c = tid >> 1, k2 = tid & 1;               // c = mt * 16 + row
            float t = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) t += lsum[((w2 * MT * 16) + c) * 2 + k2];
            wsum[tid] = t;
        }
        __syncthreads();
        fwd_stats_commit(p.fst, p.Cout, b, blockIdx.x, 0, MT * 16, wsum, tid);
    }
}

// ---- input gradient of the same convs (conv_lean.hip MODE 1 semantics): g'[ci][v] = sum_co W[co][ci] dY[co][v], back through
// LeakyReLU / the modality mask in the epilogue, the InstanceNorm-backward sums (sum g', sum g' x_hat per (sample, channel)) as one
// fixed-point atomic per workgroup and value (order-independent: bit-reproducible), g' stored into the segments' dx tensors; the
// finish kernel (conv.hip) turns it into dx.  K axis = the Cout channels of dY (no normalisation), rows = Cin in 16-row tiles.
struct alignas(16) PwEpi {       // the four output rows (= forward input channels) of one lane group of one row tile
    const bf16_t* x;
    bf16_t* dx;
    int C, ok, has_stats;
    float mval;
    float mean[4], rstd[4];
};

// FINAL: the InstanceNorm-backward means (p.pwm: m1, m2 per (sample, channel), from the weight gradient -- pw_ident_kernel below)
// are known before the kernel starts: the epilogue writes dx = rstd (g' - m1 - x_hat m2) itself and no finish pass follows.
template <int NSTEP, int MT, bool FINAL>
__global__ __launch_bounds__(256) void conv_pw_dgrad_kernel(ConvK p, int K8, long V) {
    constexpr int TT = 4;
    __shared__ float lsum[4 * MT * 16 * 2];
    __shared__ PwEpi edesc[MT * 4];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4, wv = tid >> 6;
    const int b = blockIdx.y;
    if (tid < MT * 4) {
        const int cg = (tid / 4) * 16 + (tid % 4) * 4;
        PwEpi d;
        d.x = reinterpret_cast<const bf16_t*>(p.seg[0].x); d.dx = nullptr; d.C = 0; d.ok = 0; d.has_stats = 0; d.mval = 1.f;
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
                    d.dx = reinterpret_cast<bf16_t*>(sd.dx) + (long)b * sd.bstride + cl;
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
        edesc[tid] = d;
    }
    // A fragments: the transposed weight images (rows = Cin padded to 16, K = Cout in blocks of 8)
    bf16x8 wfr[NSTEP][MT];
    const bf16_t* src = reinterpret_cast<const bf16_t*>(p.dy) + (long)b * V * p.Cout;
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        const int q = 4 * s + g;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            wfr[s][mt] = zero8();
            const int row = mt * 16 + n;
            if (p.bd_rows) {        // batched passes summed first (conv.hip): K block q = (segment q / bd_k8, block q % bd_k8 of the real Cout); a row sees its own segment only
                if (q < K8 && row / p.bd_rows == q / p.bd_k8)
                    wfr[s][mt] = gload<bf16x8>(reinterpret_cast<const bf16x8*>(p.wpack) + wvec_of(p, b) + ((long)(q % p.bd_k8) * p.Mpad + row));
            } else if (q < K8) {
                wfr[s][mt] = gload<bf16x8>(reinterpret_cast<const bf16x8*>(p.wpack) + wvec_of(p, b) + ((long)q * p.Mpad + row));
            }
        }
    }
    __syncthreads();
    // per-lane copies of what the epilogue needs every trip
    const bf16_t* ex[MT];
    unsigned eC[MT], xS[MT];
    bool e_ok[MT], e_hs[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const PwEpi& e = edesc[mt * 4 + g];
        ex[mt] = e.x; eC[mt] = (unsigned)e.C; e_ok[mt] = e.ok != 0; e_hs[mt] = e.has_stats != 0;
        xS[mt] = e_hs[mt] ? eC[mt] : 0u;               // inputs that were not normalised: x is not needed (LeakyReLU' = 1) -- every lane re-reads one cached address
    }
    float s1[MT][4], s2v[MT][4];                               // FINAL: (m1, m2) of the lane's rows instead of running sums
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s1[mt][j] = 0.f; s2v[mt][j] = 0.f;
            const int row = mt * 16 + g * 4 + j;
            if (FINAL && row < p.Cin) { s1[mt][j] = p.pwm[((long)b * p.Cin + row) * 2]; s2v[mt][j] = p.pwm[((long)b * p.Cin + row) * 2 + 1]; }
        }
    const unsigned cdy = (unsigned)p.Cout;

    for (long v0 = ((long)blockIdx.x * 4 + wv) * (16 * TT); v0 < V; v0 += (long)gridDim.x * 4 * (16 * TT)) {
        u32x4 raw[NSTEP][TT];
        u32x2 xpre[MT][TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const long v = v0 + t * 16 + n;
            const long vc = v < V ? v : V - 1;
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                const int q = 4 * s + g;
                raw[s][t] = gload<u32x4>(src + vc * cdy + (q < K8 ? q * 8 : 0));
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xpre[mt][t] = gload<u32x2>(ex[mt] + vc * xS[mt]);
        }
        f32x4 acc[MT][TT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < TT; ++t) acc[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NSTEP; ++s)
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                const bf16x8 bfr = __builtin_bit_cast(bf16x8, raw[s][t]);      // blocks past K8 meet zero weights
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[s][mt], bfr, acc[mt][t], 0, 0, 0);
            }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const PwEpi& e = edesc[mt * 4 + g];
            f32x4 mean, rstd;
#pragma unroll
            for (int j = 0; j < 4; ++j) { mean[j] = e.mean[j]; rstd[j] = e.rstd[j]; }
            const float f_pos = e.mval, f_neg = e_hs[mt] ? e.mval * p.slope : e.mval;       // LeakyReLU' x modality mask
            bf16_t* dxp = e.dx;
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                const long v = v0 + t * 16 + n;
                const bool valid = e_ok[mt] && v < V;
                const u32x2 xr = xpre[mt][t];
                const float xv[4] = {__uint_as_float(xr.x << 16), __uint_as_float(xr.x & 0xffff0000u),
                                     __uint_as_float(xr.y << 16), __uint_as_float(xr.y & 0xffff0000u)};
                const float vf = (valid && e_hs[mt]) ? 1.f : 0.f;
                f32x4 gv = acc[mt][t];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xh = (xv[j] - mean[j]) * rstd[j];
                    const float gg = gv[j] * (xh > 0.f ? f_pos : f_neg);
                    if (FINAL) {
                        gv[j] = e_hs[mt] ? rstd[j] * (gg - s1[mt][j] - xh * s2v[mt][j]) : gg;
                    } else {
                        gv[j] = gg;
                        const float gs = gg * vf;
                        s1[mt][j] += gs;
                        s2v[mt][j] = __builtin_fmaf(gs, xh, s2v[mt][j]);
                    }
                }
                if (valid) store4(dxp + v * eC[mt], gv);
            }
        }
    }
    if (FINAL) return;
    // one reduction per workgroup: 16 voxel lanes (shuffles) -> 4 waves (LDS, fixed order) -> one replica of the global sums
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const bool st = e_ok[mt] && e_hs[mt];
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
        const float t = ((lsum[(0 * MT * 16 + row) * 2 + k2] + lsum[(1 * MT * 16 + row) * 2 + k2]) + lsum[(2 * MT * 16 + row) * 2 + k2]) +
                        lsum[(3 * MT * 16 + row) * 2 + k2];
        if (row < p.Cin && t != 0.f) {
            const long idx = (long)(blockIdx.x % p.sums_rep) * p.B * p.Cin * 2 + ((long)b * p.Cin + row) * 2 + k2;
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

template <int NSTEP, int MT>
int launch_pw_dgrad(ConvK& k, int K8, hipStream_t st) {
    const long V = (long)k.Di * k.Hi * k.Wi;
    const long chunks = (V + 255) / 256;
    const long nwg_b = hmax(1, hmin(chunks / 2 > 0 ? chunks / 2 : 1, (g_conv_target + k.B - 1) / k.B));
    passion_prof_tag("conv_pw_dgrad_kernel<%d, %d, %d>", NSTEP, MT, k.pwm ? 1 : 0);
    if (k.pwm) hipLaunchKernelGGL((conv_pw_dgrad_kernel<NSTEP, MT, true>), dim3((unsigned)nwg_b, k.B), dim3(256), 0, st, k, K8, V);
    else hipLaunchKernelGGL((conv_pw_dgrad_kernel<NSTEP, MT, false>), dim3((unsigned)nwg_b, k.B), dim3(256), 0, st, k, K8, V);
    LAUNCH_RET();
    k.pw_done = 1;
    return 0;
}

// ---- 8-channel K axis (Cin = 8 forward; Cout <= 8 gradient: the level-1 d1_out convs and the logit heads).  With K = 8 the
// 16x16x32 form above feeds data through a quarter of its lanes (3.0 TB/s).  v_mfma_f32_32x32x8_bf16 has K = 8 exactly: lane
// (col, h) holds channels 4h .. 4h + 3 of voxel col -- EVERY lane loads 8 useful bytes, a wavefront instruction covers 32 voxels
// = 512 contiguous bytes, and the result rows 8i + 4h + j give lane (col, h) four consecutive output channels of its voxel.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float group32_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// forward, one 8-channel segment, Cout <= 8 (bf16 or f32 out).  grid (nwg_b, B)
__global__ __launch_bounds__(256) void conv_pw8_fwd_kernel(ConvK p, long V) {
    constexpr int TT = 8;
    __shared__ float lsum[4 * 8 * 2 + 16];
    const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, h = lane >> 5, wv = tid >> 6;
    const int b = blockIdx.y;
    const SegD& sd = p.seg[0];
    const int bs = sd.bsrc(b);
    const bf16_t* src = reinterpret_cast<const bf16_t*>(sd.x) + (long)bs * sd.bstride + 4 * h;
    bool live = true;
    if (sd.mask) live = gload<uint8_t>(sd.mask + (long)b * sd.mask_ld + sd.mask_col) != 0;
    const bool hn = sd.stats != nullptr;
    f32x2 sc[2] = {{1.f, 1.f}, {1.f, 1.f}}, sh[2] = {{0.f, 0.f}, {0.f, 0.f}};
    if (hn) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const f32x4 st4 = gload<f32x4>(sd.stats + ((long)bs * 8 + 4 * h + 2 * c) * 2);
            sc[c] = (f32x2){st4[1], st4[3]};
            sh[c] = (f32x2){-st4[0] * st4[1], -st4[2] * st4[3]};
        }
    }
    // A fragment: W[row = col][4h .. 4h + 3] from the packed image [blk 0][Mpad rows] of 8 bf16
    s16x4 wf = {0, 0, 0, 0};
    if (col < p.Mpad) {
        const u32x2 w2 = gload<u32x2>(reinterpret_cast<const char*>(p.wpack) + (wvec_of(p, b) + col) * 16 + 8 * h);
        wf = __builtin_bit_cast(s16x4, w2);
    }
    const bool row_ok = 4 * h < p.Cout;
    f32x4 biasv = {0.f, 0.f, 0.f, 0.f};
    if (bias_of(p, b) && row_ok) biasv = load4(bias_of(p, b) + 4 * h);
    const f32x2 slope2 = {p.slope, p.slope};
    const bool fstats = p.fst.stats != nullptr;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2v[4] = {0.f, 0.f, 0.f, 0.f};
    char* yb = reinterpret_cast<char*>(p.y) + (long)b * V * p.Cout * (p.y_f32 ? 4 : 2);

    for (long v0 = ((long)blockIdx.x * 4 + wv) * (32 * TT); v0 < V; v0 += (long)gridDim.x * 4 * (32 * TT)) {
        u32x2 raw[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const long v = v0 + t * 32 + col;
            raw[t] = gload<u32x2>(src + (v < V ? v : V - 1) * 8);
        }
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            u32x2 o = raw[t];
            if (hn) {
                unsigned pk[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const f32x2 xv = {__uint_as_float(o[c] << 16), __uint_as_float(o[c] & 0xffff0000u)};
                    f32x2 tv = __builtin_elementwise_fma(xv, sc[c], sh[c]);
                    const f32x2 ts = tv * slope2;
                    tv.x = fmaxf(tv.x, ts.x);
                    tv.y = fmaxf(tv.y, ts.y);
                    const __bf16 lo = (__bf16)tv.x, hi = (__bf16)tv.y;
                    pk[c] = (unsigned)__builtin_bit_cast(unsigned short, lo) | ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
                }
                o = (u32x2){pk[0], pk[1]};
            }
            if (!live) o = (u32x2){0u, 0u};
            f32x16 acc;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(wf, __builtin_bit_cast(s16x4, o), acc, 0, 0, 0);
            const long v = v0 + t * 32 + col;
            const bool valid = v < V;
            const f32x4 val = (f32x4){acc[0], acc[1], acc[2], acc[3]} + biasv;          // rows 4h .. 4h + 3
            const long el = v * p.Cout + 4 * h;
            if (p.y_f32) {
                if (valid && row_ok) *reinterpret_cast<f32x4*>(yb + el * 4) = val;
            } else {
                unsigned pk[2];
                float rb[4];
                round4_bf16(val, pk, rb);
                if (valid && row_ok) *reinterpret_cast<u32x2*>(yb + el * 2) = (u32x2){pk[0], pk[1]};
                if (fstats) {
                    const float vf = valid ? 1.f : 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float r = rb[j] * vf;
                        s1[j] += r;
                        s2v[j] = __builtin_fmaf(r, r, s2v[j]);
                    }
                }
            }
        }
    }
    if (fstats) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a1 = group32_sum(s1[j]), a2 = group32_sum(s2v[j]);
            if (col == 0) { lsum[(wv * 8 + 4 * h + j) * 2 + 0] = a1; lsum[(wv * 8 + 4 * h + j) * 2 + 1] = a2; }
        }
        __syncthreads();
        float* wsum = lsum + 4 * 8 * 2;
        if (tid < 16) {
            float t = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) t += lsum[w2 * 16 + tid];
            wsum[tid] = t;
        }
        __syncthreads();
        fwd_stats_commit(p.fst, p.Cout, b, blockIdx.x, 0, 8, wsum, tid);
    }
}

// gradient with a K axis of <= 8 dY channels (bf16, or f32 with Cout <= 4 per 16 bytes ... two loads for 8): rows = Cin <= 32
template <int RT, bool FINAL>      // row tiles of 8 (Cin <= 8 RT); FINAL as above
__global__ __launch_bounds__(256) void conv_pw8_dgrad_kernel(ConvK p, long V) {
    constexpr int TT = 4;
    __shared__ float lsum[4 * RT * 8 * 2];
    __shared__ PwEpi edesc[RT * 2];
    const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, h = lane >> 5, wv = tid >> 6;
    const int b = blockIdx.y;
    if (tid < RT * 2) {
        const int cg = (tid / 2) * 8 + (tid % 2) * 4;
        PwEpi d;
        d.x = reinterpret_cast<const bf16_t*>(p.seg[0].x); d.dx = nullptr; d.C = 0; d.ok = 0; d.has_stats = 0; d.mval = 1.f;
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
                    d.dx = reinterpret_cast<bf16_t*>(sd.dx) + (long)b * sd.bstride + cl;
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
        edesc[tid] = d;
    }
    // A fragment: transposed image row = col (< Mpad), dY channels 4h .. 4h + 3
    s16x4 wf = {0, 0, 0, 0};
    if (col < p.Mpad) {
        const u32x2 w2 = gload<u32x2>(reinterpret_cast<const char*>(p.wpack) + (wvec_of(p, b) + col) * 16 + 8 * h);
        wf = __builtin_bit_cast(s16x4, w2);
    }
    const bool kf32 = p.y_f32 != 0;
    const char* src = reinterpret_cast<const char*>(p.dy) + (long)b * V * p.Cout * (kf32 ? 4 : 2);
    const bool k_ok = 4 * h < p.Cout;                       // this lane's four dY channels exist (Cout is 4 or 8 here)
    __syncthreads();
    const bf16_t* ex[RT];
    unsigned eC[RT], xS[RT];
    bool e_ok[RT], e_hs[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const PwEpi& e = edesc[i * 2 + h];
        ex[i] = e.x; eC[i] = (unsigned)e.C; e_ok[i] = e.ok != 0; e_hs[i] = e.has_stats != 0;
        xS[i] = e_hs[i] ? eC[i] : 0u;
    }
    float s1[RT][4], s2v[RT][4];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s1[i][j] = 0.f; s2v[i][j] = 0.f;
            const int row = 8 * i + 4 * h + j;
            if (FINAL && row < p.Cin) { s1[i][j] = p.pwm[((long)b * p.Cin + row) * 2]; s2v[i][j] = p.pwm[((long)b * p.Cin + row) * 2 + 1]; }
        }

    for (long v0 = ((long)blockIdx.x * 4 + wv) * (32 * TT); v0 < V; v0 += (long)gridDim.x * 4 * (32 * TT)) {
        u32x2 raw[TT];
        u32x2 xpre[RT][TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const long v = v0 + t * 32 + col;
            const long vc = v < V ? v : V - 1;
            if (kf32) {
                f32x4 f = {0.f, 0.f, 0.f, 0.f};
                f = gload<f32x4>(src + (vc * p.Cout + (k_ok ? 4 * h : 0)) * 4);
                unsigned pk[2];
                float rb[4];
                round4_bf16(f, pk, rb);
                raw[t] = (u32x2){pk[0], pk[1]};
            } else {
                raw[t] = gload<u32x2>(src + (vc * p.Cout + (k_ok ? 4 * h : 0)) * 2);
            }
            if (!k_ok) raw[t] = (u32x2){0u, 0u};
#pragma unroll
            for (int i = 0; i < RT; ++i) xpre[i][t] = gload<u32x2>(ex[i] + vc * xS[i]);
        }
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            f32x16 acc;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(wf, __builtin_bit_cast(s16x4, raw[t]), acc, 0, 0, 0);
            const long v = v0 + t * 32 + col;
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                const PwEpi& e = edesc[i * 2 + h];
                const float f_pos = e.mval, f_neg = e_hs[i] ? e.mval * p.slope : e.mval;
                const bool valid = e_ok[i] && v < V;
                const u32x2 xr = xpre[i][t];
                const float xv[4] = {__uint_as_float(xr.x << 16), __uint_as_float(xr.x & 0xffff0000u),
                                     __uint_as_float(xr.y << 16), __uint_as_float(xr.y & 0xffff0000u)};
                const float vf = (valid && e_hs[i]) ? 1.f : 0.f;
                f32x4 gv = {acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3]};      // rows 8 i + 4 h + j
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xh = (xv[j] - e.mean[j]) * e.rstd[j];
                    const float gg = gv[j] * (xh > 0.f ? f_pos : f_neg);
                    if (FINAL) {
                        gv[j] = e_hs[i] ? e.rstd[j] * (gg - s1[i][j] - xh * s2v[i][j]) : gg;
                    } else {
                        gv[j] = gg;
                        const float gs = gg * vf;
                        s1[i][j] += gs;
                        s2v[i][j] = __builtin_fmaf(gs, xh, s2v[i][j]);
                    }
                }
                if (valid) store4(e.dx + v * eC[i], gv);
            }
        }
    }
    if (FINAL) return;
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const bool st = e_ok[i] && e_hs[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a1 = group32_sum(s1[i][j]), a2 = group32_sum(s2v[i][j]);
            if (col == 0) {
                lsum[((wv * RT + i) * 8 + 4 * h + j) * 2 + 0] = st ? a1 : 0.f;
                lsum[((wv * RT + i) * 8 + 4 * h + j) * 2 + 1] = st ? a2 : 0.f;
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < RT * 8 * 2; i += 256) {
        const int row = i >> 1, k2 = i & 1;
        const float t = ((lsum[(0 * RT * 8 + row) * 2 + k2] + lsum[(1 * RT * 8 + row) * 2 + k2]) + lsum[(2 * RT * 8 + row) * 2 + k2]) +
                        lsum[(3 * RT * 8 + row) * 2 + k2];
        if (row < p.Cin && t != 0.f) {
            const long idx = (long)(blockIdx.x % p.sums_rep) * p.B * p.Cin * 2 + ((long)b * p.Cin + row) * 2 + k2;
            if (p.sums_fixed) {
                double v = (double)t * 1099511627776.0;
                v = fmin(fmax(v, -9.0e18), 9.0e18);
                atomicAdd(reinterpret_cast<unsigned long long*>(p.sums) + idx, (unsigned long long)(long long)llrint(v));
            } else {
                atomicAdd(&p.sums[idx], t);
            }
        }
    }
}

template <int NSTEP, int MT>
int launch_pw(ConvK& k, int K8, hipStream_t st) {
    const long V = (long)k.Do * k.Ho * k.Wo;
    const long chunks = (V + 64 * PW_TT(NSTEP) - 1) / (64 * PW_TT(NSTEP));      // voxels of a workgroup trip
    // about g_conv_target workgroups per launch, each at least two trips
    static const int wgs = []() { const char* e = getenv("PASSION_PW_WGS"); return e && atoi(e) > 0 ? atoi(e) : 0; }();
    const long target = wgs ? wgs : g_conv_target;
    long nwg_b = hmax(1, hmin(chunks / 2 > 0 ? chunks / 2 : 1, (target + k.B - 1) / k.B));
    k.fst.stats = nullptr;
    if (k.fst_want && !k.y_f32) {
        const int rc = fwd_stats_setup(k, (int)nwg_b, 1, st);
        if (rc) return rc < 0 ? rc : -rc;
    }
    passion_prof_tag("conv_pw_fwd_kernel<%d, %d>", NSTEP, MT);
    hipLaunchKernelGGL((conv_pw_fwd_kernel<NSTEP, MT>), dim3((unsigned)nwg_b, k.B), dim3(256), 0, st, k, K8, V);
    LAUNCH_RET();
    return 0;
}

}  // namespace

namespace pconv {

// 0 = launched, < 0 = HIP error, 1 = shape not covered (the caller goes on to the lean kernel).  k.wpack / k.Mpad set by the caller.
int launch_conv_pw_fwd(ConvK& k, int K8, hipStream_t st) {
    if (!g_conv_pw || k.residual || k.bd_rows) return 1;
    if (!(k.slope >= 0.f && k.slope <= 1.f)) return 1;
    if (K8 < 1 || K8 > 16 || k.Mpad > 32) return 1;
    if (k.y_f32 && k.fst_want) return 1;
    const long V = (long)k.Do * k.Ho * k.Wo;
    if (V < 32768) return 1;                                  // the deep levels: few voxels, many channels -- the lean kernel's ground
    if (g_conv_pw8 && K8 == 1 && k.nseg == 1 && k.seg[0].C == 8 && k.Cout <= 8 && k.Cout % 4 == 0) {
        const long chunks = (V + 1023) / 1024;
        const long nwg_b = hmax(1, hmin(chunks / 2 > 0 ? chunks / 2 : 1, (g_conv_target + k.B - 1) / k.B));
        k.fst.stats = nullptr;
        if (k.fst_want && !k.y_f32) {
            const int rc = fwd_stats_setup(k, (int)nwg_b, 1, st);
            if (rc) return rc < 0 ? rc : -rc;
        }
        passion_prof_tag("conv_pw8_fwd_kernel");
        hipLaunchKernelGGL(conv_pw8_fwd_kernel, dim3((unsigned)nwg_b, k.B), dim3(256), 0, st, k, V);
        LAUNCH_RET();
        return 0;
    }
    if (V * (long)(k.Cin > k.Cout ? k.Cin : k.Cout) >= (1L << 31)) return 1;      // 32-bit element offsets inside a sample
    const int nstep = (K8 + 3) / 4, mt = k.Mpad / 16;
#define PW_CASE(NS, M) if (nstep <= NS && mt == M) return launch_pw<NS, M>(k, K8, st);
    PW_CASE(1, 1) PW_CASE(2, 1) PW_CASE(4, 1) PW_CASE(1, 2) PW_CASE(2, 2) PW_CASE(4, 2)
#undef PW_CASE
    return 1;
}

// ---- the InstanceNorm-backward means of a pointwise layer FROM ITS WEIGHT GRADIENT.  LeakyReLU is positively
// homogeneous, a = lrelu(x_hat) = x_hat s with s = lrelu'(x_hat) in {1, slope}; with g = W^T dY and g' = mask s g:
//   sum_v g'[ci] x_hat[ci] = sum_co W[co][ci] dW_b[co][ci]                      (dW_b: this sample's weight gradient)
//   sum_v g'[ci]           = sum_co W[co][ci] (slope db_b[co] mask + (1 - slope) P_b[co][ci]),  P_b = sum_v dY[co] [x_hat[ci] > 0]
// (P_b: the second slab of wgrad_lean_kernel<1, *>; W: the bf16 values the dgrad multiplies with).  Two folds per sample and one
// tiny contraction replace the sums of the dgrad epilogue -- and with the means known BEFORE the dgrad, its epilogue writes dx.
// per-sample folds of both slabs in one launch: grid (ceil(nA / 16) + ceil(nB / 16), B)
__global__ __launch_bounds__(256) void pw_fold_sample_kernel(const float* __restrict__ partA, long nA, long strideA, float* __restrict__ outA,
                                                             const float* __restrict__ partB, long nB, long strideB, float* __restrict__ outB, int R) {
    __shared__ float red[16][17];
    const int li = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int xa = (int)((nA + 15) / 16);
    const bool second = (int)blockIdx.x >= xa;
    const float* part = second ? partB : partA;
    const long n = second ? nB : nA, stride = second ? strideB : strideA;
    float* out = second ? outB : outA;
    const long i = (long)(second ? blockIdx.x - xa : blockIdx.x) * 16 + li;
    const float* pb = part + (long)blockIdx.y * R * stride;
    float a = 0.f;
    if (i < n) {
        // eight loads in flight per thread, added in slot order (the sum's bits are those of the one-at-a-time loop: with up to 2048
        // slots per sample -- a batch-1 launch of the streaming weight gradient -- that loop was a chain of 128 memory latencies,
        // 35-50 us on the main stream for 1.2 MB)
        int r = sl;
        for (; r + 7 * 16 < R; r += 8 * 16) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = pb[(long)(r + 16 * u) * stride + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += t[u];
        }
        for (; r < R; r += 16) a += pb[(long)r * stride + i];
    }
    red[sl][li] = a;
    __syncthreads();
    if (sl == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) t += red[k2][li];
        out[(long)blockIdx.y * n + i] = t;
    }
}

// blocks 0 .. B - 1: the means of sample b; blocks >= B: the weight / bias gradient = the sum of the per-sample folds (sample order).
// dwb [B][Cout Cin + Cout] (= [dW_b | db_b]), pb [B][Cout Cin]; wdg: bf16 [Cout / 8][mpad][8] (rows = input channels)
__global__ __launch_bounds__(128) void pw_ident_kernel(ConvK p, const float* __restrict__ dwb, const float* __restrict__ pb, const bf16_t* __restrict__ wdg,
                                                       int mpad, float inv_v, float* __restrict__ out, float* __restrict__ dweight, float* __restrict__ dbias,
                                                       int bm) {
    const long nW = (long)p.Cout * p.Cin;
    if ((int)blockIdx.x >= p.B) {
        const long i = (long)(blockIdx.x - p.B) * 128 + threadIdx.x;
        if (i >= nW + p.Cout) return;
        float t = 0.f;
        for (int b2 = 0; b2 < p.B; ++b2) t += dwb[(long)b2 * (nW + p.Cout) + i];
        if (i < nW) dweight[i] = t; else dbias[i - nW] = t;
        return;
    }
    // physical sample b = the passes p * bm + b of a batched launch (SegD.bmod; bm = B without passes): the dgrad then runs once
    // over the masked sum of the passes' dY (conv.hip, block-diagonal form), so the means are the sums over the passes
    const int b = blockIdx.x, ci = threadIdx.x;
    if (ci >= p.Cin || b >= bm) return;
    const uint8_t* mk = nullptr;
    int mld = 0, mcol = 0;
    bool hs = false;
#pragma unroll
    for (int s2 = 0; s2 < PASSION_MAX_SEG; ++s2) {
        if (s2 < p.nseg) {
            const SegD& sd = p.seg[s2];
            if (ci >= sd.cbase && ci < sd.cbase + sd.C) {
                hs = sd.stats != nullptr && sd.dx != nullptr;
                mk = sd.mask; mld = sd.mask_ld; mcol = sd.mask_col;
            }
        }
    }
    float m1 = 0.f, m2 = 0.f;
    for (int bv = b; bv < p.B; bv += bm) {
        const float mval = (mk && mk[(long)bv * mld + mcol] == 0) ? 0.f : 1.f;
        const float* dw = dwb + (long)bv * (nW + p.Cout);
        const float* pp = pb + (long)bv * nW;
        for (int co = 0; co < p.Cout; ++co) {
            const float w = (float)wdg[((long)(co >> 3) * mpad + ci) * 8 + (co & 7)];
            m2 = __builtin_fmaf(w, dw[(long)co * p.Cin + ci], m2);
            m1 = __builtin_fmaf(w, p.slope * dw[nW + co] * mval + (1.f - p.slope) * pp[(long)co * p.Cin + ci], m1);
        }
    }
    out[((long)b * p.Cin + ci) * 2 + 0] = hs ? m1 * inv_v : 0.f;
    out[((long)b * p.Cin + ci) * 2 + 1] = hs ? m2 * inv_v : 0.f;
}

// The same for a 3x3x3 layer (27 taps; conv_wgrad_march.hip leaves the indicator slab): blocks 0 .. B - 1: the means of sample b, one
// thread per input channel; blocks >= B: the weight / bias gradient of every weight set = the sum of its samples' folds (sample order).
// dwb [B][Cout Cin 27 + Cout], pb [B][Cout Cin 27]; wdg: bf16 dgrad image [27][Cout / 8][mpad][8] of set 0, sets wset_el elements apart.
struct MarchOuts { float* dw[PASSION_MAX_GROUPS]; float* db[PASSION_MAX_GROUPS]; };
__global__ __launch_bounds__(128) void march_ident_kernel(ConvK p, const float* __restrict__ dwb, const float* __restrict__ pb, const bf16_t* __restrict__ wdg,
                                                          long wset_el, int mpad, float inv_v, float* __restrict__ out, MarchOuts mo, int G) {
    const long nW = (long)p.Cout * p.Cin * 27;
    const int bg = p.B / G;                                     // samples per weight set
    if ((int)blockIdx.x >= p.B) {
        const long i = (long)(blockIdx.x - p.B) * 128 + threadIdx.x;
        if (i >= nW + p.Cout) return;
        for (int g2 = 0; g2 < G; ++g2) {
            float t = 0.f;
            for (int b2 = g2 * bg; b2 < (g2 + 1) * bg; ++b2) t += dwb[(long)b2 * (nW + p.Cout) + i];
            if (i < nW) { if (mo.dw[g2]) mo.dw[g2][i] = t; }
            else if (mo.db[g2]) mo.db[g2][i - nW] = t;
        }
        return;
    }
    // 128 threads: (input channel ci, slice of the (co, tap) axis); slices meet in LDS in fixed order.  pb == nullptr: m2 only
    // ("m2 first", conv.hip): the indicator slab does not exist and m1 comes from the input gradient's own sums.
    __shared__ float red[2][128];
    const int b = blockIdx.x, ci = threadIdx.x % p.Cin, part = threadIdx.x / p.Cin, nparts = 128 / p.Cin;      // Cin <= 16 (launcher)
    const uint8_t* mk = nullptr;
    int mld = 0, mcol = 0;
    bool hs = false;
#pragma unroll
    for (int s2 = 0; s2 < PASSION_MAX_SEG; ++s2) {
        if (s2 < p.nseg) {
            const SegD& sd = p.seg[s2];
            if (ci >= sd.cbase && ci < sd.cbase + sd.C) {
                hs = sd.stats != nullptr && sd.dx != nullptr;
                mk = sd.mask; mld = sd.mask_ld; mcol = sd.mask_col;
            }
        }
    }
    const float mval = (mk && mk[(long)b * mld + mcol] == 0) ? 0.f : 1.f;
    const float* dw = dwb + (long)b * (nW + p.Cout);
    const float* pp = pb ? pb + (long)b * nW : nullptr;
    const bf16_t* w0 = wdg + (long)(b / bg) * wset_el;
    const int K8 = p.Cout / 8;
    float m1 = 0.f, m2 = 0.f;
    if (part < nparts) {
        for (int e0 = part; e0 < p.Cout * 27; e0 += nparts) {
            const int co = e0 / 27, t = e0 % 27;
            const float w = (float)w0[(((long)t * K8 + (co >> 3)) * mpad + ci) * 8 + (co & 7)];
            const long e = ((long)co * p.Cin + ci) * 27 + t;
            m2 = __builtin_fmaf(w, dw[e], m2);
            if (pp) m1 = __builtin_fmaf(w, p.slope * dw[nW + co] * mval + (1.f - p.slope) * pp[e], m1);
        }
    }
    red[0][threadIdx.x] = m1; red[1][threadIdx.x] = m2;
    __syncthreads();
    if (threadIdx.x < p.Cin) {
        float a1 = 0.f, a2 = 0.f;
        for (int q = 0; q < nparts; ++q) { a1 += red[0][q * p.Cin + ci]; a2 += red[1][q * p.Cin + ci]; }
        out[((long)b * p.Cin + ci) * 2 + 0] = hs ? a1 * inv_v : 0.f;
        out[((long)b * p.Cin + ci) * 2 + 1] = hs ? a2 * inv_v : 0.f;
    }
}

// the input gradient (conv_lean.hip mode 1 for ksize 1): K8 = blocks of 8 dY channels; k.wpack (transposed images), k.Mpad = pad16(Cin),
// k.sums_rep / sums_fixed set by the caller
// shapes whose gradient the kernels below take (so that conv.hip may plan the finish-free order: weight gradient first)
// `kblocks`: 8-channel blocks of the gradient's K axis (Cout / 8; nseg * Cout / 8 for the block-diagonal form of batched passes)
bool conv_pw_dgrad_covers(const ConvK& k, int kblocks, int bphys) {
    if (!g_conv_pw) return false;
    const long V = (long)k.Di * k.Hi * k.Wi;
    if (V < 32768 || k.Cin > 64) return false;
    if (k.y_f32) {                   // f32 dY of a logit head: the 8-channel form only
        if (!(g_conv_pw8 && kblocks == 1 && (k.Cout == 4 || k.Cout == 8) && pad16(k.Cin) <= 32)) return false;
    } else if (k.Cout % 8) return false;
    static const long pwf_min = []() { const char* e = getenv("PASSION_PWFINAL_MIN_M"); return e ? atol(e) : 8L; }();      // M elements (24 while the step was host-bound; 8 measured best under graph replay)
    if ((long)bphys * V * k.Cin < (pwf_min << 20)) return false;      // small tensors: the two extra launches and the lost dgrad || wgrad overlap cost more than the finish pass
    const int K8 = kblocks, mpad = pad16(k.Cin);
    for (int s = 0; s < k.nseg; ++s) if (k.seg[s].C % 8) return false;
    if (g_conv_pw8 && K8 == 1 && mpad <= 32) return true;
    const int nstep = (K8 + 3) / 4, mt = mpad / 16;
    if (K8 > 16) return false;
    return (nstep == 1 && (mt == 1 || mt == 2 || mt == 4)) || (nstep == 2 && (mt == 1 || mt == 2 || mt == 4)) ||
           (nstep <= 4 && nstep > 2 && (mt == 1 || mt == 2 || mt == 4));
}

// per-sample folds of the two partial slabs (one launch), then the contraction with the weights -> out [B][Cin][2] = (m1, m2) / V
// together with the weight / bias gradient itself (the sum of the per-sample folds): two launches in all
int launch_pw_ident(const ConvK& k, const float* wslab, long wslot, const float* sslab, int wgs_per_b, const void* wdg_bf16, float* dwb, float* pb,
                    float* out, float* dweight, float* dbias, int bm, hipStream_t st) {
    const long nW = (long)k.Cout * k.Cin;
    const unsigned gx = (unsigned)((nW + k.Cout + 15) / 16 + (nW + 15) / 16);
    hipLaunchKernelGGL(pw_fold_sample_kernel, dim3(gx, k.B), dim3(256), 0, st, wslab, nW + k.Cout, wslot, dwb, sslab, nW, nW, pb, wgs_per_b);
    LAUNCH_RET();
    const double V = (double)k.Di * k.Hi * k.Wi;
    hipLaunchKernelGGL(pw_ident_kernel, dim3((unsigned)(k.B + (nW + k.Cout + 127) / 128)), dim3(128), 0, st, k, (const float*)dwb, (const float*)pb,
                       (const bf16_t*)wdg_bf16, pad16(k.Cin), (float)(1.0 / V), out, dweight, dbias, bm);
    LAUNCH_RET();
    return 0;
}

// 3x3x3 (conv.hip: march_final): per-sample folds of the weight-gradient slab and the indicator slab (both with slot stride wslot),
// the means, and dW / db of each of the G weight sets.  dw_out[g] / db_out[g]: nullptr = not wanted.
int launch_march_ident(const ConvK& k, const float* wslab, long wslot, const float* sslab, int wgs_per_b, const void* wdg_bf16, long wset_el,
                       float* dwb, float* pb, float* out, float* const* dw_out, float* const* db_out, int G, hipStream_t st) {
    const long nW = (long)k.Cout * k.Cin * 27;
    if (k.Cin > 16) return -1024;
    const unsigned gx = (unsigned)((nW + k.Cout + 15) / 16 + (sslab ? (nW + 15) / 16 : 0));      // sslab == nullptr: no indicator slab, m2 only
    if (!sslab) pb = nullptr;
    hipLaunchKernelGGL(pw_fold_sample_kernel, dim3(gx, k.B), dim3(256), 0, st, wslab, nW + k.Cout, wslot, dwb, sslab, nW, wslot, pb, wgs_per_b);
    LAUNCH_RET();
    MarchOuts mo;
    for (int g = 0; g < PASSION_MAX_GROUPS; ++g) { mo.dw[g] = g < G ? dw_out[g] : nullptr; mo.db[g] = g < G ? db_out[g] : nullptr; }
    const double V = (double)k.Di * k.Hi * k.Wi;
    hipLaunchKernelGGL(march_ident_kernel, dim3((unsigned)(k.B + (nW + k.Cout + 127) / 128)), dim3(128), 0, st, k, (const float*)dwb, (const float*)pb,
                       (const bf16_t*)wdg_bf16, wset_el, pad16(k.Cin), (float)(1.0 / V), out, mo, G);
    LAUNCH_RET();
    return 0;
}

int launch_conv_pw_dgrad(ConvK& k, int K8, hipStream_t st) {
    if (!g_conv_pw) return 1;
    const long V = (long)k.Di * k.Hi * k.Wi;
    if (V < 32768) return 1;
    for (int s = 0; s < k.nseg; ++s) if (k.seg[s].C % 4) return 1;
    if (g_conv_pw8 && !k.bd_rows && K8 == 1 && (k.Cout == 4 || k.Cout == 8) && k.Mpad <= 32) {       // dY of <= 8 channels (bf16, or the f32 logit gradient)
        const int rt = (k.Cin + 7) / 8;
        const long chunks = (V + 511) / 512;
        const long nwg_b = hmax(1, hmin(chunks / 2 > 0 ? chunks / 2 : 1, (g_conv_target + k.B - 1) / k.B));
        passion_prof_tag("conv_pw8_dgrad_kernel<%d, %d>", rt, k.pwm ? 1 : 0);
#define PW8D(R) if (rt == R) { if (k.pwm) hipLaunchKernelGGL((conv_pw8_dgrad_kernel<R, true>), dim3((unsigned)nwg_b, k.B), dim3(256), 0, st, k, V); \
                            else hipLaunchKernelGGL((conv_pw8_dgrad_kernel<R, false>), dim3((unsigned)nwg_b, k.B), dim3(256), 0, st, k, V); }
        PW8D(1) PW8D(2) PW8D(3) PW8D(4)
#undef PW8D
        LAUNCH_RET();
        k.pw_done = 1;
        return 0;
    }
    if (k.y_f32) return 1;
    if (K8 < 1 || K8 > 16 || k.Mpad > 64) return 1;           // 128 rows: 512 registers and spills -- the lean kernel's
    const int nstep = (K8 + 3) / 4, mt = k.Mpad / 16;
#define PWD_CASE(NS, M) if (nstep <= NS && mt == M) return launch_pw_dgrad<NS, M>(k, K8, st);
    PWD_CASE(1, 1) PWD_CASE(1, 2) PWD_CASE(1, 4) PWD_CASE(2, 1) PWD_CASE(2, 2) PWD_CASE(2, 4) PWD_CASE(4, 1) PWD_CASE(4, 2) PWD_CASE(4, 4)
#undef PWD_CASE
    return 1;
}

}  // namespace pconv
