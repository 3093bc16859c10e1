// Lean bf16 wgrad for the stride-1 convs with 8-channel-aligned tensors:
//   dW[co][ci][tap] = sum_vox dY[vox][co] * act(x)[vox + tap][ci]      (+ dbias[co] = sum_vox dY[vox][co])
// Same decomposition, operand layouts and deterministic slab flush as conv_wgrad_bf16_kernel (conv_bf16.hip):
// D[m = co][n = (tap, ci)] with K = 32 voxels per v_mfma_f32_16x16x32_bf16, both operands k-strided in their
// voxel-major LDS images and read with ds_read_b64_tr_b16; a workgroup owns (a run of bricks of ONE sample) x (one
// 8-channel block of the virtual concat) x (MT row tiles) and keeps its partial sums in registers across bricks.
// What changed is the instruction diet -- the counters of the general kernel on the level-1 conv (128^3, 8 -> 8):
// 1620 VALU + 490 SALU instructions per wavefront and brick against 40 MFMAs.  Here the concat lookup, mask and
// InstanceNorm statistics are resolved once per workgroup, the source / dY / LDS offsets of every thread are
// brick-invariant constants (interior bricks: address = brick base + constant), the MFMA loop is fully unrolled
// over immediates, and brick coordinates advance incrementally.
#include "conv_common.h"
#include "prof.h"
#include "wgrad_lean_body.h"

using namespace pconv;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

using pconv_wl::tr_read8;
using pconv_wl::wgrad_lean_body;

template <int KS, int MT>
__global__ __launch_bounds__(256) void wgrad_lean_kernel(ConvK p, float* __restrict__ dW, float* __restrict__ dbias) {
    wgrad_clear(p);
    wgrad_lean_body<KS, MT>(p, dW, dbias, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x);
}


// ---------------------------------------------------------------------------------------------------------------
// First-layer weight gradient (reference nn.Conv3d(1, 8, 3, padding_mode='reflect'), mmformer.py:28): Cin = 1 -- one raw
// modality volume, no InstanceNorm in front -- and Cout = 8.  As an MFMA GEMM this is a 16 x 32 tile with 8 x 27 useful
// entries and a K axis (voxels) whose B operand would have to be gathered element by element; the generic bf16 wgrad
// kernel spent 136 us per launch on it.  Here a thread owns voxels, keeps the 9 x 8 partial sums of one depth tap (+ 8 bias sums) in
// registers as channel pairs (v_pk_fma_f32: two channels per instruction), and the workgroup folds them once at the end
// into its slot of the partial slab ([co][tap] | dbias, the same layout the other wgrad kernels fill).
__global__ __launch_bounds__(256) void wgrad_c1_kernel(ConvK p, float* __restrict__ dW, float* __restrict__ dbias, int nact) {
    wgrad_clear(p);
    // the slab has gridDim.x slots (the fold's contract); only the first `nact` workgroups of a row sum voxels, the others flush a
    // zero slot: with one workgroup per slot a thread owned < 3 voxels and the launch was all prologue + LDS fold (69 us for 38 MB)
    if ((int)blockIdx.x >= nact) {
        const int t = threadIdx.x, td0 = blockIdx.y;
        if (dW && t < 72) dW[(long)blockIdx.x * p.wslot + (t % 8) * 27 + td0 * 9 + t / 8] = 0.f;
        if (dbias && td0 == 0 && t < 8) dbias[(long)blockIdx.x * p.wslot + t] = 0.f;
        return;
    }
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    __shared__ float red[256 * 33];
    __shared__ float red2[8][32];
    const SegD& sd = p.seg[0];
    const int D = p.Di, H = p.Hi, W = p.Wi, tid = threadIdx.x;
    const int td = blockIdx.y;            // depth tap of this workgroup: 9 taps x 8 channels of partial sums per thread, so that
                                          // several waves fit a SIMD (all 27 taps = 260 registers = one wave, latency-bound)
    const unsigned V = (unsigned)D * H * W, total = (unsigned)p.B * V;        // < 2^31 (checked by the launcher)
    const bool reflect = p.pad_mode == PASSION_PAD_REFLECT;
    const bool pow2 = !(W & (W - 1)) && !(H & (H - 1)) && !(V & (V - 1));
    const int lw = __builtin_ctz(W), lh = __builtin_ctz(H), lv = __builtin_ctz(V);
    const bf16_t* dy = reinterpret_cast<const bf16_t*>(p.dy);
    f32x2 acc[9][4], bacc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        bacc[c] = (f32x2){0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t][c] = (f32x2){0.f, 0.f};
    }
    // Software pipeline: the 9 neighbour loads + the dY vector of the next two voxels are issued before the packed FMAs
    // of the current two.
    struct Vox { unsigned short x[9]; uint4 dy; unsigned okbits; };
    // A thread's voxels are `stride` apart.  When that is a whole number of planes (the 128^3 / 64^3 volumes: 256 workgroups x 256
    // threads = 4 / 16 planes) its (h, w) never changes: the reflected in-plane offsets and their validity bits are computed ONCE
    // (per voxel they were ~120 of the ~200 VALU instructions of a voxel-tap; the kernel is VALU-bound, not byte-bound)
    const unsigned HW = (unsigned)H * W;
    const unsigned stride = (unsigned)nact * 256u;
    const bool fixed_hw = stride % HW == 0u;
    int off9[9];
    unsigned okhw = 0u;
    auto hw_setup = [&](unsigned h, unsigned w) {
        int ih[3], iw[3];
        unsigned okh = 0, okw = 0;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            int ch = (int)h + t - 1, cw = (int)w + t - 1;
            if (reflect || (ch >= 0 && ch < H)) okh |= 1u << t;
            if (reflect || (cw >= 0 && cw < W)) okw |= 1u << t;
            ch = ch < 0 ? -ch : (ch >= H ? 2 * H - 2 - ch : ch);
            cw = cw < 0 ? -cw : (cw >= W ? 2 * W - 2 - cw : cw);
            ih[t] = min(max(ch, 0), H - 1); iw[t] = min(max(cw, 0), W - 1);
        }
        okhw = 0u;
#pragma unroll
        for (int th = 0; th < 3; ++th)
#pragma unroll
            for (int tw = 0; tw < 3; ++tw) {
                off9[th * 3 + tw] = ih[th] * W + iw[tw];
                if ((okh >> th) & (okw >> tw) & 1u) okhw |= 1u << (th * 3 + tw);
            }
    };
    {
        const unsigned v0 = (blockIdx.x * 256u + tid) % HW;
        hw_setup(v0 / (unsigned)W, v0 % (unsigned)W);
    }
    auto fetch = [&](unsigned i, Vox& o) {
        o.okbits = 0u;
        o.dy = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int t = 0; t < 9; ++t) o.x[t] = 0;
        if (i >= total) return;
        unsigned b, d;
        if (pow2) { b = i >> lv; d = (i >> (lw + lh)) & (unsigned)(D - 1); }
        else { b = i / V; d = (i - b * V) / HW; }
        if (!fixed_hw) {
            const unsigned r = (i - b * V) - d * HW;
            hw_setup(r / (unsigned)W, r % (unsigned)W);
        }
        o.dy = *reinterpret_cast<const uint4*>(dy + (long)i * 8);
        const bool present = !sd.mask || sd.mask[(long)b * sd.mask_ld + sd.mask_col];      // absent modality: its input is zero
        const unsigned short* xb = reinterpret_cast<const unsigned short*>(sd.x) + (long)sd.bsrc((int)b) * sd.bstride;
        int cd = (int)d + td - 1;
        const bool okd = reflect || (cd >= 0 && cd < D);
        cd = cd < 0 ? -cd : (cd >= D ? 2 * D - 2 - cd : cd);
        cd = min(max(cd, 0), D - 1);
        const unsigned short* pl = xb + (long)cd * HW;
#pragma unroll
        for (int t = 0; t < 9; ++t) o.x[t] = pl[off9[t]];
        if (present && okd) o.okbits = okhw;
    };
    auto accumulate = [&](const Vox& q) {
        const unsigned dw[4] = {q.dy.x, q.dy.y, q.dy.z, q.dy.w};
        f32x2 dy2[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {      // bf16 pairs -> f32: low half = even channel
            dy2[c] = (f32x2){__uint_as_float(dw[c] << 16), __uint_as_float(dw[c] & 0xffff0000u)};
            bacc[c] += dy2[c];
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float xv = __uint_as_float((unsigned)q.x[t] << 16);
            if (!((q.okbits >> t) & 1u)) xv = 0.f;
            const f32x2 x2 = {xv, xv};
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[t][c] = __builtin_elementwise_fma(x2, dy2[c], acc[t][c]);
        }
    };
    unsigned i = blockIdx.x * 256u + tid;
    Vox n0, n1;                            // (four voxels in flight with unconditional loads: 220 registers, two waves per SIMD, 65 us)
    fetch(i, n0);
    fetch(i + stride, n1);
    for (; i < total; i += 2 * stride) {
        const Vox c0 = n0, c1 = n1;
        fetch(i + 2 * stride, n0);
        fetch(i + 3 * stride, n1);
        accumulate(c0);
        accumulate(c1);
    }
    // fold over the 256 threads, 32 values at a time: value k = t * 8 + co (k < 72, t = th * 3 + tw), 72 + co = bias sums
    float* slot = dW ? dW + (long)blockIdx.x * p.wslot : nullptr;
#pragma unroll
    for (int ck = 0; ck < 3; ++ck) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int k = ck * 32 + j;
            float val = 0.f;
            if (k < 72) val = acc[k / 8][(k % 8) / 2][k % 2];
            else if (k < 80) val = bacc[(k - 72) / 2][(k - 72) % 2];
            red[tid * 33 + j] = val;
        }
        __syncthreads();
        {
            const int j = tid & 31, part = tid >> 5;
            float sacc = 0.f;
            for (int t = part * 32; t < part * 32 + 32; ++t) sacc += red[t * 33 + j];
            red2[part][j] = sacc;
        }
        __syncthreads();
        if (tid < 32) {
            float sacc = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) sacc += red2[q][tid];
            const int k = ck * 32 + tid;
            if (k < 72) { if (slot) slot[(k % 8) * 27 + td * 9 + k / 8] = sacc; }
            else if (k < 80 && dbias && td == 0) dbias[(long)blockIdx.x * p.wslot + (k - 72)] = sacc;
        }
    }
}

template <int KS>
int launch_wl(ConvK& k, int MT, int gx, int gy, float* dW, float* db, hipStream_t st) {
    dim3 grid(gx, gy);
    passion_prof_tag("wgrad_lean_kernel<%d, %d>", KS, MT == 2 ? 2 : 1);
    if (MT == 2) hipLaunchKernelGGL((wgrad_lean_kernel<KS, 2>), grid, dim3(256), 0, st, k, dW, db);
    else hipLaunchKernelGGL((wgrad_lean_kernel<KS, 1>), grid, dim3(256), 0, st, k, dW, db);
    LAUNCH_RET();
    return 0;
}

}  // namespace

namespace pconv {

// returns 0 = launched, < 0 = HIP error, 1 = shape not covered (caller uses conv_wgrad_bf16_kernel).
// gx = slots of the partial slab = workgroups in x (spare ones flush zeros), gy = (8-channel blocks) x (row-tile groups)
bool wgrad_lean_covers(const ConvK& k, int ksize, int stride, int MT, int gx, int gy) {
    if (stride != 1 || (k.Cout % 8 && !(k.y_f32 && k.Cout % 4 == 0 && k.Cout <= 8)) || gx < k.B) return false;
    if (!(k.slope >= 0.f && k.slope <= 1.f)) return false;      // the loader's LeakyReLU is max(t, slope * t)
    int K8 = 0;
    for (int s = 0; s < k.nseg; ++s) {
        if (k.seg[s].C % 8) return false;
        K8 += k.seg[s].C / 8;
    }
    return K8 * (pad16(k.Cout) / 16 / MT) == gy;       // else the plan was made for a different chunking
}

int launch_wgrad_lean(ConvK& k, int ksize, int stride, int MT, int gx, int gy, float* dW, float* db, hipStream_t st) {
    if (!wgrad_lean_covers(k, ksize, stride, MT, gx, gy)) return 1;
    if (ksize == 1) {
        const int rc = launch_wgrad_pw(k, MT, gx, gy, dW, db, st);                       // 1x1x1: the streaming kernel (conv_wgrad_pw.hip)
        if (rc <= 0) return rc;
    }
    {
        const int rc = launch_wgrad_march(k, ksize, stride, MT, gx, gy, dW, db, st);     // 3x3x3 with 8 output channels
        if (rc <= 0) return rc;
    }
    if (defer_next_wgrad() && wgrad_lean_defer(k, ksize, stride, MT, gx, gy, dW, db) == 0) return 0;      // parked for conv_deep.hip's pair launch
    k.Mpad = pad16(k.Cout);
    const int tiles_per_b = k.tilesD * k.tilesH * k.tilesW;
    const int wgs = gx / k.B;                          // workgroups per sample
    k.per = (tiles_per_b + wgs - 1) / wgs;
    if (ksize == 1) return launch_wl<1>(k, MT, gx, gy, dW, db, st);
    return launch_wl<3>(k, MT, gx, gy, dW, db, st);
}

// the same launch, parked for conv_deep.hip's pair kernel (3x3x3 layers that the marching weight gradient does not take)
int wgrad_lean_defer(ConvK& k, int ksize, int stride, int MT, int gx, int gy, float* dW, float* db) {
    if (ksize != 3 || MT != 2 || !wgrad_lean_covers(k, ksize, stride, MT, gx, gy) || k.y_f32) return 1;      // MT = 2: >= 32 output channels (the marching
                                                                                                               // weight gradients take 8 / 16)
    k.Mpad = pad16(k.Cout);
    const int tiles_per_b = k.tilesD * k.tilesH * k.tilesW;
    const int wgs = gx / k.B;
    k.per = (tiles_per_b + wgs - 1) / wgs;
    PendingWgrad& pw = pending_wgrad();
    pw.on = true; pw.k = k; pw.MT = MT; pw.gx = gx; pw.gy = gy; pw.dW = dW; pw.db = db;
    return 0;
}
int wgrad_lean_flush(hipStream_t st) {
    PendingWgrad& pw = pending_wgrad();
    if (!pw.on) return 0;
    pw.on = false;
    return launch_wl<3>(pw.k, pw.MT, pw.gx, pw.gy, pw.dW, pw.db, st);
}

// Cin = 1, Cout = 8, 3x3x3, stride 1, bf16 x and dY, no InstanceNorm in the loader: 0 = launched, 1 = not this shape
int launch_wgrad_c1(ConvK& k, int ksize, int stride, int gx, int gy, float* dW, float* db, hipStream_t st) {
    if (ksize != 3 || stride != 1 || k.nseg != 1 || k.Cin != 1 || k.seg[0].C != 1 || k.Cout != 8 || k.seg[0].stats || gy != 1 || k.y_f32) return 1;
    if ((long)k.B * k.Di * k.Hi * k.Wi >= (1L << 31)) return 1;
    passion_prof_tag("wgrad_c1_kernel");
    static const int act_env = getenv("PASSION_C1_ACT") ? atoi(getenv("PASSION_C1_ACT")) : 256;
    const int nact = act_env > 0 && act_env < gx ? act_env : gx;
    hipLaunchKernelGGL(wgrad_c1_kernel, dim3(gx, 3), dim3(256), 0, st, k, dW, db, nact);
    LAUNCH_RET();
    return 0;
}

}  // namespace pconv
