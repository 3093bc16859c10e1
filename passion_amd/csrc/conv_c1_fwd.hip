// Forward of the first encoder layer (reference nn.Conv3d(1, 8, 3, padding_mode='reflect'), code/models/mmformer.py:28): ONE input
// channel -- a raw modality volume, no InstanceNorm in front -- and 8 output channels.  As an implicit GEMM this is a K axis of
// 27 with 7/8 of every 8-channel block zero: the general bf16 kernel (conv_bf16.hip) spent 287 us on the grouped launch of the four
// modality encoders (4 x 128^3: 151 MB, 6.6 % of 8 TB/s), the slowest forward launch of the step.  Here a thread owns one (h, w)
// column of a 16 x 16 tile and marches along D: the 27 neighbours live in registers as three 3 x 3 planes (9 two-byte loads per
// output voxel instead of 27; padding, mask and offsets resolved once per thread), the 27 x 8 weights are broadcast from LDS,
// the products are packed f32 FMAs (v_pk_fma_f32: two output channels per instruction), the 8 bf16 outputs of a voxel leave
// as one 16-byte store.  Same values as the MFMA path: bf16 weights and inputs, f32 accumulation, bias, bf16 rounding; the
// InstanceNorm statistics of the output (the next layer normalises it) are summed from the rounded values and leave through the
// slot / ticket fold of conv_common.h.  The matching weight gradient is wgrad_c1_kernel (conv_wgrad_lean.hip).
#include "conv_common.h"
#include "prof.h"

namespace pconv { int g_conv_c1 = 1; }
using namespace pconv;

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int CT = 16;            // tile edge in H and W (256 threads)
constexpr int CD = 16;            // output planes per workgroup

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void conv_c1_fwd_kernel(ConvK p, int tilesH, int tilesW, int planes) {
    __shared__ __attribute__((aligned(16))) float wl[27 * 8];     // [tap][co]
    __shared__ float red[256 * 16];
    __shared__ float red2[16 * 16];
    __shared__ float wsum[16];
    const SegD& sd = p.seg[0];
    const int tid = threadIdx.x, b = blockIdx.y, wq = blockIdx.x;
    const int D = p.Di, H = p.Hi, W = p.Wi;
    const int tile = wq % (tilesH * tilesW), chunk = wq / (tilesH * tilesW);
    const int h = (tile / tilesW) * CT + (tid >> 4), w = (tile % tilesW) * CT + (tid & 15);
    const bool reflect = p.pad_mode == PASSION_PAD_REFLECT;
    if (tid < 216) {              // the bf16 forward image [tap][1 block][16 rows][8]: element 0 of a row is input channel 0
        const bf16_t* wimg = reinterpret_cast<const bf16_t*>(p.wpack) + wvec_of(p, b) * 8;
        wl[tid] = (float)wimg[((long)(tid >> 3) * 16 + (tid & 7)) * 8];
    }
    const float* bp = bias_of(p, b);
    f32x2 bia[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) bia[c] = bp ? (f32x2){bp[2 * c], bp[2 * c + 1]} : (f32x2){0.f, 0.f};
    const bool present = !sd.mask || sd.mask[(long)b * sd.mask_ld + sd.mask_col] != 0;      // absent modality: its input is zero
    const unsigned short* xb = reinterpret_cast<const unsigned short*>(sd.x) + (long)sd.bsrc(b) * sd.bstride;
    // the 3 x 3 in-plane neighbourhood of this thread's column: offsets and validity, once
    int off9[9];
    unsigned ok9 = 0u;
    {
        int ih[3], iw[3];
        unsigned okh = 0u, okw = 0u;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            int ch = h + t - 1, cw = w + t - 1;
            if (reflect || (ch >= 0 && ch < H)) okh |= 1u << t;
            if (reflect || (cw >= 0 && cw < W)) okw |= 1u << t;
            ih[t] = min(max(reflect_idx(ch, H), 0), H - 1); iw[t] = min(max(reflect_idx(cw, W), 0), W - 1);
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            off9[t] = ih[t / 3] * W + iw[t % 3];
            if (present && ((okh >> (t / 3)) & (okw >> (t % 3)) & 1u)) ok9 |= 1u << t;
        }
    }
    auto load_plane = [&](int pd, float (&dst)[9]) {
        const bool okd = reflect || (pd >= 0 && pd < D);
        const int cd = min(max(reflect_idx(pd, D), 0), D - 1);
        const unsigned short* pl = xb + (long)cd * H * W;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float f = __uint_as_float((unsigned)pl[off9[t]] << 16);
            dst[t] = (okd && ((ok9 >> t) & 1u)) ? f : 0.f;
        }
    };
    __syncthreads();
    const int d0 = chunk * planes, d1 = min(d0 + planes, D);
    float xa[9], xm[9], xc[9];
    load_plane(d0 - 1, xa);
    load_plane(d0, xm);
    f32x2 s[4], q[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { s[c] = (f32x2){0.f, 0.f}; q[c] = (f32x2){0.f, 0.f}; }
    bf16_t* yb = reinterpret_cast<bf16_t*>(p.y) + (((long)b * D * H + h) * W + w) * 8;
    for (int d = d0; d < d1; ++d) {
        load_plane(d + 1, xc);
        asm volatile("" ::: "memory");          // the weights are re-read from LDS every plane (216 registers otherwise: one wave per SIMD)
        f32x2 acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = (f32x2){0.f, 0.f};
#pragma unroll
        for (int td = 0; td < 3; ++td) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float xv = td == 0 ? xa[t] : (td == 1 ? xm[t] : xc[t]);
                const f32x2 xx = {xv, xv};
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(wl + (td * 9 + t) * 8);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(wl + (td * 9 + t) * 8 + 4);
                acc[0] = __builtin_elementwise_fma(xx, (f32x2){w0[0], w0[1]}, acc[0]);
                acc[1] = __builtin_elementwise_fma(xx, (f32x2){w0[2], w0[3]}, acc[1]);
                acc[2] = __builtin_elementwise_fma(xx, (f32x2){w1[0], w1[1]}, acc[2]);
                acc[3] = __builtin_elementwise_fma(xx, (f32x2){w1[2], w1[3]}, acc[3]);
            }
        }
        union { bf16_t hv[8]; u32x4 u; } o;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x2 v = acc[c] + bia[c];
            o.hv[2 * c] = (bf16_t)v.x; o.hv[2 * c + 1] = (bf16_t)v.y;
            const f32x2 r = {(float)o.hv[2 * c], (float)o.hv[2 * c + 1]};          // what gets stored
            s[c] = s[c] + r;
            q[c] = __builtin_elementwise_fma(r, r, q[c]);
        }
        *reinterpret_cast<u32x4*>(yb + (long)d * H * W * 8) = o.u;
#pragma unroll
        for (int t = 0; t < 9; ++t) { xa[t] = xm[t]; xm[t] = xc[t]; }
    }
    if (!p.fst.stats) return;
    // statistics: 256 threads x (8 sums, 8 sums of squares) -> 16 parts of 16 threads -> the workgroup's slot, fixed order
#pragma unroll
    for (int c = 0; c < 8; ++c) { red[tid * 16 + 2 * c] = s[c >> 1][c & 1]; red[tid * 16 + 2 * c + 1] = q[c >> 1][c & 1]; }
    __syncthreads();
    {
        const int o = tid & 15, part = tid >> 4;
        float t = 0.f;
        for (int l = part * 16; l < part * 16 + 16; ++l) t += red[l * 16 + o];
        red2[part * 16 + o] = t;
    }
    __syncthreads();
    if (tid < 16) {
        float t = 0.f;
#pragma unroll
        for (int part = 0; part < 16; ++part) t += red2[part * 16 + tid];
        wsum[tid] = t;                // [channel][sum, sum of squares]
    }
    __syncthreads();
    fwd_stats_commit(p.fst, 8, b, wq, 0, 8, wsum, tid);
}

}  // namespace

namespace pconv {

// 0 = launched, < 0 = HIP error, 1 = not this shape (the caller goes on to the general kernels).  bf16 x and y.
int launch_conv_c1_fwd(ConvK& k, int ksize, int stride, const void* packed_bf16, hipStream_t st) {
    if (!g_conv_c1 || ksize != 3 || stride != 1 || k.nseg != 1 || k.Cin != 1 || k.seg[0].C != 1 || k.Cout != 8) return 1;
    if (k.seg[0].stats || k.residual || k.y_f32 || k.seg[0].bmod > 0) return 1;
    if (k.Hi % CT || k.Wi % CT || (long)k.Di * k.Hi * k.Wi >= (1L << 28)) return 1;
    const int tilesH = k.Hi / CT, tilesW = k.Wi / CT;
    int planes = CD;
    while ((long)tilesH * tilesW * cdiv(k.Di, planes) > 1024) planes *= 2;      // <= 1024 statistics slots per sample
    const int nwg = tilesH * tilesW * cdiv(k.Di, planes);
    k.Mpad = pad16(k.Cout);
    k.wpack = packed_bf16;
    if (k.fst_want) {
        const int rc = fwd_stats_setup(k, nwg, 1, st);
        if (rc) return rc;
    }
    passion_prof_tag("conv_c1_fwd_kernel");
    hipLaunchKernelGGL(conv_c1_fwd_kernel, dim3(nwg, k.B), dim3(256), 0, st, k, tilesH, tilesW, planes);
    LAUNCH_RET();
    return 0;
}

}  // namespace pconv
