// Shared between the f32-MFMA (conv.hip) and bf16-MFMA (conv_bf16.hip) implicit-GEMM conv kernels:
// launch parameters, brick geometry, the reflect-adjoint tap decoding and the two epilogues
// (forward: bias + residual + store; dgrad: back through LeakyReLU, InstanceNorm-backward sums).
// Both kernels produce the same accumulator layout: D[row = channel][col = voxel] in 16x16 MFMA tiles,
// lane l holds column n = l & 15 and rows 4 * (l >> 4) + r, r = 0..3.
#pragma once
#include "common.h"
#include "passion_hip.h"

namespace pconv {

constexpr int PASSION_MAX_GROUPS = 4;
constexpr int SUMS_REP = 256;           // replicas of the InstanceNorm-backward sums (atomic contention): one-brick-per-workgroup kernels
constexpr int SUMS_SLOTS = 1024;         // capacity of the sums region in replicas: the marching dgrad (conv_march.hip) gives every workgroup of a sample its own slot
constexpr int SUMS_REP_LEAN = 16;       // persistent kernels add once per workgroup: few replicas, folded by the finish kernel itself
constexpr int TD = 4, TH = 8, TW = 8;   // output brick of one 256-thread workgroup (wave w = depth slice w)

struct SegD {
    const void* x;
    const float* stats;
    const uint8_t* mask;
    void* dx;
    long bstride;
    int C, mask_col, mask_ld, cbase;
    int bmod;   // > 0: the tensor has only `bmod` samples and sample b reads x[b % bmod] (several masked passes over the same
                // features batched into one launch); dx and the mask are still indexed by b
    __host__ __device__ int bsrc(int b) const { return bmod > 0 ? b % bmod : b; }
    const void* dx_add;   // backward: added into dx by the kernel that writes dx in its final form (passion_seg_t.dx_add); nullptr: none
    const void* dx_add2;  // ... and a second one (only with dx_add)
};

// InstanceNorm statistics of the conv's OUTPUT produced by the forward kernel itself (marching / lean bf16 kernels): every
// workgroup leaves the (sum, sum of squares) of the bf16 values it stored in its own slot, groups of STATS_GROUP workgroups
// are folded by whichever of them arrives last (ticket), the group sums by the group that arrives last -- fixed order both
// times, so the statistics repeat bit for bit -- and that workgroup writes (mean, rstd).  Replaces the separate statistics
// launches (one or two per conv) and their pass over y.
constexpr int STATS_GROUP = 16;
struct FwdStats {
    float* stats;          // [B][Cout][2] (mean, rstd); nullptr: not requested / not fused
    float* slots;          // [B][nwg][Cout][2] f32, plain stores (scratch)
    double* gslots;        // [B][ngroup][Cout][2] (scratch)
    unsigned* tickets;     // [B][ngroup + 1]: zero before the launch, left at zero
    int nwg, ngroup, ny;   // workgroups per sample along x, groups of STATS_GROUP, workgroups per x position (row-tile groups)
    int direct;            // few slots: one fold by the last workgroup of the sample instead of groups + final
    float eps;
    double inv_v;
};

struct ConvK {
    int B, Di, Hi, Wi, Do, Ho, Wo;
    int pad_mode, nseg, Cin, Cout;
    float slope;
    SegD seg[PASSION_MAX_SEG];
    const void* wpack;
    int Mpad;
    const float* bias;
    const void* residual;
    void* y;
    const void* dy;
    float* sums;
    int tilesD, tilesH, tilesW;
    int hi_t[3], hi_u[3];  // dgrad reflect: tap / source coordinate of the high-side mirror term per axis
    int y_f32;             // lean kernels: y (forward) / dy (backward) is f32 with Cout <= 8 -- the 4-class logit heads
    long wslot;            // wgrad: floats per workgroup slot of the partial slab ([dW | dbias])
    int sums_rep;          // dgrad: replicas of the InstanceNorm-backward sums this launch spreads its atomics over
    int sums_fixed;        // the replicas hold 64-bit fixed-point (2^-40) integers instead of floats (lean dgrad: order-independent atomics)
    int sums_float_only;   // caller needs float sums (two-phase backward of the depth-sharded path all-reduces them)
    int per, wres;         // bf16 kernel: bricks per (persistent) workgroup; all chunks' weight fragments resident in LDS
    int bd_rows, bd_k8;    // lean dgrad with block-diagonal weights (batched passes summed first, conv.hip): K block gb = (segment
                           // gb / bd_k8, block gb % bd_k8 of the real Cout); rows of another segment (bd_rows each) read zero weights
    int dbg;               // developer timing switch (PASSION_CONV_DBG): bit0 skip staging, bit1 skip MFMA loop, bit2 skip epilogue
    // Grouped launch along the batch (the four modality encoders of mmformer as ONE launch per layer): samples b with the same
    // b / bgroup share a weight set; group g's packed bf16 images start wg_vec 16-byte vectors after group g - 1's, its bias is
    // biasg[g].  0 = one weight set for the whole batch.
    int bgroup;
    long wg_vec;
    const float* biasg[PASSION_MAX_GROUPS];
    float* zero_ptr;       // wgrad kernels: zero_n floats every launch clears before anything else (the InstanceNorm-backward sum
    long zero_n;           // replicas the following dgrad adds into; the fold that used to clear them now rides in the finish launch)
    float* spos;           // 1x1x1 wgrad (conv_wgrad_lean.hip): second partial slab [slot][Cout][Cin] of sum_v dY[v][co] * [x_hat[v][ci] > 0]
                           // (the InstanceNorm-backward sums of the pointwise layers come from the weight gradient: conv_pw.hip)
    const float* pwm;      // pointwise "final" dgrad: (m1, m2) per (sample, input channel), already divided by V
    int march_k8;          // conv_march.hip: real 8-channel blocks when a launch pads its K axis to a power of two (0 = NB)
    int pw_done;           // set by the pointwise gradient launchers (conv.hip checks that a planned final dgrad really ran there)
    FwdStats fst;          // forward: statistics of y from the kernel's epilogue (fst.stats != nullptr)
    float* fst_want;       // forward: the caller wants statistics of y here (a launcher that fuses them moves it to fst.stats)
    float fst_eps;
};

#if defined(__HIPCC__)
// first thing in every weight-gradient kernel (conv.hip: why)
__device__ __forceinline__ void wgrad_clear(const ConvK& p) {
    if (p.zero_n <= 0) return;
    const long nb = (long)gridDim.x * gridDim.y, id = (long)blockIdx.y * gridDim.x + blockIdx.x;
    for (long z = id * blockDim.x + threadIdx.x; z < p.zero_n; z += nb * blockDim.x) p.zero_ptr[z] = 0.f;
}
#endif


// per-stream scratch of the library (api.cpp): `zero` bytes that are zero at every kernel start (kernels that use them leave
// them zero: tickets), `plain` bytes of uninitialised scratch.  Kernels of one stream run one after the other, so one region
// per stream is enough; it grows on demand (synchronising that stream once).
int stream_scratch(hipStream_t st, size_t zero_bytes, size_t plain_bytes, void** zero, void** plain);
// sets up k.fst for a launch of nwg x ny workgroups per sample (0 = ok)
int fwd_stats_setup(ConvK& k, int nwg, int ny, hipStream_t st);
int fwd_stats_setup_raw(FwdStats& f, int B, int C, int nwg, int ny, double inv_v, float eps, float* stats, hipStream_t st);

__host__ __device__ inline int pad16(int c) { return (c + 15) / 16 * 16; }
__host__ __device__ inline int pad8(int c) { return (c + 7) / 8 * 8; }

// bf16 matrix-core path (conv_bf16.hip)
long bf16_pack_elems(int Cin, int Cout, int ntap);
int pack_weight_bf16(const float* w, int Cin, int Cout, int ntap, void* out, hipStream_t st);
struct ConvK;
bool bf16_eligible(const ConvK& k, bool dgrad);
int launch_conv_bf16(ConvK& k, int ksize, int stride, bool dgrad, const void* packed_bf16, hipStream_t st);
int launch_conv_lean(ConvK& k, int ksize, int stride, int mode, const void* packed_bf16, hipStream_t st);   // conv_lean.hip
bool lean_covers(const ConvK& k, int stride, bool dgrad);
int launch_conv_march(ConvK& k, int ksize, int stride, int mode, const void* packed_bf16, hipStream_t st);   // conv_march.hip
extern int g_conv_march;
int launch_conv_pair(ConvK& k, int mode, hipStream_t st);   // conv_pair.hip: 8-channel K axis, <= 8 rows: two output planes per MFMA tile
extern int g_conv_pair;
// A deep-level weight gradient that has been PLANNED but not launched: conv3d_bwd_impl parks it here when the conv's input gradient is
// expected on conv_deep.hip, whose launcher then issues both as one grid (deep_pair_kernel); whatever is still parked after the input
// gradient's launch is issued on its own (wgrad_lean_flush).  One per host thread: a conv backward is planned and launched by one thread.
struct PendingWgrad { bool on; ConvK k; int MT, gx, gy; float* dW; float* db; };
PendingWgrad& pending_wgrad();                                                            // conv_deep.hip
bool& defer_next_wgrad();                                                                 // conv_deep.hip: set by conv3d_bwd_impl around its weight-gradient launch
int wgrad_lean_defer(ConvK& k, int ksize, int stride, int MT, int gx, int gy, float* dW, float* db);     // conv_wgrad_lean.hip: 0 = parked, 1 = not this shape
int wgrad_lean_flush(hipStream_t st);                                                     // launches a parked weight gradient, if any
int launch_conv_deep(ConvK& k, int ksize, int stride, int mode, hipStream_t st);   // conv_deep.hip: 32-channel-chunk K axes over small volumes (levels 3..5)
extern int g_conv_deep, g_deep_pair;
bool wgrad_pw_covers(const ConvK& k, bool y_f32);                                      // conv_wgrad_pw.hip: 1x1x1 weight gradient, streaming form
int launch_wgrad_pw(ConvK& k, int MT, int gx, int gy, float* dW, float* db, hipStream_t st);
extern int g_wgrad_pw, g_wgrad_pw_minvox;
int launch_conv_s2_dgrad(ConvK& k, const void* packed_bf16, hipStream_t st);   // conv_s2_dgrad.hip: stride-2 padded-domain dgrad, gather form
extern int g_conv_s2;
int launch_conv_pw_fwd(ConvK& k, int K8, hipStream_t st);   // conv_pw.hip: 1x1x1 forward as a streaming GEMM (wide volumes)
int launch_conv_pw_dgrad(ConvK& k, int K8, hipStream_t st);
bool conv_pw_dgrad_covers(const ConvK& k, int kblocks, int bphys);
int launch_pw_ident(const ConvK& k, const float* wslab, long wslot, const float* sslab, int wgs_per_b, const void* wdg_bf16, float* dwb, float* pb,
                    float* out, float* dweight, float* dbias, int bm, hipStream_t st);
extern int g_conv_pwfinal;
int launch_march_ident(const ConvK& k, const float* wslab, long wslot, const float* sslab, int wgs_per_b, const void* wdg_bf16, long wset_el,
                       float* dwb, float* pb, float* out, float* const* dw_out, float* const* db_out, int G, hipStream_t st);   // conv_pw.hip
extern int g_conv_marchfinal;
extern int g_conv_pw, g_conv_pw8;
bool conv_march_covers(const ConvK& k, int ksize, int stride, int mode, int* K8out);
int launch_wgrad_lean(ConvK& k, int ksize, int stride, int MT, int gx, int gy, float* dWslab, float* dbslab, hipStream_t st);   // conv_wgrad_lean.hip
bool wgrad_lean_covers(const ConvK& k, int ksize, int stride, int MT, int gx, int gy);
int launch_wgrad_march(ConvK& k, int ksize, int stride, int MT, int gx, int gy, float* dWslab, float* dbslab, hipStream_t st);   // conv_wgrad_march.hip: Cout = 8
extern int g_wgrad_march;
int launch_conv_c1_fwd(ConvK& k, int ksize, int stride, const void* packed_bf16, hipStream_t st);   // conv_c1_fwd.hip: Cin = 1, Cout = 8
extern int g_conv_c1;
int launch_wgrad_c1(ConvK& k, int ksize, int stride, int gx, int gy, float* dWslab, float* dbslab, hipStream_t st);   // first layer, Cin = 1
bool wgrad_c1m_covers(const ConvK& k, int ksize, int stride, int gx);                                                  // conv_c1_mfma.hip: the same on the matrix cores
int launch_wgrad_c1m(ConvK& k, int ksize, int stride, int gx, float* dWslab, float* dbslab, hipStream_t st);
int launch_conv_c1m_fwd(ConvK& k, int ksize, int stride, const void* packed_bf16, hipStream_t st);                     // ... and the forward
extern int g_c1_mfma;
int launch_conv_s2f(ConvK& k, int ksize, int stride, const void* packed_bf16, hipStream_t st);                         // conv_s2_fwd.hip: marching forward of the 8 -> 16 stride-2 conv
extern int g_conv_s2f;
bool wgrad_s2m_covers(const ConvK& k, int ksize, int stride, int gx);                                                   // conv_s2_wgrad.hip: its marching weight gradient
int launch_wgrad_s2m(ConvK& k, int ksize, int stride, int gx, float* dWslab, float* dbslab, hipStream_t st);
extern int g_wgrad_s2m;
extern int g_conv_lean;
extern int g_pack_f32;
int launch_wgrad_bf16(ConvK& k, int ksize, int stride, int MT, int gx, int gy, float* dWslab, float* dbslab, hipStream_t st);
int wgrad_bf16_chunk(const ConvK& k, int ksize, int stride);

// Workgroups are dealt round-robin to the 8 XCDs, each with a private L2.  Neighbouring bricks share halo planes
// and the 128-byte lines that straddle a brick edge, so consecutive brick indices must land on the SAME XCD:
// workgroup i (XCD i % 8) takes brick (i % 8) * (n / 8) + i / 8, i.e. every XCD owns one contiguous run of bricks.
// Measured on the level-1 forward conv: L2 fill traffic 5.6x the input (= 3 W-lines x 10/8 H x 6/4 D, no sharing
// across XCDs at all) before the swizzle.
__device__ __forceinline__ int xcd_swizzle(int i, int n) {
    const int n8 = n & ~7;
    return i < n8 ? (i & 7) * (n8 >> 3) + (i >> 3) : i;
}

// The (H, W) tiles on the volume's faces carry the reflect-padding adjoint of an input gradient (halo outputs as extra column tiles, or
// the mirror passes of the final-form kernels) and run up to 45 % longer than interior tiles.  An XCD hands out its run of workgroups
// in index order, two per CU at a time: with the face tiles FIRST in that order the short interior ones fill up behind them
// (longest-first list scheduling) instead of a CU drawing two long ones at the end of the launch (round 6: 16 -> 8 final-form input
// gradient over 4 x 128^3 214 -> 179 us).  Position q of the old row-major order -> tile (th, tw); needs TH, TW >= 3.
__device__ __forceinline__ void face_tiles_first(int TH, int TW, int& th, int& tw) {
    int q = th * TW + tw;
    if (q < 2 * TW) { th = q < TW ? 0 : TH - 1; tw = q % TW; }
    else if ((q -= 2 * TW) < 2 * (TH - 2)) { th = 1 + (q >> 1); tw = (q & 1) ? TW - 1 : 0; }
    else { q -= 2 * (TH - 2); th = 1 + q / (TW - 2); tw = 1 + q % (TW - 2); }
}

struct ConvK;
__device__ __forceinline__ const float* bias_of(const ConvK& p, int b);

// One axis of a tap entry.  a < KS: regular tap; a == KS: low-side mirror term; a == KS+1: high-side.
// Forward conv of stride S reads padded coordinate o*S + t - 1; 'reflect' maps -1 -> 1 and Sin -> Sin-2, so
// input index 1 (resp. Sin-2) receives an extra gradient term from (o = 0, t = 0) (resp. the unique (o, t) with
// o*S + t - 1 = Sin, precomputed on the host as hi_t / hi_u = o*S).
template <int KS, int HALO, bool DG, int S>
__device__ __forceinline__ void axis_tap(int a, int il, int i0, int Sin, int hi_t, int hi_u,
                                         int& t, int& pos, bool& valid) {
    if (a < KS) {
        t = a; valid = true;
        pos = DG ? (il + 2 * HALO - a) : (il * S + a);
    } else if (a == KS) {
        t = 0; valid = (i0 + il == 1); pos = 1 - i0;
    } else {
        t = hi_t; valid = (hi_t >= 0) && (i0 + il == Sin - 2); pos = hi_u - (i0 - HALO);
    }
    if (!valid) pos = 0;
}


__device__ __forceinline__ const float* bias_of(const ConvK& p, int b) { return p.bgroup ? p.biasg[b / p.bgroup] : p.bias; }
__device__ __forceinline__ long wvec_of(const ConvK& p, int b) { return p.bgroup ? (long)(b / p.bgroup) * p.wg_vec : 0; }

// End of a forward kernel with fused statistics.  wsum: LDS, [rows][2] partial sums of this workgroup for channels
// row0 .. row0 + rows - 1 of sample b (already combined over the workgroup's threads); wq: workgroup index inside the sample.
__device__ __forceinline__ void fwd_stats_commit(const FwdStats& fs, int Cout, int b, int wq, int row0, int rows,
                                                 const float* wsum, int tid) {
    __shared__ int fst_flag;
    const int NV = Cout * 2;
    float* slot = fs.slots + ((long)b * fs.nwg + wq) * NV;
    for (int i = tid; i < rows * 2; i += 256) {
        const int c = row0 + (i >> 1);
        if (c < Cout) st_agent(&slot[c * 2 + (i & 1)], wsum[i]);
    }
    stores_done();
    __syncthreads();                                           // waits for the stores (vmcnt 0) of every thread: slot before ticket
    unsigned* tk = fs.tickets + (long)b * (fs.ngroup + 1);
    __shared__ double fst_part[256];
    if (fs.direct) {
        // few enough slots (nwg * NV <= 16384): the workgroup that arrives last adds all of them, value v x SL slot slices per
        // thread (batches of eight loads), slices combined through LDS in order -- one hand-over instead of two
        if (tid == 0) fst_flag = atomicAdd(&tk[fs.ngroup], 1u) == (unsigned)(fs.nwg * fs.ny - 1);
        __syncthreads();
        if (!fst_flag) return;
        acquire_agent();
        const int SLd = 256 / NV, v = tid % NV, sl = tid / NV;      // NV <= 256 here
        double a = 0.0;
        if (sl < SLd) {
            const float* s0 = fs.slots + (long)b * fs.nwg * NV + v;
            int r = sl;
            for (; r + 7 * SLd < fs.nwg; r += 8 * SLd) {
                float t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = ld_agent(&s0[(long)(r + u * SLd) * NV]);
#pragma unroll
                for (int u = 0; u < 8; ++u) a += (double)t[u];
            }
            for (; r < fs.nwg; r += SLd) a += (double)ld_agent(&s0[(long)r * NV]);
        }
        fst_part[tid] = a;
        __syncthreads();
        if (tid < NV / 2) {
            double sm = 0.0, sq = 0.0;
            for (int q = 0; q < SLd; ++q) { sm += fst_part[q * NV + 2 * tid]; sq += fst_part[q * NV + 2 * tid + 1]; }
            const double mean = sm * fs.inv_v;
            double var = sq * fs.inv_v - mean * mean;
            if (var < 0) var = 0;
            fs.stats[((long)b * Cout + tid) * 2 + 0] = (float)mean;
            fs.stats[((long)b * Cout + tid) * 2 + 1] = (float)(1.0 / sqrt(var + (double)fs.eps));
        }
        if (tid == 0) __hip_atomic_store(&tk[fs.ngroup], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const int grp = wq / STATS_GROUP, gsz = min(STATS_GROUP, fs.nwg - grp * STATS_GROUP);
    if (tid == 0) fst_flag = atomicAdd(&tk[grp], 1u) == (unsigned)(gsz * fs.ny - 1);
    __syncthreads();
    if (!fst_flag) return;
    acquire_agent();
    double* gs = fs.gslots + ((long)b * fs.ngroup + grp) * NV;
    for (int i = tid; i < NV; i += 256) {
        const float* s0 = fs.slots + ((long)b * fs.nwg + (long)grp * STATS_GROUP) * NV + i;
        float v[STATS_GROUP];
#pragma unroll
        for (int u = 0; u < STATS_GROUP; ++u) v[u] = u < gsz ? ld_agent(&s0[(long)u * NV]) : 0.f;
        double a = 0.0;
#pragma unroll
        for (int u = 0; u < STATS_GROUP; ++u) a += (double)v[u];
        st_agent(&gs[i], a);
    }
    if (tid == 0) __hip_atomic_store(&tk[grp], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    stores_done();
    __syncthreads();
    if (tid == 0) fst_flag = atomicAdd(&tk[fs.ngroup], 1u) == (unsigned)(fs.ngroup - 1);
    __syncthreads();
    if (!fst_flag) return;
    acquire_agent();
    // the group sums: value v = (channel, moment) x SL slices of the groups per thread, slices combined through LDS in order
    const int SL = NV <= 256 ? 256 / NV : 1;
    for (int v0 = 0; v0 < NV; v0 += 256) {                     // one pass unless Cout > 128
        const int v = v0 + tid % min(NV, 256), sl = tid / min(NV, 256);
        double a = 0.0;
        if (v < NV && sl < SL) {
            const double* g0 = fs.gslots + (long)b * fs.ngroup * NV + v;
            int gi = sl;
            for (; gi + 7 * SL < fs.ngroup; gi += 8 * SL) {     // eight loads in flight, added in group order
                double t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = ld_agent(&g0[(long)(gi + u * SL) * NV]);
#pragma unroll
                for (int u = 0; u < 8; ++u) a += t[u];
            }
            for (; gi < fs.ngroup; gi += SL) a += ld_agent(&g0[(long)gi * NV]);
        }
        __syncthreads();
        fst_part[tid] = a;
        __syncthreads();
        const int c = (v0 >> 1) + tid;                         // thread c: both moments of channel c
        if (tid < min(NV, 256) / 2 && c < Cout) {
            double sm = 0.0, sq = 0.0;
            for (int q = 0; q < SL; ++q) { sm += fst_part[q * min(NV, 256) + 2 * tid]; sq += fst_part[q * min(NV, 256) + 2 * tid + 1]; }
            const double mean = sm * fs.inv_v;
            double var = sq * fs.inv_v - mean * mean;
            if (var < 0) var = 0;
            fs.stats[((long)b * Cout + c) * 2 + 0] = (float)mean;
            fs.stats[((long)b * Cout + c) * 2 + 1] = (float)(1.0 / sqrt(var + (double)fs.eps));
        }
    }
    if (tid == 0) __hip_atomic_store(&tk[fs.ngroup], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Marching input gradients (conv_march.hip, conv_pair.hip, conv_s2_dgrad.hip): the workgroup that took the last ticket of sample b folds
// the sample's nslots sums slots (slot r at sums + r * B * Cin * 2) into slot 0, in double, in a fixed order.  All 256 threads load:
// value v = tid % NVP, slot slice tid / NVP (NVP = the power of two >= 2 Cin), eight loads in flight per thread -- with 64 loading
// threads (the first version) the tail of a 1024-slot fold was ~14 us of a 52 us launch at batch 1.
__device__ __forceinline__ void fold_sums_slots(float* sums, int b, int B, int Cin, int nslots, int tid, double* fpart /* LDS [256] */) {
    const int NV = Cin * 2;                                    // <= 64
    const int NVP = NV <= 16 ? 16 : (NV <= 32 ? 32 : 64), SL = 256 / NVP;
    const int v = tid & (NVP - 1), sl = tid / NVP;
    double a = 0.0;
    if (v < NV) {
        const float* s0 = sums + ((long)b * Cin) * 2 + v;
        const long rstride = (long)B * Cin * 2;
        int r = sl;
        for (; r + 7 * SL < nslots; r += 8 * SL) {
            float tv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) tv[u] = ld_agent(&s0[(long)(r + u * SL) * rstride]);
#pragma unroll
            for (int u = 0; u < 8; ++u) a += (double)tv[u];
        }
        for (; r < nslots; r += SL) a += (double)ld_agent(&s0[(long)r * rstride]);
    }
    fpart[tid] = a;
    __syncthreads();
    if (tid < NV) {
        double t = 0.0;
        for (int q = 0; q < SL; ++q) t += fpart[q * NVP + tid];
        sums[((long)b * Cin) * 2 + tid] = (float)t;
    }
}

// v -> the four bf16 values that get stored (packed) and what they read back as
__device__ __forceinline__ void round4_bf16(const f32x4& v, unsigned (&packed)[2], float (&r)[4]) {
    union { bf16_t h[4]; unsigned u[2]; } t;
#pragma unroll
    for (int j = 0; j < 4; ++j) { t.h[j] = (bf16_t)v[j]; r[j] = (float)t.h[j]; }
    packed[0] = t.u[0]; packed[1] = t.u[1];
}

// ---------------------------------------------------------------------------------------------
template <typename TO, int MT>
__device__ __forceinline__ void epilogue_fwd(const ConvK& p, const f32x4 (&acc)[MT][4], int b, int mg,
                                             int od, int ow, const int (&oh4)[4], int kq) {
    TO* y = reinterpret_cast<TO*>(p.y);
    const TO* res = reinterpret_cast<const TO*>(p.residual);
    const float* bias = bias_of(p, b);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int oh = oh4[nt];
        if (od >= p.Do || oh >= p.Ho || ow >= p.Wo) continue;
        const long vox = (((long)b * p.Do + od) * p.Ho + oh) * p.Wo + ow;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int co = (mg * MT + mt) * 16 + kq * 4;
            if (co >= p.Cout) continue;
            f32x4 v = acc[mt][nt];
            if (bias) v += load4(bias + co);
            if (res) v += load4(res + vox * p.Cout + co);
            store4(y + vox * p.Cout + co, v);
        }
    }
}


// The same with the statistics of the stored values: per-lane sums over the lane's voxels, 16 voxel lanes by shuffles, four
// waves through LDS (fixed order), then the slot / ticket fold.  lsum: LDS float[4 * MT * 16 * 2 + MT * 16 * 2].
template <int MT>
__device__ __forceinline__ void epilogue_fwd_stats(const ConvK& p, const f32x4 (&acc)[MT][4], float* lsum, int b, int wq, int mg,
                                                   int od, int ow, const int (&oh4)[4], int kq, int n, int wv, int tid) {
    bf16_t* y = reinterpret_cast<bf16_t*>(p.y);
    const bf16_t* res = reinterpret_cast<const bf16_t*>(p.residual);
    const float* bias = bias_of(p, b);
    float s1[MT][4], s2[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) { s1[mt][j] = 0.f; s2[mt][j] = 0.f; }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int oh = oh4[nt];
        if (od >= p.Do || oh >= p.Ho || ow >= p.Wo) continue;
        const long vox = (((long)b * p.Do + od) * p.Ho + oh) * p.Wo + ow;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int co = (mg * MT + mt) * 16 + kq * 4;
            if (co >= p.Cout) continue;
            f32x4 v = acc[mt][nt];
            if (bias) v += load4(bias + co);
            if (res) v += load4(res + vox * p.Cout + co);
            unsigned pk[2];
            float rb[4];
            round4_bf16(v, pk, rb);
            *reinterpret_cast<uint2*>(y + vox * p.Cout + co) = make_uint2(pk[0], pk[1]);
#pragma unroll
            for (int j = 0; j < 4; ++j) { s1[mt][j] += rb[j]; s2[mt][j] = __builtin_fmaf(rb[j], rb[j], s2[mt][j]); }
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a1 = group16_sum(s1[mt][j]), a2 = group16_sum(s2[mt][j]);
            if (n == 0) {
                lsum[((wv * MT + mt) * 16 + kq * 4 + j) * 2 + 0] = a1;
                lsum[((wv * MT + mt) * 16 + kq * 4 + j) * 2 + 1] = a2;
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

// rows are input channels of the forward conv.  lsum: shared float[4 * MT*16*2] (one slice per wave), zeroed here.
// The workgroup's sums are formed in a fixed order (lanes by shuffles, waves through their own LDS slices) and added to the global
// replicas as 2^-40 fixed-point 64-bit integers when p.sums_fixed is set (order-independent: the f32 parity mode and the general
// bf16 kernels repeat bit for bit, like the lean / marching kernels), as float atomics otherwise (depth-sharded phases).
template <typename TO, int MT>
__device__ __forceinline__ void epilogue_dgrad(const ConvK& p, const f32x4 (&acc)[MT][4], float* lsum, int b, int mg,
                                               int od, int ow, const int (&oh4)[4], int kq, int n, int tid) {
    const int Od = p.Di, Oh = p.Hi, Ow = p.Wi;
    const int nrows = MT * 16;
    const int wv = tid >> 6;
    for (int i = tid; i < 4 * nrows * 2; i += 256) lsum[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int cg = (mg * MT + mt) * 16 + kq * 4;  // global input channel of row 0 of this lane
        int sg = -1;
        if (cg < p.Cin) {
#pragma unroll
            for (int s2 = 0; s2 < PASSION_MAX_SEG; ++s2)
                if (s2 < p.nseg && cg >= p.seg[s2].cbase && cg < p.seg[s2].cbase + p.seg[s2].C) sg = s2;
        }
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2v[4] = {0.f, 0.f, 0.f, 0.f};
        const bool seg_ok = (sg >= 0) && (p.seg[sg].dx != nullptr);
        if (seg_ok) {
            const SegD& sd = p.seg[sg];
            const int cl = cg - sd.cbase;
            float mval = 1.f;
            if (sd.mask) mval = sd.mask[(long)b * sd.mask_ld + sd.mask_col] ? 1.f : 0.f;
            f32x4 mean = {0.f, 0.f, 0.f, 0.f}, rstd = {1.f, 1.f, 1.f, 1.f};
            if (sd.stats) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    mean[j] = sd.stats[((long)sd.bsrc(b) * sd.C + cl + j) * 2 + 0];
                    rstd[j] = sd.stats[((long)sd.bsrc(b) * sd.C + cl + j) * 2 + 1];
                }
            }
            const TO* x = reinterpret_cast<const TO*>(sd.x) + (long)sd.bsrc(b) * sd.bstride;
            TO* dx = reinterpret_cast<TO*>(sd.dx) + (long)b * sd.bstride;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int oh = oh4[nt];
                if (od >= Od || oh >= Oh || ow >= Ow) continue;
                const long vox = (((long)od) * Oh + oh) * Ow + ow;
                f32x4 g = acc[mt][nt];
                if (sd.stats) {
                    f32x4 xv = load4(x + vox * sd.C + cl);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float xh = (xv[j] - mean[j]) * rstd[j];
                        float gg = g[j] * (xh > 0.f ? 1.f : p.slope) * mval;
                        g[j] = gg;
                        s1[j] += gg;
                        s2v[j] += gg * xh;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) g[j] *= mval;
                }
                store4(dx + vox * sd.C + cl, g);
            }
        }
        // reduce the per-channel sums over the 16 voxel lanes; one lane per (wave, row) leaves them in the wave's LDS slice
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a1 = group16_sum(s1[j]), a2 = group16_sum(s2v[j]);
            if (n == 0 && seg_ok && p.seg[sg].stats) {
                lsum[(wv * nrows + mt * 16 + kq * 4 + j) * 2 + 0] = a1;
                lsum[(wv * nrows + mt * 16 + kq * 4 + j) * 2 + 1] = a2;
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < nrows * 2; i += 256) {
        const int cg = mg * MT * 16 + (i >> 1);
        const float t = ((lsum[i] + lsum[nrows * 2 + i]) + lsum[2 * nrows * 2 + i]) + lsum[3 * nrows * 2 + i];      // waves in order
        if (cg < p.Cin && t != 0.f) {
            // thousands of bricks add into the same 2*Cin words: spread them over the replicas (folded afterwards)
            const long idx = (long)(blockIdx.x % p.sums_rep) * p.B * p.Cin * 2 + ((long)b * p.Cin + cg) * 2 + (i & 1);
            if (p.sums_fixed) {
                double v = (double)t * 1099511627776.0;                          // 2^40
                v = fmin(fmax(v, -9.0e18), 9.0e18);
                atomicAdd(reinterpret_cast<unsigned long long*>(p.sums) + idx, (unsigned long long)(long long)llrint(v));
            } else {
                atomicAdd(&p.sums[idx], t);
            }
        }
    }
}

}  // namespace pconv
