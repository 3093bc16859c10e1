// Fused multi-head self-attention for bf16 tensors with 64-wide heads (reference code/models/mmformer.py:192-227:
// softmax(q k^T * scale) -> attention dropout -> @ v, call sites :422-425 intra-modal N = (S/16)^3, :437-443 inter-modal 4N).
// No N x N tensor ever reaches HBM: forward keeps only the row log-sum-exp, backward recomputes the probabilities.
//
// gfx950 mapping (MI355X guide, "accumulator tile as the next MFMA's operand"):
//   * v_mfma_f32_32x32x16_bf16 everywhere.  The score tile is computed TRANSPOSED, S^T[key][q] = K . Q^T, so a lane owns one query
//     (column l & 31) and 16 of a tile's 32 keys in registers (rows (r & 3) + 8 (r >> 2) + 4 (l >> 5)): row max / row sum are
//     in-register plus ONE exchange with lane l ^ 32, and the probabilities -- converted to bf16 in place -- ARE the B operand
//     of the next product (O^T = V^T . P^T, dQ^T = K^T . dS^T), whose A operand comes out of the [key][d] LDS image through
//     ds_read_b64_tr_b16 in the matching k order.  The key-major backward kernel is the mirror image (S[q][key] = Q . K^T, a
//     lane owns one key): there P and dS are the B operands of dV^T = dO^T . P and dK^T = Q^T . dS.
//   * backward = two kernels, both free of atomics and cross-workgroup sums (bitwise reproducible): a query-major one for dQ and
//     a key-major one for dK, dV.  S and dP are recomputed in both: 7 MFMA products instead of 5, no dQ reduction.
//   * online softmax with a lazy rescale: the running reference only moves when a block exceeds it by 2^8 (bf16 / f32 have the
//     exponent range; the row sum uses the same reference, so nothing is approximated).
//   * attention dropout (round 5): a counter-based mask whose cost is 2.5 vector instructions per score instead of 5.  One 32-bit
//     word serves the two keys (2 kp, 2 kp + 1) of a query row: word(row, kp) = mix(R(row) + C(kp)), R and C two-round hashes of
//     (seed, (b H + h) N + q) and (seed, kp) that are formed ONCE per row / key pair (query-major kernels: R once per lane, C
//     for the block's 32 pairs by one wavefront into LDS; key-major kernel: C once per lane, R for the block's 64 queries into
//     LDS beside lse / delta), mix(x) = one xorshift + one 24-bit multiply-add (tests/_util.py restates it in numpy;
//     tests/test_hostlogic_cpu.py holds its keep rate and its row / column / diagonal correlations at the noise floor).
//     keep = (signed 16-bit field >= thr - 32768).  The forward applies it to the PACKED bf16 probabilities (v_pk_sub_i16 clamp,
//     v_pk_ashrrev_i16, v_bfi: 1.5 instructions per pair) and takes the row sum of the undropped ones from one more MFMA against
//     a tile of ones instead of 32 v_add_f32 per block; the 1 / (1 - p) factor is folded into the final row normalisation.
//     The backward regenerates the same bits (key-major: lanes 2i, 2i + 1 hold the two keys of a pair and the same queries --
//     each hashes every other row, one quad permute + one v_perm_b32 put both rows' fields of a lane's key into one register).
#include "common.h"
#include "passion_hip.h"
#include "prof.h"

int g_flash_split = 0;      // option "flash_split" / PASSION_FLASH_SPLIT: 0 = automatic, n = that many splits (1 = the unsplit kernels)
namespace pconv { int stream_scratch(hipStream_t st, size_t zero_bytes, size_t plain_bytes, void** zero, void** plain); }   // api.cpp
extern const unsigned* g_passion_step_words;      // api.cpp: passion_set_step_words

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int DH = 64;          // head width
constexpr int BLK = 64;         // keys (query-major kernels) / queries (key-major kernel) staged per step
constexpr int RS = 144;         // bytes per row of a [64][64] bf16 LDS image: 16 lanes reading 16 rows at one column hit 16
                                // different 4-bank groups (ds_read_b128 row reads); transposed reads of the same image see a
                                // 2-way conflict on half of their lanes
constexpr int TS = 192;         // ... of an image that is only read transposed (ds_read_b64_tr_b16: 4 consecutive rows x 64 B
                                // land in disjoint 16-bank windows)
constexpr float LAZY = 8.0f;    // log2 units

struct FlashP {
    const bf16_t* qkv;          // [B][N][3][H][64]
    bf16_t* out;                // [B][N][H][64]
    float* lse;                 // [B][H][N], log2 domain: log2 sum_key exp2(c * s)
    const bf16_t* dout;         // backward
    float* delta;               // [B][H][N] rowsum(dO o O)
    bf16_t* dqkv;
    int B, N, H;
    float c;                    // scale * log2(e)
    float scale;
    unsigned thr;               // keep when the signed 16-bit field >= thr - 32768 (thr = p * 65536); 0 = no dropout
    float inv_keep;
    unsigned s0, s1;            // seed halves
    const unsigned* soff;       // device words added to the seed at run time (a replayed HIP graph: passion_set_step_words), or nullptr
    // split-K (small batches: fewer query tiles than SIMDs): `nsplit` workgroups share a query (key) tile, each walks 1 / nsplit of
    // the key (query) blocks and leaves UNNORMALISED f32 partials; flash_combine_* merges them (fixed order: bit-reproducible)
    int nsplit;
    float* po;                  // query-major: [nsplit][B][H][N][64] partial O (forward) / dQ (backward); key-major: dK then dV
    float* pm;                  // forward: [nsplit][B][H][N] running reference (log2 units) ...
    float* pl;                  // ... and row sum of the split
};

// two rounds of (24-bit multiply-add, xorshift): every instruction is full rate on the vector ALU.  The per-row / per-key-pair halves
// of the dropout word (formed once per row / pair, not per score)
__device__ __forceinline__ unsigned drop_hash(unsigned pair, unsigned s0, unsigned s1) {
    unsigned x = pair ^ s0;
    x = __umul24(x, 0xB5297Au) + (x >> 9);
    x ^= x >> 15;
    x += s1;
    x = __umul24(x, 0x68E31Du) + (x >> 11);
    x ^= x >> 14;
    return x;
}
__device__ __forceinline__ unsigned row_word(unsigned row, unsigned s0, unsigned s1) { return drop_hash(row, s0, s1); }
__device__ __forceinline__ unsigned pair_word(unsigned kp, unsigned s0, unsigned s1) { return drop_hash(kp ^ 0x5BD1E995u, s1, s0 + 0x2545F491u); }
// the word of (row, key pair) from the sum of the two halves: 4 instructions
__device__ __forceinline__ unsigned drop_mix(unsigned x) {
    x ^= x >> 15;
    return __umul24(x, 0x68E31Du) + (x >> 11);
}
typedef short s16x2 __attribute__((ext_vector_type(2)));
// 0xFFFF in every 16-bit field of `u` that is DROPPED (signed field < thr - 32768), 0 where it is kept
__device__ __forceinline__ unsigned drop_fields(unsigned u, s16x2 thr2) {
    const s16x2 d = __builtin_elementwise_sub_sat(__builtin_bit_cast(s16x2, u), thr2);
    return __builtin_bit_cast(unsigned, d >> 15);
}

__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (__bf16)0.f;
    return z;
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int j = 0; j < 16; ++j) z[j] = 0.f;
    return z;
}

// registers 8s .. 8s+7 of a 32x32 accumulator as the bf16 B fragment of k-step s of the next product
__device__ __forceinline__ bf16x8 pack8(const f32x16& a, int s) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (__bf16)a[8 * s + j];
    return r;
}

// A fragment of a product that sums over the ROWS (r0 .. r0+15) of a [row][64] LDS image (row stride `stride` bytes) and whose
// output rows are the image's COLUMNS c0 .. c0+31, in the k order of pack8(): lane (column c0 + 16 * ((l >> 4) & 1) + (l & 15),
// half h = l >> 5), element j <-> image row r0 + 8 (j >> 2) + 4 h + (j & 3).
__device__ __forceinline__ bf16x8 tr_frag(const char* img, int stride, int r0, int c0, int lane) {
    const int n = lane & 15, g = lane >> 4, h = g >> 1, q4 = n >> 2, pq = n & 3;
    const char* p0 = img + (r0 + 4 * h + q4) * stride + (c0 + 16 * (g & 1) + 4 * pq) * 2;
    const char* p1 = p0 + 8 * stride;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p1));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// row fragment: lane (row r0 + (l & 31), half h), 8 consecutive columns 16 s + 8 h of a [row][64] image
__device__ __forceinline__ bf16x8 row_frag(const char* img, int stride, int r0, int s, int lane) {
    return *reinterpret_cast<const bf16x8*>(img + (r0 + (lane & 31)) * stride + (16 * s + 8 * (lane >> 5)) * 2);
}

// 64 rows x 64 columns of one head's q / k / v (or dout) slice -> registers (NT threads, 512 16-byte chunks), zero past N
template <int NT>
__device__ __forceinline__ void fetch_tile(u32x4 (&r)[512 / NT], const bf16_t* base, long rowstride, int row0, int N, int tid) {
#pragma unroll
    for (int i = 0; i < 512 / NT; ++i) {
        const int ci = i * NT + tid, row = ci >> 3, c8 = ci & 7;
        r[i] = (u32x4){0u, 0u, 0u, 0u};
        if (row0 + row < N) r[i] = *reinterpret_cast<const u32x4*>(base + (long)(row0 + row) * rowstride + c8 * 8);
    }
}
template <int NT>
__device__ __forceinline__ void commit_tile(const u32x4 (&r)[512 / NT], char* img, int stride, int tid) {
#pragma unroll
    for (int i = 0; i < 512 / NT; ++i) {
        const int ci = i * NT + tid, row = ci >> 3, c8 = ci & 7;
        *reinterpret_cast<u32x4*>(img + row * stride + c8 * 16) = r[i];
    }
}

// max / sum of a value over the two lanes l, l ^ 32 that share a query: v_permlane32_swap leaves [lo, lo] and [hi, hi]
__device__ __forceinline__ float half_max(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_sum(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

// --------------------------------------------------------------------------------------------------------------
// (Round 5, built, measured, removed: an eight-wave "ping-pong" form of these kernels -- waves in two groups held half a block apart by
// workgroup barriers, one group in its MFMA segment (second product of block j - 1 + scores of block j) while the other runs the softmax
// arithmetic, K / V in a ring of three LDS slots, groups assigned by the SIMD a wave reports in HW_REG_HW_ID.  Correct on every test shape and
// 1.7x SLOWER at B = 16, N = 2048 (forward 0.49 ms against 0.28): alone in its phase, the vector segment pays every latency its partner
// used to cover -- an added vector instruction cost 17 cycles there, 5 in the four-wave kernel below.  A start-up stagger of the odd wave
// slot changed nothing either.  What did pay is fewer vector instructions per score: see the dropout note at the top.)
// query-major kernels: MODE 0 = forward (out, lse), MODE 1 = backward dQ; DROP: attention dropout on (its own instantiation: one path
// through the loop -- with both in one kernel the register allocator spilled the dQ kernel's accumulators around the branch)
// grid (ceil(N / (32 NW)), H, B), block 64 NW: wave w owns queries q0 .. q0+31
template <int NW, int MODE, bool DROP>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2))) void flash_q_kernel(FlashP p) {
    const unsigned ds0 = p.s0 + (p.soff ? p.soff[0] : 0u), ds1 = p.s1 + (p.soff ? p.soff[1] : 0u);
    constexpr int NT = 64 * NW;
    constexpr int KIMG = BLK * RS, VIMG = (MODE == 0) ? BLK * TS : BLK * RS;
    constexpr int CTAB = 128;                                       // the block's 32 key-pair words C(kp), as [h][16] in a lane's register order
    constexpr int BUFQ = KIMG + VIMG + CTAB;
    extern __shared__ __attribute__((aligned(16))) char lds[];      // [2][K image | V image | pair words]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, h = lane >> 5;
    const int hh = blockIdx.y, b = blockIdx.z, N = p.N;
    const long rs = 3L * p.H * DH;
    const bf16_t* qb = p.qkv + (long)b * N * rs + hh * DH;
    const bf16_t* kbase = qb + (long)p.H * DH;
    const bf16_t* vbase = qb + 2L * p.H * DH;
    const int split = blockIdx.x % p.nsplit;
    const int q0 = ((blockIdx.x / p.nsplit) * NW + wv) * 32, q = q0 + (lane & 31);
    const bool qok = q < N;
    const int nblk_all = (N + BLK - 1) / BLK;
    const int kb0 = (int)((long)split * nblk_all / p.nsplit), nblk = (int)((long)(split + 1) * nblk_all / p.nsplit);   // key blocks [kb0, nblk)

    // B operands held for the whole kernel: Q^T (and dO^T) of this wave's 32 queries
    bf16x8 qf[4], dof[MODE == 1 ? 4 : 1];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        qf[s] = zero8();
        if (qok) qf[s] = *reinterpret_cast<const bf16x8*>(qb + (long)q * rs + 16 * s + 8 * h);
        if (MODE == 1) {
            dof[s] = zero8();
            if (qok) dof[s] = *reinterpret_cast<const bf16x8*>(p.dout + ((long)b * N + q) * (p.H * DH) + hh * DH + 16 * s + 8 * h);
        }
    }
    float lse2 = 0.f, dlt = 0.f;
    if (MODE == 1 && qok) {
        lse2 = p.lse[((long)b * p.H + hh) * N + q];
        dlt = p.delta[((long)b * p.H + hh) * N + q];
    }
    // dropout: this lane's row half of every word it will need, once; the pair halves of a block come from LDS
    const unsigned rq = DROP ? row_word((unsigned)(b * p.H + hh) * (unsigned)N + (unsigned)q, ds0, ds1) : 0u;
    const short ths = (short)((int)p.thr - 32768);
    const s16x2 thr2 = {ths, ths};
    const int thr_hi = ((int)p.thr - 32768) * 65536;
    // thread t < 32 forms the word of the block's pair t and stores it where lane half h2 = (t >> 1) & 1 reads its i-th register pair:
    // pair t = kt * 16 + 4 * (r >> 2) + 2 h2 + j  <->  i = kt * 8 + 2 * (r >> 2) + j
    const int ct_slot = ((tid >> 1) & 1) * 16 + (tid >> 4) * 8 + 2 * ((tid >> 2) & 3) + (tid & 1);
    auto put_pairs = [&](int kb_, char* buf) {
        if (DROP && tid < 32) reinterpret_cast<unsigned*>(buf + KIMG + VIMG)[ct_slot] = pair_word((unsigned)(kb_ * (BLK / 2) + tid), ds0, ds1);
    };

    f32x16 o[2] = {zero16(), zero16()};          // forward: O^T[d][q]; backward: dQ^T[d][q]
    f32x16 lacc = zero16();                      // forward with dropout: every register = the running row sum of this lane's query (ones . P^T)
    float l_run = 0.f;                           // forward without dropout: the row sum in f32, as before round 5 -- the evaluation / parity path
                                                 // keeps its arithmetic (the sum of the bf16-rounded probabilities moved the bf16 PASSION
                                                 // trajectory by one near-tie gate decision: test_trajectory_gpu)
    float m_run = -INFINITY;
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;

    u32x4 kr[512 / NT], vr[512 / NT];
    fetch_tile<NT>(kr, kbase, rs, kb0 * BLK, N, tid);
    fetch_tile<NT>(vr, vbase, rs, kb0 * BLK, N, tid);
    commit_tile<NT>(kr, lds + (kb0 & 1) * BUFQ, RS, tid);
    commit_tile<NT>(vr, lds + (kb0 & 1) * BUFQ + KIMG, MODE == 0 ? TS : RS, tid);
    put_pairs(kb0, lds + (kb0 & 1) * BUFQ);
    __syncthreads();

    for (int kb = kb0; kb < nblk; ++kb) {
        const char* kimg = lds + (kb & 1) * BUFQ;
        const char* vimg = kimg + KIMG;
        const u32x4* ctab = reinterpret_cast<const u32x4*>(vimg + VIMG) + 4 * h;
        if (kb + 1 < nblk) {
            fetch_tile<NT>(kr, kbase, rs, (kb + 1) * BLK, N, tid);
            fetch_tile<NT>(vr, vbase, rs, (kb + 1) * BLK, N, tid);
        }
        // ---- S^T = K . Q^T (and dP^T = V . dO^T): two 32-key tiles.  All A fragments of the step are requested before the first
        // MFMA (the compiler otherwise alternates read / wait / MFMA on two registers and every LDS latency is exposed)
        f32x16 st[2], dp[MODE == 1 ? 2 : 1];
        {
            bf16x8 kfr[2][4];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int s = 0; s < 4; ++s) kfr[kt][s] = row_frag(kimg, RS, kt * 32, s, lane);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                st[kt] = zero16();
#pragma unroll
                for (int s = 0; s < 4; ++s) st[kt] = MFMA32(kfr[kt][s], qf[s], st[kt]);
            }
            if (MODE == 1) {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int s = 0; s < 4; ++s) kfr[kt][s] = row_frag(vimg, RS, kt * 32, s, lane);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    dp[kt] = zero16();
#pragma unroll
                    for (int s = 0; s < 4; ++s) dp[kt] = MFMA32(kfr[kt][s], dof[s], dp[kt]);
                }
            }
        }
        // transposed fragments of the second product: requested now, consumed after the softmax arithmetic
        const char* timg = (MODE == 0) ? vimg : kimg;
        constexpr int tstride = (MODE == 0) ? TS : RS;
        bf16x8 tfr[2][2][2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) tfr[kt][s][dt] = tr_frag(timg, tstride, kt * 32 + 16 * s, dt * 32, lane);
        const int key0 = kb * BLK;
        const bool ragged = key0 + BLK > N;
        if (ragged) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (key0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h >= N) st[kt][r] = -INFINITY;
        }
        // pair halves of this lane's 16 register pairs: ctab[2 kt + s][jj] <-> registers 8 s + 2 jj, + 1 of tile kt (read where they are
        // used: held for the whole block they cost the dQ kernel its 256-register budget)
        if (MODE == 0) {
            // ---- online softmax, lazy reference
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[kt][r]);
            mx = half_max(mx) * p.c;
            const bool move = mx > m_run + LAZY;
            if (__any(move)) {
                const float m_new = move ? mx : m_run;
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);     // first block: exp2(-inf) = 0
                m_run = m_new;
                lacc[0] *= alpha;                        // the other fifteen registers hold the same sum and are never read
                l_run *= alpha;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
            }
            float psum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    st[kt][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kt][r], p.c, -m_run));
                    if (!DROP) psum += st[kt][r];
                }
            l_run += psum;
            // ---- row sum (DROP): ones . P^T on the matrix pipe (all keys of the tile, both lane halves); O^T += V^T . (P^T, dropped):
            // the mask is applied to the packed bf16 pairs (register pair (r, r + 1) = keys 2 kp, 2 kp + 1 = the word's two fields)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 pu = pack8(st[kt], s);
                    if (DROP) lacc = MFMA32(ones, pu, lacc);
                    bf16x8 pf = pu;
                    if (DROP) {
                        u32x4 pw = __builtin_bit_cast(u32x4, pu);
                        const u32x4 c4 = ctab[2 * kt + s];
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) pw[jj] &= ~drop_fields(drop_mix(rq + c4[jj]), thr2);
                        pf = __builtin_bit_cast(bf16x8, pw);
                    }
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) o[dt] = MFMA32(tfr[kt][s][dt], pf, o[dt]);
                }
        } else {
            // ---- P^T from the saved log-sum-exp; dS^T = P^T o (dP^T_dropped - delta)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) st[kt][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kt][r], p.c, -lse2));
            if (DROP) {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const unsigned u = drop_mix(rq + ctab[2 * kt + (r >> 3)][(r & 7) >> 1]);
                        const float t0 = (short)(u & 0xffffu) >= ths ? p.inv_keep : 0.f, t1 = (int)u >= thr_hi ? p.inv_keep : 0.f;
                        st[kt][r] *= __builtin_fmaf(dp[kt][r], t0, -dlt);
                        st[kt][r + 1] *= __builtin_fmaf(dp[kt][r + 1], t1, -dlt);
                    }
            } else {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) st[kt][r] *= (dp[kt][r] - dlt);
            }
            // ---- dQ^T += K^T . dS^T
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 pf = pack8(st[kt], s);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) o[dt] = MFMA32(tfr[kt][s][dt], pf, o[dt]);
                }
        }
        if (kb + 1 < nblk) {
            char* nk = lds + ((kb + 1) & 1) * BUFQ;
            commit_tile<NT>(kr, nk, RS, tid);
            commit_tile<NT>(vr, nk + KIMG, MODE == 0 ? TS : RS, tid);
            put_pairs(kb + 1, nk);
        }
        __syncthreads();
    }

    if (p.nsplit > 1) {
        // partials of this split: O^T (or dQ^T) unnormalised in f32, [split][b][h][q][d]; forward also (reference, row sum)
        const long row = (((long)split * p.B + b) * p.H + hh) * N + q;
        if (qok) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) p.po[row * DH + dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] = o[dt][r];
        }
        if (MODE == 0) {
            const float l_tot = DROP ? lacc[0] : half_sum(l_run);
            if (qok && h == 0) { p.pm[row] = m_run; p.pl[row] = l_tot; }
        }
        return;
    }
    // ---- epilogue: scale, transpose through LDS (the tile buffers are free after the last barrier), whole 128-byte rows out
    float f;
    if (MODE == 0) {
        const float l_tot = DROP ? lacc[0] : half_sum(l_run);
        f = p.inv_keep / l_tot;
        if (qok && h == 0) p.lse[((long)b * p.H + hh) * N + q] = m_run + __builtin_amdgcn_logf(l_tot);
    } else {
        f = p.scale;
    }
    char* obuf = lds + wv * (32 * RS);            // [32 queries][64 d] bf16 per wave
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            union { bf16_t hv[4]; uint2 u; } t;
#pragma unroll
            for (int j = 0; j < 4; ++j) t.hv[j] = (bf16_t)(o[dt][4 * g4 + j] * f);
            *reinterpret_cast<uint2*>(obuf + (lane & 31) * RS + (dt * 32 + 8 * g4 + 4 * h) * 2) = t.u;
        }
    __syncthreads();
    bf16_t* dst = (MODE == 0) ? p.out + (long)b * N * (p.H * DH) + hh * DH : p.dqkv + (long)b * N * rs + hh * DH;
    const long drs = (MODE == 0) ? (long)p.H * DH : rs;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ci = i * 64 + lane, row = ci >> 3, c8 = ci & 7;
        if (q0 + row < N)
            *reinterpret_cast<u32x4*>(dst + (long)(q0 + row) * drs + c8 * 8) = *reinterpret_cast<const u32x4*>(obuf + row * RS + c8 * 16);
    }
}

// --------------------------------------------------------------------------------------------------------------
// key-major backward kernel: dK, dV.  grid (ceil(N / (32 NW)), H, B); wave w owns keys k0 .. k0+31 and walks the queries in
// steps of 64 (Q and dO staged in LDS, read by rows for S / dP and transposed for dK^T / dV^T; lse, delta beside them)
template <int NW, bool DROP>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2))) void flash_k_kernel(FlashP p) {
    const unsigned ds0 = p.s0 + (p.soff ? p.soff[0] : 0u), ds1 = p.s1 + (p.soff ? p.soff[1] : 0u);
    constexpr int NT = 64 * NW;
    constexpr int IMG = BLK * RS, BUF = 2 * IMG + 3 * BLK * 4;
    extern __shared__ __attribute__((aligned(16))) char lds[];      // [2][Q image | dO image | lse[64] | delta[64] | row words[64]]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, h = lane >> 5;
    const int hh = blockIdx.y, b = blockIdx.z, N = p.N;
    const long rs = 3L * p.H * DH, os = (long)p.H * DH;
    const bf16_t* qb = p.qkv + (long)b * N * rs + hh * DH;
    const bf16_t* dob = p.dout + (long)b * N * os + hh * DH;
    const float* lseb = p.lse + ((long)b * p.H + hh) * N;
    const float* dltb = p.delta + ((long)b * p.H + hh) * N;
    const int split = blockIdx.x % p.nsplit;
    const int k0 = ((blockIdx.x / p.nsplit) * NW + wv) * 32, key = k0 + (lane & 31);
    const bool kok = key < N;
    const int nblk_all = (N + BLK - 1) / BLK;
    const int qb0 = (int)((long)split * nblk_all / p.nsplit), nblk = (int)((long)(split + 1) * nblk_all / p.nsplit);   // query blocks [qb0, nblk)

    bf16x8 kf[4], vf[4];                                  // B operands: K^T, V^T of this wave's 32 keys
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        kf[s] = zero8(); vf[s] = zero8();
        if (kok) {
            kf[s] = *reinterpret_cast<const bf16x8*>(qb + (long)key * rs + p.H * DH + 16 * s + 8 * h);
            vf[s] = *reinterpret_cast<const bf16x8*>(qb + (long)key * rs + 2 * p.H * DH + 16 * s + 8 * h);
        }
    }
    const unsigned bhrow = (unsigned)(b * p.H + hh) * (unsigned)N;
    const int khalf = key & 1;
    // dropout: the pair half of every word of this lane's key, once; the row halves of a block's 64 queries come from LDS
    const unsigned ck = DROP ? pair_word((unsigned)key >> 1, ds0, ds1) : 0u;
    const short ths = (short)((int)p.thr - 32768);
    const int thr_hi = ((int)p.thr - 32768) * 65536;
    // lanes 2i, 2i + 1 (keys 2 kp, 2 kp + 1, the same queries) share a word per query: each forms the word of every other row (2 jp + khalf)
    // and takes the neighbour's through a quad permute; v_perm_b32 then puts THIS key's field of row 2 jp into the high half of one
    // register and that of row 2 jp + 1 into the low half (selector bytes: 0-3 = the neighbour's word, 4-7 = the own one)
    const unsigned psel = khalf ? 0x03020706u : 0x05040100u;

    f32x16 dk[2] = {zero16(), zero16()}, dv[2] = {zero16(), zero16()};       // dK^T[d][key], dV^T[d][key]
    u32x4 qr[512 / NT], dr[512 / NT];
    float lr = 0.f, tr = 0.f;                             // thread tid < 64: lse / delta of query q0 + tid
    unsigned rw = 0u;                                     // ... and the row half of its dropout words
    auto fetch_rows = [&](int q0) {
        fetch_tile<NT>(qr, qb, rs, q0, N, tid);
        fetch_tile<NT>(dr, dob, os, q0, N, tid);
        if (tid < BLK) {
            const bool ok = q0 + tid < N;
            lr = ok ? lseb[q0 + tid] : INFINITY;           // exp2(. - inf) = 0: queries past N contribute nothing
            tr = ok ? dltb[q0 + tid] : 0.f;
            if (DROP) rw = row_word(bhrow + (unsigned)(q0 + tid), ds0, ds1);
        }
    };
    auto commit_rows = [&](char* buf) {
        commit_tile<NT>(qr, buf, RS, tid);
        commit_tile<NT>(dr, buf + IMG, RS, tid);
        if (tid < BLK) {
            reinterpret_cast<float*>(buf + 2 * IMG)[tid] = lr;
            reinterpret_cast<float*>(buf + 2 * IMG + BLK * 4)[tid] = tr;
            if (DROP) reinterpret_cast<unsigned*>(buf + 2 * IMG + 2 * BLK * 4)[tid] = rw;
        }
    };
    fetch_rows(qb0 * BLK);
    commit_rows(lds + (qb0 & 1) * BUF);
    __syncthreads();

    for (int qbk = qb0; qbk < nblk; ++qbk) {
        const char* qimg = lds + (qbk & 1) * BUF;
        const char* dimg = qimg + IMG;
        const float* lsev = reinterpret_cast<const float*>(qimg + 2 * IMG);
        const float* dltv = lsev + BLK;
        const unsigned* rwv = reinterpret_cast<const unsigned*>(dltv + BLK) + khalf;
        if (qbk + 1 < nblk) fetch_rows((qbk + 1) * BLK);
        const int q0 = qbk * BLK;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            // ---- S[q][key] = Q . K^T, dP[q][key] = dO . V^T for 32 queries (rows) x this lane's key
            f32x16 sa = zero16(), da = zero16();
            {
                bf16x8 qfr[4], dfr[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    qfr[s] = row_frag(qimg, RS, qt * 32, s, lane);
                    dfr[s] = row_frag(dimg, RS, qt * 32, s, lane);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    sa = MFMA32(qfr[s], kf[s], sa);
                    da = MFMA32(dfr[s], vf[s], da);
                }
            }
            bf16x8 tq[2][2], td[2][2];                       // transposed fragments of the two gradient products, requested early
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    td[s][dt] = tr_frag(dimg, RS, qt * 32 + 16 * s, dt * 32, lane);
                    tq[s][dt] = tr_frag(qimg, RS, qt * 32 + 16 * s, dt * 32, lane);
                }
            // row constants of the 16 queries this lane holds: rows 8 g4 + 4 h + j
            f32x16 pd;                                        // dropped probabilities (B operand of dV^T)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(lsev + qt * 32 + 8 * g4 + 4 * h);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(dltv + qt * 32 + 8 * g4 + 4 * h);
                float t4[4] = {1.f, 1.f, 1.f, 1.f};         // keep ? 1 / (1 - p) : 0
                if (DROP) {
#pragma unroll
                    for (int jp = 0; jp < 2; ++jp) {
                        const unsigned u = drop_mix(rwv[qt * 32 + 8 * g4 + 4 * h + 2 * jp] + ck);        // row 2 jp + khalf of the group
                        const unsigned recv = (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0xB1, 0xf, 0xf, false);      // quad_perm [1,0,3,2]
                        const unsigned w = __builtin_amdgcn_perm(u, recv, psel);
                        t4[2 * jp] = (int)w >= thr_hi ? p.inv_keep : 0.f;
                        t4[2 * jp + 1] = (short)(w & 0xffffu) >= ths ? p.inv_keep : 0.f;
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = 4 * g4 + j;
                    // (a lane whose key lies past N computes on a zero K row: its column of dK^T / dV^T is its own and is never stored)
                    const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(sa[r], p.c, -l4[j]));
                    pd[r] = pr * t4[j];
                    sa[r] = pr * __builtin_fmaf(da[r], t4[j], -d4[j]);               // dS = P o (dP_dropped - delta)
                }
            }
            // ---- dV^T += dO^T . P_dropped,  dK^T += Q^T . dS   (sum over the 32 queries = rows of the accumulators)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 pf = pack8(pd, s), sf = pack8(sa, s);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dv[dt] = MFMA32(td[s][dt], pf, dv[dt]);
                    dk[dt] = MFMA32(tq[s][dt], sf, dk[dt]);
                }
            }
        }
        if (qbk + 1 < nblk) commit_rows(lds + ((qbk + 1) & 1) * BUF);
        __syncthreads();
    }

    if (p.nsplit > 1) {
        // partial dK^T / dV^T of this query range, unscaled f32: po = [dK | dV], each [split][b][h][key][d]
        const long row = (((long)split * p.B + b) * p.H + hh) * N + key;
        const long half_ = (long)p.nsplit * p.B * p.H * N * DH;
        if (kok) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int d = dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    p.po[row * DH + d] = dk[dt][r];
                    p.po[half_ + row * DH + d] = dv[dt][r];
                }
        }
        return;
    }
    // ---- epilogue: dK = scale * dK^T^T, dV through LDS, whole rows out
    char* obuf = lds + wv * (2 * 32 * RS);
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        const float f = which == 0 ? p.scale : 1.f;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                union { bf16_t hv[4]; uint2 u; } t;
#pragma unroll
                for (int j = 0; j < 4; ++j) t.hv[j] = (bf16_t)((which == 0 ? dk[dt][4 * g4 + j] : dv[dt][4 * g4 + j]) * f);
                *reinterpret_cast<uint2*>(obuf + which * 32 * RS + (lane & 31) * RS + (dt * 32 + 8 * g4 + 4 * h) * 2) = t.u;
            }
    }
    __syncthreads();
    bf16_t* dst = p.dqkv + (long)b * N * rs + hh * DH;
#pragma unroll
    for (int which = 0; which < 2; ++which)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ci = i * 64 + lane, row = ci >> 3, c8 = ci & 7;
            if (k0 + row < N)
                *reinterpret_cast<u32x4*>(dst + (long)(k0 + row) * rs + (which + 1) * p.H * DH + c8 * 8) =
                    *reinterpret_cast<const u32x4*>(obuf + which * 32 * RS + row * RS + c8 * 16);
        }
}

// delta[b][h][q] = sum_d dO[b][q][h][d] * O[b][q][h][d]: 8 lanes per (b, q, h) row of 64
__global__ __launch_bounds__(256) void flash_delta_kernel(const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout, float* __restrict__ delta,
                                                          int B, int N, int H) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long row = t >> 3;                      // (b * N + q) * H + h
    const int c8 = (int)(t & 7);
    float s = 0.f;
    if (row < (long)B * N * H) {
        float a[8], g[8];
        VecIO<bf16_t, 8>::load(o + row * DH + c8 * 8, a);
        VecIO<bf16_t, 8>::load(dout + row * DH + c8 * 8, g);
#pragma unroll
        for (int j = 0; j < 8; ++j) s = __builtin_fmaf(a[j], g[j], s);
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (c8 == 0 && row < (long)B * N * H) {
        const long bq = row / H;
        const int h = (int)(row % H);
        const long b = bq / N, q = bq % N;
        delta[(b * H + h) * N + q] = s;
    }
}

// merge of the forward partials: out[b][q][h][:] = sum_s 2^(m_s - M) O_s / sum_s 2^(m_s - M) l_s * inv_keep, lse = M + log2(L).
// one thread per (b, h, q, 8 consecutive d); splits in index order (fixed).
__global__ __launch_bounds__(256) void flash_combine_fwd_kernel(FlashP p) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long row = t >> 3;                      // (b * H + h) * N + q
    const int c8 = (int)(t & 7);
    const long rows = (long)p.B * p.H * p.N;
    if (row >= rows) return;
    float M = -INFINITY;
    for (int s = 0; s < p.nsplit; ++s) M = fmaxf(M, p.pm[s * rows + row]);
    float L = 0.f, acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int s = 0; s < p.nsplit; ++s) {
        const float w = __builtin_amdgcn_exp2f(p.pm[s * rows + row] - M);
        L = __builtin_fmaf(w, p.pl[s * rows + row], L);
        const f32x4 a = *reinterpret_cast<const f32x4*>(p.po + (s * rows + row) * DH + c8 * 8), c = *reinterpret_cast<const f32x4*>(p.po + (s * rows + row) * DH + c8 * 8 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[j] = __builtin_fmaf(w, a[j], acc[j]); acc[4 + j] = __builtin_fmaf(w, c[j], acc[4 + j]); }
    }
    const float f = p.inv_keep / L;
    const int q = (int)(row % p.N);
    const long bh = row / p.N;
    const int hh = (int)(bh % p.H);
    const long b = bh / p.H;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] *= f;
    VecIO<bf16_t, 8>::store(p.out + ((b * p.N + q) * p.H + hh) * DH + c8 * 8, acc);
    if (c8 == 0) p.lse[row] = M + __builtin_amdgcn_logf(L);
}

// merge of the backward partials: slot 0 (dQ) from pq [nsplit][B][H][N][64] * scale, slots 1, 2 (dK * scale, dV) from pk = [dK | dV]
__global__ __launch_bounds__(256) void flash_combine_bwd_kernel(FlashP p, const float* pq, const float* pk) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long row = t >> 3;
    const int c8 = (int)(t & 7);
    const long rows = (long)p.B * p.H * p.N;
    if (row >= rows) return;
    const int q = (int)(row % p.N);
    const long bh = row / p.N;
    const int hh = (int)(bh % p.H);
    const long b = bh / p.H;
    const long half_ = (long)p.nsplit * rows * DH;
    bf16_t* dst = p.dqkv + (b * p.N + q) * (3L * p.H * DH) + hh * DH + c8 * 8;
#pragma unroll
    for (int which = 0; which < 3; ++which) {
        const float* src = which == 0 ? pq : (which == 1 ? pk : pk + half_);
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int s = 0; s < p.nsplit; ++s) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(src + (s * rows + row) * DH + c8 * 8), c = *reinterpret_cast<const f32x4*>(src + (s * rows + row) * DH + c8 * 8 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[j] += a[j]; acc[4 + j] += c[j]; }
        }
        const float f = which == 2 ? 1.f : p.scale;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] *= f;
        VecIO<bf16_t, 8>::store(dst + (long)which * p.H * DH, acc);
    }
}

// split-K factor (measured, tools/attn_split_sweep.sh: B=1 N=2048 0.276 -> 0.165 ms fwd+bwd at 4, B=2 0.308 -> 0.277 at 2,
// B=5 0.723 -> 0.662 at 2, B=4 N=512 0.087 -> 0.081 at 4; B=20 N=512 loses): double while the launch has fewer than two waves
// per SIMD and a split keeps >= 2 blocks, once more below 4 per SIMD when a split keeps >= 16 blocks
int pick_split(int B, int N, int H) {
    const long tiles = (long)((N + 31) / 32) * H * B;
    const int nblk = (N + BLK - 1) / BLK;
    int s = 1;
    if (g_flash_split > 0) s = g_flash_split;
    else {
        while (tiles * s < 2048 && nblk / (2 * s) >= 2) s *= 2;
        if (tiles * s < 4096 && nblk / (2 * s) >= 16) s *= 2;
    }
    while (s > 1 && s > nblk) --s;
    return s;
}

int set_dropout(FlashP& p, float drop_p, uint64_t seed) {
    if (!(drop_p >= 0.f && drop_p < 1.f)) return -1003;
    p.thr = (unsigned)(drop_p * 65536.0f + 0.5f);
    p.inv_keep = p.thr ? 1.0f / (1.0f - (float)p.thr / 65536.0f) : 1.0f;
    p.s0 = (unsigned)(seed & 0xffffffffu);
    p.s1 = (unsigned)(seed >> 32);
    p.soff = g_passion_step_words;
    return 0;
}

template <typename K>
int grant_lds(K kern, int bytes) {
    HIP_RET(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    return 0;
}

// waves per workgroup: 4 when that still gives the chip >= 2 workgroups per CU, else fewer
int pick_nw(int B, int N, int H, int nsplit) {
    const long tiles = (long)((N + 31) / 32) * H * B * nsplit;
    return tiles >= 4 * 512 ? 4 : (tiles >= 2 * 256 ? 2 : 1);
}

}  // namespace

extern "C" int passion_flash_attn_supported(int dtype, int Dh) { return dtype == PASSION_BF16 && Dh == DH; }

extern "C" int passion_flash_attn_fwd(const void* qkv, int B, int N, int H, int Dh, float scale, float drop_p, uint64_t seed,
                                      void* out, float* lse, void* stream) {
    if (Dh != DH || B <= 0 || N <= 0 || H <= 0) return -1003;
    if ((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(out)) & 15) return -1005;
    FlashP p{};
    p.qkv = (const bf16_t*)qkv; p.out = (bf16_t*)out; p.lse = lse;
    p.B = B; p.N = N; p.H = H; p.scale = scale; p.c = scale * 1.4426950408889634f;
    int rc = set_dropout(p, drop_p, seed);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int nsplit = pick_split(B, N, H);
    const int nw = pick_nw(B, N, H, nsplit);
    const dim3 grid((N + 32 * nw - 1) / (32 * nw) * nsplit, H, B);
    const int bytes = 2 * (BLK * RS + BLK * TS + 128);
    const size_t rows = (size_t)B * H * N;
    p.nsplit = nsplit;
    if (nsplit > 1) {
        void* plain = nullptr;
        if ((rc = pconv::stream_scratch(st, 0, nsplit * rows * (DH + 2) * sizeof(float), nullptr, &plain))) return rc;
        p.po = (float*)plain; p.pm = p.po + nsplit * rows * DH; p.pl = p.pm + nsplit * rows;
    }
    ProfScope prof(4, 2.0 * B * N * H * Dh * 2 * 2, 4.0 * B * H * (double)N * N * Dh, st);
    passion_prof_tag("flash_q_kernel<%d, 0>", nw);
    static bool granted = false;
    if (!granted) {
        if ((rc = grant_lds(flash_q_kernel<4, 0, false>, bytes)) || (rc = grant_lds(flash_q_kernel<2, 0, false>, bytes)) || (rc = grant_lds(flash_q_kernel<1, 0, false>, bytes)) ||
            (rc = grant_lds(flash_q_kernel<4, 0, true>, bytes)) || (rc = grant_lds(flash_q_kernel<2, 0, true>, bytes)) || (rc = grant_lds(flash_q_kernel<1, 0, true>, bytes))) return rc;
        granted = true;
    }
    if (p.thr) {
        if (nw == 4) hipLaunchKernelGGL((flash_q_kernel<4, 0, true>), grid, dim3(256), bytes, st, p);
        else if (nw == 2) hipLaunchKernelGGL((flash_q_kernel<2, 0, true>), grid, dim3(128), bytes, st, p);
        else hipLaunchKernelGGL((flash_q_kernel<1, 0, true>), grid, dim3(64), bytes, st, p);
    } else {
        if (nw == 4) hipLaunchKernelGGL((flash_q_kernel<4, 0, false>), grid, dim3(256), bytes, st, p);
        else if (nw == 2) hipLaunchKernelGGL((flash_q_kernel<2, 0, false>), grid, dim3(128), bytes, st, p);
        else hipLaunchKernelGGL((flash_q_kernel<1, 0, false>), grid, dim3(64), bytes, st, p);
    }
    LAUNCH_RET();
    if (nsplit > 1) {
        hipLaunchKernelGGL(flash_combine_fwd_kernel, dim3((unsigned)((rows * 8 + 255) / 256)), dim3(256), 0, st, p);
        LAUNCH_RET();
    }
    return 0;
}

extern "C" int passion_flash_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, int B, int N, int H, int Dh,
                                      float scale, float drop_p, uint64_t seed, float* delta, void* dqkv, void* stream) {
    if (Dh != DH || B <= 0 || N <= 0 || H <= 0) return -1003;
    if ((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(dqkv)) & 15) return -1005;
    FlashP p{};
    p.qkv = (const bf16_t*)qkv; p.out = nullptr; p.lse = const_cast<float*>(lse); p.dout = (const bf16_t*)dout; p.delta = delta; p.dqkv = (bf16_t*)dqkv;
    p.B = B; p.N = N; p.H = H; p.scale = scale; p.c = scale * 1.4426950408889634f;
    int rc = set_dropout(p, drop_p, seed);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(5, 2.0 * B * N * H * Dh * 2 * 4, 10.0 * B * H * (double)N * N * Dh, st);      // backward = 2.5 x forward FLOPs
    const long rows8 = (long)B * N * H * 8;
    hipLaunchKernelGGL(flash_delta_kernel, dim3((unsigned)((rows8 + 255) / 256)), dim3(256), 0, st, (const bf16_t*)out, (const bf16_t*)dout, delta, B, N, H);
    LAUNCH_RET();
    const int nsplit = pick_split(B, N, H);
    const int nw = pick_nw(B, N, H, nsplit);
    const dim3 grid((N + 32 * nw - 1) / (32 * nw) * nsplit, H, B);
    const int qbytes = 2 * (2 * BLK * RS + 128);
    const size_t rows = (size_t)B * H * N;
    p.nsplit = nsplit;
    float *pq = nullptr, *pk = nullptr;
    if (nsplit > 1) {
        void* plain = nullptr;
        if ((rc = pconv::stream_scratch(st, 0, 3 * nsplit * rows * DH * sizeof(float), nullptr, &plain))) return rc;
        pq = (float*)plain; pk = pq + nsplit * rows * DH;
    }
    FlashP pkk = p;
    p.po = pq; pkk.po = pk;
    const int kbytes_min = 2 * (2 * BLK * RS + 3 * BLK * 4);
    passion_prof_tag("flash_q_kernel<%d, 1> + flash_k_kernel<%d>", nw, nw);
    static bool granted = false;
    if (!granted) {
        const int kb4 = kbytes_min > 4 * 2 * 32 * RS ? kbytes_min : 4 * 2 * 32 * RS;
        if ((rc = grant_lds(flash_q_kernel<4, 1, false>, qbytes)) || (rc = grant_lds(flash_q_kernel<2, 1, false>, qbytes)) || (rc = grant_lds(flash_q_kernel<1, 1, false>, qbytes)) ||
            (rc = grant_lds(flash_q_kernel<4, 1, true>, qbytes)) || (rc = grant_lds(flash_q_kernel<2, 1, true>, qbytes)) || (rc = grant_lds(flash_q_kernel<1, 1, true>, qbytes)) ||
            (rc = grant_lds(flash_k_kernel<4, false>, kb4)) || (rc = grant_lds(flash_k_kernel<2, false>, kb4)) || (rc = grant_lds(flash_k_kernel<1, false>, kb4)) ||
            (rc = grant_lds(flash_k_kernel<4, true>, kb4)) || (rc = grant_lds(flash_k_kernel<2, true>, kb4)) || (rc = grant_lds(flash_k_kernel<1, true>, kb4))) return rc;
        granted = true;
    }
    const int kbytes = kbytes_min > nw * 2 * 32 * RS ? kbytes_min : nw * 2 * 32 * RS;      // the epilogue stages 2 x [32][64] per wave
#define FLASH_BWD(NWV, DR) do { hipLaunchKernelGGL((flash_q_kernel<NWV, 1, DR>), grid, dim3(64 * NWV), qbytes, st, p); \
                                hipLaunchKernelGGL((flash_k_kernel<NWV, DR>), grid, dim3(64 * NWV), kbytes, st, pkk); } while (0)
    if (p.thr) {
        if (nw == 4) FLASH_BWD(4, true); else if (nw == 2) FLASH_BWD(2, true); else FLASH_BWD(1, true);
    } else {
        if (nw == 4) FLASH_BWD(4, false); else if (nw == 2) FLASH_BWD(2, false); else FLASH_BWD(1, false);
    }
#undef FLASH_BWD
    LAUNCH_RET();
    if (nsplit > 1) {
        hipLaunchKernelGGL(flash_combine_bwd_kernel, dim3((unsigned)((rows * 8 + 255) / 256)), dim3(256), 0, st, p, (const float*)pq, (const float*)pk);
        LAUNCH_RET();
    }
    return 0;
}
