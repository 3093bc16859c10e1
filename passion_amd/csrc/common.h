// Shared device helpers for the PASSION hot-path kernels (gfx950 / CDNA4 only).
// Activations are channels-last [B][D][H][W][C]; storage is f32 or bf16, all
// arithmetic accumulates in f32.  Wavefront = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __bf16 bf16_t;

#define PASSION_F32 0
#define PASSION_BF16 1

#define WAVE 64

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// Load 4 consecutive channel values as f32 (pointer must be 4-element aligned).
__device__ __forceinline__ f32x4 load4(const float* p) {
    return *reinterpret_cast<const f32x4*>(p);
}
__device__ __forceinline__ f32x4 load4(const bf16_t* p) {
    uint2 u = *reinterpret_cast<const uint2*>(p);
    f32x4 r;
    r[0] = __uint_as_float(u.x << 16);
    r[1] = __uint_as_float(u.x & 0xffff0000u);
    r[2] = __uint_as_float(u.y << 16);
    r[3] = __uint_as_float(u.y & 0xffff0000u);
    return r;
}
__device__ __forceinline__ void store4(float* p, f32x4 v) {
    *reinterpret_cast<f32x4*>(p) = v;
}
__device__ __forceinline__ void store4(bf16_t* p, f32x4 v) {
    union { bf16_t h[4]; uint2 u; } t;
    t.h[0] = (bf16_t)v[0]; t.h[1] = (bf16_t)v[1]; t.h[2] = (bf16_t)v[2]; t.h[3] = (bf16_t)v[3];
    *reinterpret_cast<uint2*>(p) = t.u;
}

// VEC consecutive channel values <-> float[VEC].  VEC = 8 on bf16 is one 16-byte access per lane, the width the memory
// system wants from a streaming kernel (pointer must be VEC-element aligned).
template <typename T, int VEC> struct VecIO;
template <typename T> struct VecIO<T, 4> {
    static __device__ __forceinline__ void load(const T* p, float (&v)[4]) {
        const f32x4 t = load4(p);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = t[j];
    }
    static __device__ __forceinline__ void store(T* p, const float (&v)[4]) { store4(p, (f32x4){v[0], v[1], v[2], v[3]}); }
};
template <> struct VecIO<bf16_t, 8> {
    static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
        const uint4 u = *reinterpret_cast<const uint4*>(p);
        const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(w[j] << 16); v[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u); }
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
        union { bf16_t h[8]; uint4 u; } t;
#pragma unroll
        for (int j = 0; j < 8; ++j) t.h[j] = (bf16_t)v[j];
        *reinterpret_cast<uint4*>(p) = t.u;
    }
};

// Sum over the 64 lanes of a wavefront (result valid in every lane).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Sum over aligned groups of 16 lanes.
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Hand-over of a few words between workgroups of one kernel (per-workgroup partial sums -> the workgroup that arrives last):
// RELAXED ATOMIC stores / loads at agent scope (sc1: coherent across the XCDs' L2s) + a ticket counter, ordered by the
// s_waitcnt of __syncthreads().  NOT __threadfence(): an agent-scope release fence writes back every dirty line of the XCD's L2
// (buffer_wbl2) -- at the end of a conv that just stored its output that cost 50 us per launch (128^3, 8 channels).
// ... and, in the workgroup that won the ticket, an ACQUIRE fence before it reads the others' slots (buffer_inv sc1: drops the
// lines its own L2 may still hold from the buffer's previous use -- one stale 128-byte line in ~3000 launches without it;
// an invalidate writes nothing back, so it is cheap where the release fence is not).
__device__ __forceinline__ void acquire_agent() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
// every vector-memory operation of this wave (the sc1 slot stores) acknowledged at its scope before anything that follows (the
// barrier in front of the ticket): spelled out rather than left to what __syncthreads() happens to wait for
__device__ __forceinline__ void stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ int reflect_idx(int p, int n) {
    // torch 'reflect' padding (no edge repeat); valid for |overshoot| < n.
    if (p < 0) p = -p;
    if (p >= n) p = 2 * n - 2 - p;
    return p;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

#define HIP_RET(expr)                                   \
    do {                                                \
        hipError_t e__ = (expr);                        \
        if (e__ != hipSuccess) return -(int)e__;        \
    } while (0)
#define LAUNCH_RET()                                    \
    do {                                                \
        hipError_t e__ = hipGetLastError();             \
        if (e__ != hipSuccess) return -(int)e__;        \
    } while (0)

static inline long hmin(long a, long b) { return a < b ? a : b; }
static inline long hmax(long a, long b) { return a > b ? a : b; }
