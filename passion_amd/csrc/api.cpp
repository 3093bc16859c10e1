// Library-level entry points: ABI version and the optional hipEvent timing of conv launches.
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>
#include "passion_hip.h"
#include "prof.h"

int g_passion_prof_on = 0;
double g_passion_prof_lo = 0.0, g_passion_prof_hi = 1e300;

namespace {
struct Pending { int kind; double bytes, flops; hipEvent_t e0, e1; bool closed; char name[64]; };
thread_local char g_tag[64] = "";
std::mutex g_mu;
std::vector<Pending> g_pending;
std::vector<hipEvent_t> g_free_events;        // recycled: creating / destroying two events per timed launch cost ~5 % of a step
hipEvent_t take_event() {
    if (!g_free_events.empty()) { hipEvent_t e = g_free_events.back(); g_free_events.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
}  // namespace

void passion_prof_begin(int kind, double bytes, double flops, hipStream_t st) {
    Pending p{kind, bytes, flops, nullptr, nullptr, false, ""};
    g_tag[0] = 0;
    std::lock_guard<std::mutex> lk(g_mu);
    p.e0 = take_event(); p.e1 = take_event();
    if (!p.e0 || !p.e1) return;
    (void)hipEventRecord(p.e0, st);
    g_pending.push_back(p);
}

void passion_prof_end(hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (size_t i = g_pending.size(); i-- > 0;) {
        if (!g_pending[i].closed) {
            (void)hipEventRecord(g_pending[i].e1, st);
            g_pending[i].closed = true;
            snprintf(g_pending[i].name, sizeof(g_pending[i].name), "%s", g_tag);
            break;
        }
    }
}

const char* passion_prof_last_tag() { return g_tag; }

void passion_prof_tag(const char* fmt, ...) {
    if (!g_passion_prof_on) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_tag, sizeof(g_tag), fmt, ap);
    va_end(ap);
}

extern "C" int passion_abi_version(void) { return 2; }      // 2: passion_seg_t.dx_add, dx_add2 (round 6)

// ---- step words: what changes from one replay of a captured training step to the next lives in device memory
const unsigned* g_passion_step_words = nullptr;
extern "C" void passion_set_step_words(const void* dev_words) { g_passion_step_words = static_cast<const unsigned*>(dev_words); }
namespace {
struct Words16 { uint32_t w[16]; };
__global__ void write_words_kernel(uint32_t* dst, int n, Words16 v) {
    if ((int)threadIdx.x < n) dst[threadIdx.x] = v.w[threadIdx.x];
}
}  // namespace
// device-to-device streaming copy, 16 bytes per lane (bench.py: the measured HBM peak beside the nominal one)
namespace {
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
// one 16-byte element per thread up to 2^20 blocks, non-temporal both ways (measured with tools/copy_probe.hip on 1 GiB: 2048 blocks
// 5.1 TB/s, 65536 blocks 5.9; unrolling the loop body loses)
__global__ __launch_bounds__(256) void stream_copy_kernel(const u32x4_t* __restrict__ src, u32x4_t* __restrict__ dst, long n16) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256)
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}
}  // namespace
extern "C" int passion_stream_copy(void* dst, const void* src, int64_t bytes, void* stream) {
    if (bytes < 0 || (bytes & 15) || ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15)) return -1005;
    const long n16 = bytes / 16;
    if (n16 == 0) return 0;
    const long blocks = (n16 + 255) / 256 < (1L << 20) ? (n16 + 255) / 256 : (1L << 20);
    hipLaunchKernelGGL(stream_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, static_cast<const u32x4_t*>(src),
                       static_cast<u32x4_t*>(dst), n16);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
extern "C" int passion_write_words(void* dst, int n, const uint32_t* words, void* stream) {
    if (n < 0 || n > 16 || !dst) return -1003;
    Words16 v;
    for (int i = 0; i < 16; ++i) v.w[i] = i < n ? words[i] : 0u;
    hipLaunchKernelGGL(write_words_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, static_cast<uint32_t*>(dst), n, v);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---- per-stream scratch (conv_common.h): tickets (kept zero by the kernels that use them) + uninitialised slots
#include "conv_common.h"
namespace {
struct StreamScratch { hipStream_t st; int dev; char *zero, *plain; size_t zero_sz, plain_sz; };
std::vector<StreamScratch> g_scratch;
std::mutex g_scratch_mu;
}  // namespace
namespace pconv {
int stream_scratch(hipStream_t st, size_t zero_bytes, size_t plain_bytes, void** zero, void** plain) {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    int dev = 0;
    HIP_RET(hipGetDevice(&dev));
    StreamScratch* s = nullptr;
    for (auto& e : g_scratch) if (e.st == st && e.dev == dev) { s = &e; break; }
    if (!s) {
        g_scratch.push_back(StreamScratch{st, dev, nullptr, nullptr, 0, 0});
        s = &g_scratch.back();
    }
    if (zero_bytes > s->zero_sz) {
        const size_t want = zero_bytes < (64u << 10) ? (64u << 10) : zero_bytes * 2;
        if (s->zero) { HIP_RET(hipStreamSynchronize(st)); HIP_RET(hipFree(s->zero)); s->zero = nullptr; s->zero_sz = 0; }
        HIP_RET(hipMalloc(reinterpret_cast<void**>(&s->zero), want));
        HIP_RET(hipMemsetAsync(s->zero, 0, want, st));      // ordered before the first kernel that uses it
        s->zero_sz = want;
    }
    if (plain_bytes > s->plain_sz) {
        const size_t want = plain_bytes < (4u << 20) ? (4u << 20) : plain_bytes * 2;
        if (s->plain) { HIP_RET(hipStreamSynchronize(st)); HIP_RET(hipFree(s->plain)); s->plain = nullptr; s->plain_sz = 0; }
        HIP_RET(hipMalloc(reinterpret_cast<void**>(&s->plain), want));
        s->plain_sz = want;
    }
    if (zero) *zero = s->zero;
    if (plain) *plain = s->plain;
    return 0;
}

int fwd_stats_setup_raw(FwdStats& f, int B, int C, int nwg, int ny, double inv_v, float eps, float* stats, hipStream_t st) {
    f.nwg = nwg; f.ny = ny; f.ngroup = (nwg + STATS_GROUP - 1) / STATS_GROUP;
    const size_t nv = (size_t)C * 2;
    const size_t slots = (size_t)B * nwg * nv * sizeof(float), gsl = (size_t)B * f.ngroup * nv * sizeof(double);
    const size_t gs_off = (slots + 255) / 256 * 256;
    void *zero = nullptr, *plain = nullptr;
    const int rc = stream_scratch(st, (size_t)B * (f.ngroup + 1) * sizeof(unsigned), gs_off + gsl, &zero, &plain);
    if (rc) return rc;
    f.tickets = static_cast<unsigned*>(zero);
    f.slots = static_cast<float*>(plain);
    f.gslots = reinterpret_cast<double*>(static_cast<char*>(plain) + gs_off);
    f.direct = ((long)nwg * (long)nv <= 16384 && nv <= 256) ? 1 : 0;
    f.eps = eps;
    f.inv_v = inv_v;
    f.stats = stats;
    return 0;
}

int fwd_stats_setup(ConvK& k, int nwg, int ny, hipStream_t st) {
    return fwd_stats_setup_raw(k.fst, k.B, k.Cout, nwg, ny, 1.0 / ((double)k.Do * k.Ho * k.Wo), k.fst_eps, k.fst_want, st);
}
}  // namespace pconv

// run-time A/B switches (the same ones the environment sets at load time); returns the previous value, -1 = unknown name
extern int g_passion_force_f32_mfma, g_flash_split;
namespace pconv { extern int g_conv_s2f, g_wgrad_s2m, g_pack_f32; }
namespace pconv { extern int g_conv_merge, g_wgrad_wide, g_conv_dbg, g_conv_target, g_conv_lean, g_conv_padout, g_wgrad_fork, g_conv_blockdiag, g_conv_march, g_wgrad_march, g_conv_c1, g_conv_s2, g_conv_pw, g_conv_pw8, g_conv_pwfinal, g_conv_deep, g_conv_marchfinal, g_conv_pair, g_deep_pair, g_wgrad_pw, g_wgrad_pw_minvox, g_wgrad_group_s2, g_c1_mfma; }
extern "C" int passion_set_option(const char* name, int value) {
    struct { const char* n; int* p; } opts[] = {
        {"force_f32_mfma", &g_passion_force_f32_mfma}, {"conv_lean", &pconv::g_conv_lean}, {"conv_padout", &pconv::g_conv_padout},
        {"conv_target", &pconv::g_conv_target}, {"conv_merge", &pconv::g_conv_merge}, {"wgrad_wide", &pconv::g_wgrad_wide},
        {"conv_dbg", &pconv::g_conv_dbg}, {"wgrad_fork", &pconv::g_wgrad_fork}, {"conv_march", &pconv::g_conv_march}, {"wgrad_march", &pconv::g_wgrad_march}, {"conv_c1", &pconv::g_conv_c1}, {"conv_s2", &pconv::g_conv_s2}, {"conv_pw", &pconv::g_conv_pw}, {"conv_pw8", &pconv::g_conv_pw8}, {"conv_pwfinal", &pconv::g_conv_pwfinal}, {"flash_split", &g_flash_split}, {"conv_deep", &pconv::g_conv_deep}, {"deep_pair", &pconv::g_deep_pair}, {"conv_marchfinal", &pconv::g_conv_marchfinal}, {"conv_pair", &pconv::g_conv_pair}, {"wgrad_pw", &pconv::g_wgrad_pw}, {"wgrad_pw_minvox", &pconv::g_wgrad_pw_minvox}, {"wgrad_group_s2", &pconv::g_wgrad_group_s2}, {"c1_mfma", &pconv::g_c1_mfma}, {"conv_s2f", &pconv::g_conv_s2f}, {"pack_f32", &pconv::g_pack_f32}, {"wgrad_s2m", &pconv::g_wgrad_s2m}};
    for (auto& o : opts) {
        const char *a = o.n, *b = name;
        while (*a && *a == *b) { ++a; ++b; }
        if (!*a && !*b) { const int old = *o.p; *o.p = value; return old; }
    }
    return -1;
}

extern "C" void passion_prof_enable(int on) { g_passion_prof_on = on; }
// only launches whose algorithmic bytes lie in [lo, hi] keep their event pair (bench.py: the launches of ONE kernel family inside the
// timed region -- every pair costs the step ~10 us); (0, 1e300) = all
extern "C" void passion_prof_filter(double lo, double hi) { g_passion_prof_lo = lo; g_passion_prof_hi = hi; }

// Drains finished records (synchronises on their events: call outside the timed region).
extern "C" int passion_prof_read(passion_prof_rec_t* rec, int cap) {
    std::lock_guard<std::mutex> lk(g_mu);
    int n = 0;
    size_t i = 0;
    for (; i < g_pending.size() && n < cap; ++i) {
        Pending& p = g_pending[i];
        if (!p.closed) break;
        float ms = 0.f;
        (void)hipEventSynchronize(p.e1);
        (void)hipEventElapsedTime(&ms, p.e0, p.e1);
        g_free_events.push_back(p.e0); g_free_events.push_back(p.e1);
        rec[n].kind = p.kind; rec[n].ms = ms; rec[n].bytes = p.bytes; rec[n].flops = p.flops;
        snprintf(rec[n].name, sizeof(rec[n].name), "%s", p.name);
        ++n;
    }
    g_pending.erase(g_pending.begin(), g_pending.begin() + i);
    return n;
}

// The same without consuming the records: event pairs that were recorded while a stream capture was running are nodes
// of that HIP graph and are re-recorded by every replay, so they must stay alive; ms is the elapsed time of the latest
// replay (or -1 when the runtime cannot time them).
extern "C" int passion_prof_peek(passion_prof_rec_t* rec, int cap) {
    std::lock_guard<std::mutex> lk(g_mu);
    int n = 0;
    for (size_t i = 0; i < g_pending.size() && n < cap; ++i) {
        Pending& p = g_pending[i];
        if (!p.closed) break;
        float ms = -1.f;
        if (hipEventSynchronize(p.e1) != hipSuccess || hipEventElapsedTime(&ms, p.e0, p.e1) != hipSuccess) ms = -1.f;
        rec[n].kind = p.kind; rec[n].ms = ms; rec[n].bytes = p.bytes; rec[n].flops = p.flops;
        snprintf(rec[n].name, sizeof(rec[n].name), "%s", p.name);
        ++n;
    }
    (void)hipGetLastError();
    return n;
}

// A/B switch for debugging: PASSION_FORCE_F32_MFMA=1 routes bf16 tensors through the f32-MFMA conv kernels.
extern int g_passion_force_f32_mfma, g_flash_split;
namespace pconv { extern int g_conv_merge, g_wgrad_wide, g_conv_dbg, g_conv_target, g_conv_lean, g_conv_padout, g_wgrad_fork, g_conv_blockdiag, g_conv_march, g_wgrad_march, g_conv_c1, g_conv_s2, g_conv_pw, g_conv_pw8, g_conv_pwfinal, g_conv_deep, g_conv_marchfinal, g_conv_pair, g_deep_pair, g_wgrad_pw, g_wgrad_pw_minvox, g_wgrad_group_s2, g_c1_mfma; }
__attribute__((constructor)) static void passion_read_env() {
    const char* e = getenv("PASSION_FORCE_F32_MFMA");
    if (e && e[0] == '1') g_passion_force_f32_mfma = 1;
    e = getenv("PASSION_CONV_MERGE");
    if (e && e[0] == '1') pconv::g_conv_merge = 1;
    e = getenv("PASSION_CONV_DBG");
    if (e) pconv::g_conv_dbg = atoi(e);
    e = getenv("PASSION_CONV_MARCH");
    if (e && e[0] == '0') pconv::g_conv_march = 0;
    e = getenv("PASSION_CONV_C1");
    if (e && e[0] == '0') pconv::g_conv_c1 = 0;
    e = getenv("PASSION_CONV_S2");
    if (e && e[0] == '0') pconv::g_conv_s2 = 0;
    e = getenv("PASSION_CONV_PW");
    if (e && e[0] == '0') pconv::g_conv_pw = 0;
    e = getenv("PASSION_CONV_PW8");
    if (e && e[0] == '0') pconv::g_conv_pw8 = 0;
    e = getenv("PASSION_CONV_PWFINAL");
    if (e && e[0] == '0') pconv::g_conv_pwfinal = 0;
    e = getenv("PASSION_CONV_MARCHFINAL");
    if (e) pconv::g_conv_marchfinal = e[0] == '1' ? 1 : (e[0] == '2' ? 2 : 0);
    e = getenv("PASSION_CONV_DEEP");
    if (e && e[0] == '0') pconv::g_conv_deep = 0;
    e = getenv("PASSION_WGRAD_PW");
    if (e && e[0] == '0') pconv::g_wgrad_pw = 0;
    e = getenv("PASSION_DEEP_PAIR");
    if (e) pconv::g_deep_pair = e[0] == '1' ? 1 : 0;
    e = getenv("PASSION_CONV_PAIR");
    if (e && e[0] == '0') pconv::g_conv_pair = 0;
    e = getenv("PASSION_FLASH_SPLIT");
    if (e) g_flash_split = atoi(e);
    e = getenv("PASSION_WGRAD_MARCH");
    if (e && e[0] == '0') pconv::g_wgrad_march = 0;
    e = getenv("PASSION_CONV_LEAN");
    if (e && e[0] == '0') pconv::g_conv_lean = 0;
    e = getenv("PASSION_WGRAD_FORK");
    if (e && e[0] >= '0' && e[0] <= '9') pconv::g_wgrad_fork = atoi(e);
    e = getenv("PASSION_CONV_PADOUT");
    if (e) pconv::g_conv_padout = atoi(e);          // 0 never, 1 always, N > 1: only for volumes with depth < N
    e = getenv("PASSION_CONV_BLOCKDIAG");
    if (e && e[0] == '0') pconv::g_conv_blockdiag = 0;
    e = getenv("PASSION_CONV_TARGET");
    if (e && atoi(e) > 0) pconv::g_conv_target = atoi(e);
    e = getenv("PASSION_WGRAD_WIDE");
    if (e && e[0] == '1') pconv::g_wgrad_wide = 1;
    e = getenv("PASSION_C1_MFMA");
    if (e && e[0] == '0') pconv::g_c1_mfma = 0;
    e = getenv("PASSION_WGRAD_GROUP_S2");
    if (e && e[0] == '0') pconv::g_wgrad_group_s2 = 0;
}
