// Optional per-launch hipEvent timing of the conv kernels (bench.py's roofline leg).
#pragma once
#include <hip/hip_runtime.h>
void passion_prof_begin(int kind, double bytes, double flops, hipStream_t st);
void passion_prof_end(hipStream_t st);
void passion_prof_tag(const char* fmt, ...);
const char* passion_prof_last_tag();            // the tag of the scope that closed last on this thread   // name of the kernel instantiation launched inside the open scope
extern int g_passion_prof_on;  // bit k set: time launches of kind k
extern double g_passion_prof_lo, g_passion_prof_hi;   // ... whose algorithmic bytes lie in [lo, hi] (passion_prof_filter)
struct ProfScope {
    hipStream_t st; bool on;
    ProfScope(int kind, double bytes, double flops, hipStream_t s)
        : st(s), on(((g_passion_prof_on >> kind) & 1) != 0 && bytes >= g_passion_prof_lo && bytes <= g_passion_prof_hi) {
        if (on) passion_prof_begin(kind, bytes, flops, st);
    }
    ~ProfScope() { if (on) passion_prof_end(st); }
};
