// Nested stream forks under hipStreamBeginCapture: origin O -> side E -> companion C (ROCm 7.2 crash hunt, see conv.hip Companion)
//   hipcc --offload-arch=gfx950 tools/capture_nest_probe.hip -o tools/capture_nest_probe && tools/capture_nest_probe VARIANT
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t err__ = (x); if (err__ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(err__)); return 1; } } while (0)
__global__ void k(float* p) { p[threadIdx.x] += 1.f; }
int main(int argc, char** argv) {
    const int variant = argc > 1 ? atoi(argv[1]) : 0;
    hipStream_t O, E, C;
    CK(hipStreamCreateWithFlags(&O, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&E, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&C, hipStreamNonBlocking));
    hipEvent_t e1, e2, e3, e4, f, d;
    for (hipEvent_t* e : {&e1, &e2, &e3, &e4, &f, &d}) CK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    float* buf; CK(hipMalloc(&buf, 4096)); CK(hipMemset(buf, 0, 4096));
    hipGraph_t g;
    CK(hipStreamBeginCapture(O, variant & 8 ? hipStreamCaptureModeThreadLocal : hipStreamCaptureModeGlobal));
    if (variant & 1) { CK(hipEventRecord(f, O)); CK(hipStreamWaitEvent(C, f, 0)); }          // companion joins from the origin first
    CK(hipEventRecord(e1, O)); CK(hipStreamWaitEvent(E, e1, 0));
    hipLaunchKernelGGL(k, 1, 64, 0, E, buf);
    const int reps = variant & 4 ? 3 : 1;
    for (int r = 0; r < reps; ++r) {                                                        // the nested fork, `reps` times with the same events
        CK(hipEventRecord(e2, E)); CK(hipStreamWaitEvent(C, e2, 0));
        hipLaunchKernelGGL(k, 1, 64, 0, C, buf + 64);
        hipLaunchKernelGGL(k, 1, 64, 0, E, buf + 128);
        if (variant & 16) CK(hipMemsetAsync(buf + 256, 0, 256, E));                         // a memset node on the forked stream
        if (variant & 32) CK(hipMemsetAsync(buf + 320, 0, 256, C));
        CK(hipEventRecord(e3, C)); CK(hipStreamWaitEvent(E, e3, 0));
    }
    hipLaunchKernelGGL(k, 1, 64, 0, E, buf);
    CK(hipEventRecord(e4, E)); CK(hipStreamWaitEvent(O, e4, 0));
    if (variant & 2) { CK(hipEventRecord(d, C)); CK(hipStreamWaitEvent(O, d, 0)); }          // ... and is joined back to it last
    printf("ending capture (variant %d)\n", variant); fflush(stdout);
    CK(hipStreamEndCapture(O, &g));
    hipGraphExec_t ge;
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, O)); CK(hipStreamSynchronize(O));
    float h[192]; CK(hipMemcpy(h, buf, sizeof(h), hipMemcpyDeviceToHost));
    printf("ok: %g %g %g\n", h[0], h[64], h[128]);
    return 0;
}
