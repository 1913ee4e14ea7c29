// launch-gap probe: a chain of N small dependent kernels, enqueued one by one vs replayed as a captured hipGraph
// hipcc --offload-arch=gfx950 -O2 scripts/graph_gap.hip -o /tmp/graph_gap && /tmp/graph_gap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_step(double* x, int k) {
    // ~2 us of dependent work in one wave
    double v = x[threadIdx.x];
    for (int i = 0; i < 4000; i++) v = v * 1.0000001 + 1e-9 * k;
    x[threadIdx.x] = v;
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const int N = 200;
    double* x;
    hipMalloc(&x, 64 * 8);
    hipMemset(x, 0, 64 * 8);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int rep = 0; rep < 3; rep++) {
        double t0 = now();
        for (int k = 0; k < N; k++) hipLaunchKernelGGL(k_step, dim3(1), dim3(64), 0, s, x, k);
        double t1 = now();
        hipStreamSynchronize(s);
        double t2 = now();
        printf("stream: enqueue %.1f us, total %.1f us = %.2f us per kernel\n", t1 - t0, t2 - t0, (t2 - t0) / N);
    }
    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int k = 0; k < N; k++) hipLaunchKernelGGL(k_step, dim3(1), dim3(64), 0, s, x, k);
    hipStreamEndCapture(s, &g);
    double ti = now();
    hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    printf("instantiate: %s, %.1f us\n", hipGetErrorString(e), now() - ti);
    for (int rep = 0; rep < 3; rep++) {
        double t0 = now();
        hipGraphLaunch(ge, s);
        double t1 = now();
        hipStreamSynchronize(s);
        double t2 = now();
        printf("graph : launch  %.1f us, total %.1f us = %.2f us per kernel\n", t1 - t0, t2 - t0, (t2 - t0) / N);
    }
    // one kernel alone, for the work per step
    double t0 = now();
    hipLaunchKernelGGL(k_step, dim3(1), dim3(64), 0, s, x, 0);
    hipStreamSynchronize(s);
    printf("single kernel incl. sync %.1f us\n", now() - t0);
    return 0;
}
