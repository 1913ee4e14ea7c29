// H2D / D2H bandwidth of pinned host memory on the GPU box: hipcc --offload-arch=gfx950 -O2 scripts/h2d_bw.hip -o /tmp/h2d_bw
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
int main() {
    const size_t sizes[] = {1u << 20, 16u << 20, 256u << 20};
    void *hp, *dp;
    hipHostMalloc(&hp, 256u << 20, hipHostMallocDefault);
    hipMalloc(&dp, 256u << 20);
    memset(hp, 1, 256u << 20);
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (size_t sz : sizes)
        for (int dir = 0; dir < 2; dir++) {
            const int reps = (int)((1024u << 20) / sz);
            hipStreamSynchronize(st);
            auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < reps; r++)
                dir ? hipMemcpyAsync(hp, dp, sz, hipMemcpyDeviceToHost, st) : hipMemcpyAsync(dp, hp, sz, hipMemcpyHostToDevice, st);
            hipStreamSynchronize(st);
            const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("%s %4zu MiB x %4d: %.1f GB/s\n", dir ? "D2H" : "H2D", sz >> 20, reps, (double)sz * reps / s / 1e9);
        }
    return 0;
}
