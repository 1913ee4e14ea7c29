// How much does a grid-wide barrier cost inside ONE launch (all workgroups resident), against the launch-to-launch gap of dependent
// kernels in a stream?  nwg workgroups of 128 threads, nbar barriers; every workgroup writes a word before and reads its
// neighbour's word after each barrier (so the fences carry data).  Spins are bounded.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void __launch_bounds__(128) k_bar(unsigned* cnt, unsigned* data, int nbar, unsigned* err) {
    const unsigned nwg = gridDim.x, me = blockIdx.x;
    unsigned bad = 0;
    for (int b = 0; b < nbar; b++) {
        if (threadIdx.x == 0) data[me] = (unsigned)b * 7919u + me;
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            atomicAdd(cnt, 1u);
            const unsigned target = (unsigned)(b + 1) * nwg;
            long spins = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && spins < 20000000) { __builtin_amdgcn_s_sleep(1); spins++; }
            if (spins >= 20000000) bad = 1;
            __threadfence();
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned nb = (me + 1) % nwg;
            const unsigned v = __hip_atomic_load(&data[nb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v != (unsigned)b * 7919u + nb) bad |= 2;
        }
        // second barrier so that nobody overwrites data[] before its neighbour has read it (counts as a barrier of its own)
        __syncthreads();
    }
    if (bad && threadIdx.x == 0) atomicOr(err, bad);
}
__global__ void __launch_bounds__(128) k_one(unsigned* data, int b) {
    if (threadIdx.x == 0) data[blockIdx.x] = data[(blockIdx.x + 1) % gridDim.x] + b;
}
int main(int argc, char** argv) {
    const int nwg = argc > 1 ? atoi(argv[1]) : 253, nbar = argc > 2 ? atoi(argv[2]) : 2000;
    unsigned *cnt, *data, *err;
    CHK(hipMalloc(&cnt, 4)); CHK(hipMalloc(&data, 4 * nwg)); CHK(hipMalloc(&err, 4));
    for (int rep = 0; rep < 3; rep++) {
        CHK(hipMemset(cnt, 0, 4)); CHK(hipMemset(err, 0, 4)); CHK(hipMemset(data, 0, 4 * nwg));
        CHK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        void* args[] = {&cnt, &data, (void*)&nbar, &err};
        CHK(hipLaunchCooperativeKernel((void*)k_bar, dim3(nwg), dim3(128), args, 0, 0));
        CHK(hipDeviceSynchronize());
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        unsigned e = 0; CHK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
        printf("%d workgroups, %d barriers in one launch: %.2f us per barrier (err %u)\n", nwg, nbar, us / nbar, e);
    }
    for (int rep = 0; rep < 3; rep++) {
        CHK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        for (int b = 0; b < nbar; b++) hipLaunchKernelGGL(k_one, dim3(nwg), dim3(128), 0, 0, data, b);
        CHK(hipDeviceSynchronize());
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("%d workgroups, %d dependent launches: %.2f us per launch\n", nwg, nbar, us / nbar);
    }
    return 0;
}
