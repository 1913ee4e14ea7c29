// What a compute unit's vector memory path delivers for the record gathers of the Schur walk: scattered 16-byte loads per lane.
//   mode 0: every lane reads 3 x 16 B of its OWN random 64-byte record (the lane-per-item walk: 64 records per instruction)
//   mode 1: the four lanes of a quad read ONE random 64-byte record, 16 B each (16 records per instruction, quad-contiguous)
//   mode 2: every lane reads 1 x 16 B of its own random record (one piece per record)
// table: T bytes of 64-byte records (2 MB: L2-resident per XCD; 64 MB: Infinity Cache / HBM).  Prints lane-loads per clock and CU.
// build: hipcc --offload-arch=gfx950 -O3 -o gather_rate_probe.bin gather_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
template <int MODE>
__global__ void __launch_bounds__(256) k(const double2* tab, const int* idx, int n_rec_mask, int trips, double* out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    double acc = 0.0;
    int cur = idx[t & 0xfffff];
    for (int it = 0; it < trips; it++) {
        int r[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { cur = (cur * 1103515245 + 12345 + t) & 0x7fffffff; r[u] = (cur >> 4) & n_rec_mask; }
        if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 4; u++) r[u] = __shfl(r[u], lane & ~3, 64);
        }
        double2 v[4][3];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const double2* p = tab + 4 * (size_t)r[u];
            if (MODE == 0) { v[u][0] = p[0]; v[u][1] = p[1]; v[u][2] = p[2]; }
            if (MODE == 1) { v[u][0] = p[lane & 3]; }
            if (MODE == 2) { v[u][0] = p[0]; }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            acc += v[u][0].x + v[u][0].y;
            if (MODE == 0) acc += v[u][1].x + v[u][2].y;
        }
    }
    if (acc == 1.2345e300) out[t] = acc;
}
int main(int argc, char** argv) {
    const size_t T = (argc > 1 ? atol(argv[1]) : 2) << 20;
    const int n_rec = (int)(T / 64), trips = 64;
    double2* tab; int* idx; double* out;
    hipMalloc(&tab, T); hipMemset(tab, 0, T);
    hipMalloc(&idx, 4 << 20); hipMalloc(&out, 8 << 20);
    std::vector<int> h(1 << 20); for (auto& x : h) x = rand();
    hipMemcpy(idx, h.data(), 4 << 20, hipMemcpyHostToDevice);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount; const double ghz = pr.clockRate * 1e-6;
    const int blocks = cus * 8 * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; mode++) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; rep++) {
            hipEventRecord(e0);
            if (mode == 0) k<0><<<blocks, 256>>>(tab, idx, n_rec - 1, trips, out);
            if (mode == 1) k<1><<<blocks, 256>>>(tab, idx, n_rec - 1, trips, out);
            if (mode == 2) k<2><<<blocks, 256>>>(tab, idx, n_rec - 1, trips, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        const double lane_loads = (double)blocks * 256 * trips * 4 * (mode == 0 ? 3 : 1);
        const double recs = (double)blocks * 256 * trips * 4 / (mode == 1 ? 4 : 1);
        printf("table %zu MB mode %d: %.3f ms, %.2f lane-loads per clock and CU, %.2f records per clock and CU, %.1f G records/s\n", T >> 20, mode, best,
               lane_loads / (best * 1e-3 * ghz * 1e9 * cus), recs / (best * 1e-3 * ghz * 1e9 * cus), recs / (best * 1e-3) * 1e-9);
    }
    return 0;
}
