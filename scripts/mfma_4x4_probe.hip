// Probe of v_mfma_f64_4x4x4_4b_f64 on gfx950: (1) which lane of A meets which lane of B in which lane of D -- one-hot inputs over all
// 64 x 64 lane pairs; (2) issue rate: a chain of dependent and of independent instructions, shader clock.
// build: hipcc --offload-arch=gfx950 -O3 scripts/mfma_4x4_probe.hip -o scripts/mfma_4x4_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_map(unsigned long long* out) {
    const int l = threadIdx.x;
    for (int la = 0; la < 64; la++)
        for (int lb = 0; lb < 64; lb++) {
            const double a = (l == la) ? 1.0 : 0.0, b = (l == lb) ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            const unsigned long long m = __ballot(d != 0.0);
            if (l == 0) out[la * 64 + lb] = m;
        }
}
__global__ void k_rate(double* out, unsigned long long* cyc) {
    const int l = threadIdx.x;
    double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 256; i++) c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
    asm volatile("" :: "v"(c0));
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 64; i++) {
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
    }
    asm volatile("" :: "v"(c0), "v"(c1), "v"(c2), "v"(c3));
    unsigned long long t2 = __builtin_amdgcn_s_memtime();
    out[l] = c0 + c1 + c2 + c3;
    if (l == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
}
int main() {
    unsigned long long *dm, hm[4096], *dc, hc[2];
    double* dout;
    hipMalloc(&dm, sizeof hm); hipMalloc(&dc, sizeof hc); hipMalloc(&dout, 64 * 8);
    hipLaunchKernelGGL(k_map, dim3(1), dim3(64), 0, 0, dm);
    hipMemcpy(hm, dm, sizeof hm, hipMemcpyDeviceToHost);
    // for every A lane: the set of B lanes it meets, and the D lanes
    for (int la = 0; la < 64; la++) {
        printf("A lane %2d meets B lanes:", la);
        for (int lb = 0; lb < 64; lb++)
            if (hm[la * 64 + lb]) {
                int dl = __builtin_ctzll(hm[la * 64 + lb]);
                printf(" %d->D%d%s", lb, dl, __builtin_popcountll(hm[la * 64 + lb]) > 1 ? "+" : "");
            }
        printf("\n");
    }
    hipLaunchKernelGGL(k_rate, dim3(1), dim3(64), 0, 0, dout, dc);
    hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost);
    printf("256 dependent: %llu cycles (%.1f each); 256 in four independent chains: %llu cycles (%.1f each)\n", hc[0], hc[0] / 256.0, hc[1], hc[1] / 256.0);
    return 0;
}
