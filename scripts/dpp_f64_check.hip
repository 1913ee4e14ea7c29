#include <hip/hip_runtime.h>
#include <cstdio>
template <int N, bool NOP>
__device__ __forceinline__ void fmac_bcast(double& acc, double urep, double s) {
    if (NOP) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(urep), "v"(s), "n"(N));
    else asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(urep), "v"(s), "n"(N));
}
__device__ __forceinline__ double swz16(double v) {
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), 0x401f);
    const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), 0x401f);
    return __hiloint2double(hi, lo);
}
template <int N>
__device__ __forceinline__ void fnmac_bcast(double& acc, double urep, double s) {   // acc -= urep[lane N of the row] * s  (source modifier in the DPP word)
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(urep), "v"(s), "n"(N));
}
__device__ __forceinline__ double bperm(double v, int src_lane) {   // the value of lane src_lane
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane * 4, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(src_lane * 4, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
__global__ void k2(double* out) {
    const int L = threadIdx.x;
    double u = 100.0 + L, s = 1.0 + 0.001 * L, acc = 5.0;
    const double uLow = bperm(u, L & 15), uHigh = bperm(u, 16 + (L & 15));   // lanes 0..15 / 16..31 replicated into every row
    fnmac_bcast<3>(acc, uLow, s);        // acc = 5 - u[3] * s in every row
    double acc2 = 5.0;
    fnmac_bcast<7>(acc2, uHigh, s);      // 5 - u[23] * s
    out[L] = acc; out[64 + L] = acc2;
}
__global__ void k(double* out) {
    const int L = threadIdx.x;
    double u = 100.0 + L, s = 1.0 + 0.001 * L, acc = 0.0, acc2 = 0.0;
    fmac_bcast<3, true>(acc, u, s);      // acc = u[row*16+3] * s
    fmac_bcast<15, false>(acc2, u, s);
    out[L] = acc; out[64 + L] = acc2; out[128 + L] = swz16(u);
}
int main() {
    double* d; hipMalloc(&d, 192 * 8);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    double h[192]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int L = 0; L < 64; L++) {
        const double s = 1.0 + 0.001 * L;
        const double e1 = (100.0 + (L & ~15) + 3) * s, e2 = (100.0 + (L & ~15) + 15) * s, e3 = 100.0 + (L ^ 16);
        if (h[L] != e1 || h[64 + L] != e2 || h[128 + L] != e3) { bad++; printf("lane %d: %g %g %g expected %g %g %g\n", L, h[L], h[64+L], h[128+L], e1, e2, e3); }
    }
    hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int L = 0; L < 64; L++) {
        const double s = 1.0 + 0.001 * L;
        const double e1 = fma(-103.0, s, 5.0), e2 = fma(-123.0, s, 5.0);
        if (h[L] != e1 || h[64 + L] != e2) { bad++; printf("k2 lane %d: %.17g %.17g expected %.17g %.17g\n", L, h[L], h[64+L], e1, e2); }
    }
    printf("dpp test: %s\n", bad ? "FAIL" : "ok");
    return bad != 0;
}
