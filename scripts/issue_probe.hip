// Issue cost of the instructions the tile elimination is made of, one wave alone on its SIMD (cycles per instruction from s_memtime):
// v_fmac_f64 with the DPP row broadcast, plain v_fma_f64, v_readlane_b32 pairs, ds_bpermute_b32, v_writelane_b32.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
__global__ void k(double* out, unsigned long long* cyc) {
    const int L = threadIdx.x;
    double a0 = L, a1 = L + 1, a2 = L + 2, a3 = L + 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, u = 1.0 + L, s = 1e-9 * L;
    unsigned long long t0, t1, t2, t3, t4, t5;
    t0 = __builtin_amdgcn_s_memtime();
    REP64(asm volatile("v_fmac_f64_dpp %0, -%8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, -%8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                       "v_fmac_f64_dpp %2, -%8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, -%8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                       "v_fmac_f64_dpp %4, -%8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, -%8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                       "v_fmac_f64_dpp %6, -%8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %7, -%8, %9 row_newbcast:8 row_mask:0xf bank_mask:0xf"
                       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(u), "v"(s));)
    t1 = __builtin_amdgcn_s_memtime();
    REP64(asm volatile("v_fma_f64 %0, -%8, %9, %0\n\tv_fma_f64 %1, -%8, %9, %1\n\tv_fma_f64 %2, -%8, %9, %2\n\tv_fma_f64 %3, -%8, %9, %3\n\t"
                       "v_fma_f64 %4, -%8, %9, %4\n\tv_fma_f64 %5, -%8, %9, %5\n\tv_fma_f64 %6, -%8, %9, %6\n\tv_fma_f64 %7, -%8, %9, %7"
                       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(u), "v"(s));)
    t2 = __builtin_amdgcn_s_memtime();
    int lo = __double2loint(a0), hi = __double2hiint(a0), x0 = 0, x1 = 0;
    REP64(asm volatile("v_readlane_b32 %0, %2, 3\n\tv_readlane_b32 %1, %3, 3\n\tv_readlane_b32 %0, %2, 5\n\tv_readlane_b32 %1, %3, 5\n\t"
                       "v_readlane_b32 %0, %2, 7\n\tv_readlane_b32 %1, %3, 7\n\tv_readlane_b32 %0, %2, 9\n\tv_readlane_b32 %1, %3, 9" : "=s"(x0), "=s"(x1) : "v"(lo), "v"(hi));)
    t3 = __builtin_amdgcn_s_memtime();
    int b0 = lo, b1 = hi;
    const int addr = 4 * (L & 15);
    REP64(asm volatile("ds_bpermute_b32 %0, %2, %0\n\tds_bpermute_b32 %1, %2, %1\n\tds_bpermute_b32 %0, %2, %0\n\tds_bpermute_b32 %1, %2, %1\n\t"
                       "ds_bpermute_b32 %0, %2, %0\n\tds_bpermute_b32 %1, %2, %1\n\tds_bpermute_b32 %0, %2, %0\n\tds_bpermute_b32 %1, %2, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(b0), "+v"(b1) : "v"(addr));)
    t4 = __builtin_amdgcn_s_memtime();
    // the dependent chain of one pivot: readlane x2 -> rcp -> 2 Newton steps -> mul
    double y = a1;
    REP16(asm volatile("v_readlane_b32 %1, %3, 3\n\tv_readlane_b32 %2, %4, 3" : "+v"(y), "=s"(x0), "=s"(x1) : "v"(lo), "v"(hi));
          { double piv = __hiloint2double(x1, x0) + 2.0; double r = __builtin_amdgcn_rcp(piv); r = r * (2.0 - piv * r); r = r * (2.0 - piv * r); y = y * r; lo = __double2loint(y); hi = __double2hiint(y); })
    t5 = __builtin_amdgcn_s_memtime();
    out[L] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + x0 + x1 + b0 + b1 + y;
    if (L == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; cyc[4] = t5 - t4; }
}
int main() {
    double* d; unsigned long long* c; hipMalloc(&d, 64 * 8); hipMalloc(&c, 64);
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c);
    unsigned long long h[5]; hipMemcpy(h, c, 40, hipMemcpyDeviceToHost);
    printf("cycles per instruction (s_memtime ticks; 512 instructions each): v_fmac_f64_dpp %.2f, v_fma_f64 %.2f, v_readlane_b32 %.2f, ds_bpermute_b32 %.2f; pivot chain (2 readlane + rcp + 2 Newton + mul) %.1f per round\n",
           h[0] / 512.0, h[1] / 512.0, h[2] / 512.0, h[3] / 512.0, h[4] / 16.0);
    return 0;
}
