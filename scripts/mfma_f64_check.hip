// Checks the v_mfma_f64_16x16x4_f64 operand/result lane maps on gfx950 with asymmetric integer data.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* Bm, double* D) {  // A 16x4 row-major, B 4x16 row-major, D 16x16
    const int l = threadIdx.x;
    const double a = A[(l & 15) * 4 + (l >> 4)];
    const double b = Bm[(l >> 4) * 16 + (l & 15)];
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; i++) D[((l >> 4) + 4 * i) * 16 + (l & 15)] = c[i];
}
int main() {
    double hA[64], hB[64], hD[256], ref[256];
    for (int i = 0; i < 16; i++) for (int k = 0; k < 4; k++) hA[i * 4 + k] = i * 3 + k * 7 + 1;
    for (int k = 0; k < 4; k++) for (int j = 0; j < 16; j++) hB[k * 16 + j] = k * 5 - j * 2 + 3;
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { double s = 0; for (int k = 0; k < 4; k++) s += hA[i * 4 + k] * hB[k * 16 + j]; ref[i * 16 + j] = s; }
    double *dA, *dB, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; i++) if (hD[i] != ref[i]) bad++;
    printf("mfma_f64_16x16x4 layout check: %d mismatches\n", bad);
    return bad != 0;
}
