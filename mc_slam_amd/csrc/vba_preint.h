// vba_preint.h -- on-device IMU preintegration (A15 of SURVEY 8a): IMUPreintegrator::update,
// src/IMU/IMUPreintegrator.cpp:63-112, one 128-thread workgroup per keyframe interval.  Lanes 0..80 each own one
// entry of the 9x9 covariance (A Sigma A^T + Bg Sg Bg^T + Ca Sa Ca^T through LDS); lane 0 carries the sequential
// Lie-group part (delta R, right Jacobian, the five bias Jacobians, delta P / delta V).
#pragma once
#include "vba_device.h"

__global__ void __launch_bounds__(128) k_preint(int n_edges, const int* sample_begin, const double* gyr, const double* acc,
                                                const double* dts, double gyr_cov, double acc_cov, double* meas_out,
                                                double* cov_out, double* info_out) {
    __shared__ double A[81], Sg[81], T1[81], BgCa[54];  // Bg (9x3: rows 6..8 used) and Ca (9x3: rows 0..5 used)
    __shared__ double st[61];                            // dt, dP, dV, dR, JPg, JPa, JVg, JVa, JRg
    __shared__ double Ms[9 * 18];                        // [Sigma' | I] of the final inverse (row pivoting indexes it dynamically: LDS, not registers)
    const int e = blockIdx.x, t = threadIdx.x;
    if (e >= n_edges) return;
    if (t < 61) st[t] = 0.0;
    if (t < 81) Sg[t] = 0.0;
    __syncthreads();
    if (t == 0) { st[7] = 1.0; st[11] = 1.0; st[15] = 1.0; }
    __syncthreads();
    for (int s = sample_begin[e]; s < sample_begin[e + 1]; s++) {
        const double dt = dts[s], dt2 = dt * dt;
        if (t == 0) {
            const double w[3] = {gyr[3 * s] * dt, gyr[3 * s + 1] * dt, gyr[3 * s + 2] * dt};
            const double a[3] = {acc[3 * s], acc[3 * s + 1], acc[3 * s + 2]};
            double q[4], dRk[9], Jr[9], Sa[9], RS[9];
            so3exp(w, q);
            q2R(q, dRk);         // Expmap, IMUPreintegrator.h:93-96
            so3jr(w, Jr);        // :102-119
            hat3(a, Sa);
            double* dR = st + 7;
            mm3(dR, Sa, RS);
            // A (identity + 4 blocks), Bg, Ca  :75-90, block order P,V,phi
            for (int i = 0; i < 81; i++) A[i] = (i % 10 == 0) ? 1.0 : 0.0;
            for (int i = 0; i < 54; i++) BgCa[i] = 0.0;
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) {
                    A[(6 + r) * 9 + 6 + c] = dRk[3 * c + r];
                    A[(3 + r) * 9 + 6 + c] = -RS[3 * r + c] * dt;
                    A[(0 + r) * 9 + 6 + c] = -0.5 * RS[3 * r + c] * dt2;
                    A[(0 + r) * 9 + 3 + c] = (r == c) ? dt : 0.0;
                    BgCa[(6 + r) * 3 + c] = Jr[3 * r + c] * dt;              // Bg rows 6..8
                    BgCa[27 + (3 + r) * 3 + c] = dR[3 * r + c] * dt;        // Ca rows 3..5
                    BgCa[27 + (0 + r) * 3 + c] = 0.5 * dR[3 * r + c] * dt2; // Ca rows 0..2
                }
            // bias Jacobians :98-102 (each line uses the not-yet-updated values of the ones below it)
            double *JPg = st + 16, *JPa = st + 25, *JVg = st + 34, *JVa = st + 43, *JRg = st + 52;
            double RSJ[9], T[9], dRkT[9];
            mm3(RS, JRg, RSJ);
            for (int i = 0; i < 9; i++) JPa[i] += JVa[i] * dt - 0.5 * dR[i] * dt2;
            for (int i = 0; i < 9; i++) JPg[i] += JVg[i] * dt - 0.5 * RSJ[i] * dt2;
            for (int i = 0; i < 9; i++) JVa[i] += -dR[i] * dt;
            for (int i = 0; i < 9; i++) JVg[i] += -RSJ[i] * dt;
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) dRkT[3 * r + c] = dRk[3 * c + r];
            mm3(dRkT, JRg, T);
            for (int i = 0; i < 9; i++) JRg[i] = T[i] - Jr[i] * dt;
            // deltas :106-110
            double Ra[3];
            mv3(dR, a, Ra);
            for (int k = 0; k < 3; k++) st[1 + k] += st[4 + k] * dt + 0.5 * Ra[k] * dt2;
            for (int k = 0; k < 3; k++) st[4 + k] += Ra[k] * dt;
            mm3(dR, dRk, T);
            double qn[4];
            R2q(T, qn);          // normalizeRotationM, IMUPreintegrator.h:163-174
            if (qn[3] < 0) { qn[0] = -qn[0]; qn[1] = -qn[1]; qn[2] = -qn[2]; qn[3] = -qn[3]; }
            qnorm(qn);
            q2R(qn, dR);
            st[0] += dt;
        }
        __syncthreads();
        if (t < 81) {  // T1 = A Sigma
            const int i = t / 9, j = t % 9;
            double s1 = 0;
#pragma unroll
            for (int k = 0; k < 9; k++) s1 += A[9 * i + k] * Sg[9 * k + j];
            T1[t] = s1;
        }
        __syncthreads();
        if (t < 81) {  // Sigma = T1 A^T + Bg Sg Bg^T + Ca Sa Ca^T
            const int i = t / 9, j = t % 9;
            double s1 = 0, sg = 0, sa = 0;
#pragma unroll
            for (int k = 0; k < 9; k++) s1 += T1[9 * i + k] * A[9 * j + k];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                sg += BgCa[3 * i + k] * gyr_cov * BgCa[3 * j + k];
                sa += BgCa[27 + 3 * i + k] * acc_cov * BgCa[27 + 3 * j + k];
            }
            Sg[t] = s1 + sg + sa;
        }
        __syncthreads();
    }
    if (t < 61) meas_out[61 * (size_t)e + t] = st[t];
    if (t < 81) cov_out[81 * (size_t)e + t] = Sg[t];
    if (info_out && t == 0) {
        // Matrix9d::inverse() of the V/phi-swapped covariance (src/Optimizer.cpp:273-280): Gauss-Jordan, partial pivoting
        const int perm[9] = {0, 1, 2, 6, 7, 8, 3, 4, 5};
        double (*M)[18] = reinterpret_cast<double (*)[18]>(Ms);
        for (int i = 0; i < 9; i++)
            for (int j = 0; j < 9; j++) { M[i][j] = Sg[9 * perm[i] + perm[j]]; M[i][9 + j] = (i == j) ? 1.0 : 0.0; }
        for (int c = 0; c < 9; c++) {
            int p = c;
            for (int r = c + 1; r < 9; r++)
                if (fabs(M[r][c]) > fabs(M[p][c])) p = r;
            if (p != c)
                for (int j = 0; j < 18; j++) { const double tmp = M[c][j]; M[c][j] = M[p][j]; M[p][j] = tmp; }
            const double inv = 1.0 / M[c][c];
            for (int j = 0; j < 18; j++) M[c][j] *= inv;
            for (int r = 0; r < 9; r++) {
                if (r == c) continue;
                const double f = M[r][c];
                for (int j = 0; j < 18; j++) M[r][j] -= f * M[c][j];
            }
        }
        for (int i = 0; i < 9; i++)
            for (int j = 0; j < 9; j++) info_out[81 * (size_t)e + 9 * i + j] = M[i][9 + j];
    }
}
