// vba_pose.h -- IMU-aided per-frame pose optimisation on the GPU.
// Replaces, for a batch of independent frames, what Optimizer::PoseOptimization(Frame*, KeyFrame*|Frame*, IMUPreintegrator,
// gw, bComputeMarg) (src/Optimizer.cpp:1671-2317) runs between its vertex set-up and its write-back: four rounds of
// optimize(10) with Levenberg-Marquardt (levenberg.cpp:61-164) on a 15- or 30-dimensional dense system, the
// chi2 > 5.991 reclassification after each round, kernel removal after the third, computeMarginals.
//
// One 64-lane workgroup per frame, the whole protocol in ONE launch: the system (<= 30x30) lives in LDS, lanes share
// the observations (coalesced), lane 0 evaluates the Lie-group parts of the three non-vision edges, every lane helps with
// the small dense products.  Hessian order: [frame PVR | frame Bias | last PVR | last Bias].
#pragma once
#include "vba_device.h"

struct FrameDesc {
    int last_is_frame, compute_marg, n_obs, n_last;
    int obs0, last0;           // offsets into the concatenated observation arrays
    int pad0, pad1;
    double nav[22], nav_last[22], prior_nav[22];
    double K[4], Rcb[9], tcb[3], g[3];
    double meas[61];
    double info_pvr[81];       // inverse of the P,V,phi covariance (host, as the reference's set-up code does)
    double prior_info[225];
    double inv_bg, inv_ba;
    double hub_prior, hub_pvr, hub_bias, hub_mono;  // float-rounded Huber widths (:1741, :2107, :2125, :2137)
};
struct FrameOut {
    int n_inliers, status, its[4];
    int pad[2];
    double chi2_round[4];
    double nav[22];
    double marg[225];
};
struct PoseBatch {
    const FrameDesc* desc;
    FrameOut* out;
    const double *pw, *uv, *w;   // [total obs] current-frame observations first, then the last frames'
    double* err;                 // [total obs][2] stored _error of every mono edge
    unsigned char* lvl;          // [total obs] g2o level (1 = outlier, outside the active set)
    int n_frames;
};

#define PO_N 30
// LDS layout (doubles)
#define PO_A 0                  // damped copy / Cholesky factor (30x30); the Gauss-Jordan workspace [30][60] of the marginals
                                // spans PO_A .. PO_A + 1800, i.e. A and H (H is dead by then): 30.6 KB per frame, five frames per CU
#define PO_H (PO_A + 900)       // 30x30
#define PO_HL (PO_H + 900)      // Hessian of the last linearisation (computeMarginals)
#define PO_J (PO_HL + 900)      // edge Jacobian d x 30 (d <= 15)
#define PO_T (PO_J + 450)       // Omega J
#define PO_B (PO_T + 450)       // b (30)
#define PO_X (PO_B + 30)        // x (30)
#define PO_Y (PO_X + 30)        // y (30)
#define PO_E (PO_Y + 30)        // edge error (15) + Omega e (15)
#define PO_CUR (PO_E + 30)      // nav 22
#define PO_LAST (PO_CUR + 22)
#define PO_CURBK (PO_LAST + 22)
#define PO_LASTBK (PO_CURBK + 22)
#define PO_SC (PO_LASTBK + 22)  // scalars: 0 chi imu part, 1.. rho weights
#define PO_TOTAL (PO_SC + 16)

DEVI double po_bcast(double v) { return rl64(v, 0); }
DEVI double po_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return po_bcast(v);  // every lane continues with lane 0's rounding: control flow stays uniform
}

// EdgeNavStatePVRPointXYZOnlyPose (g2otypes.cpp:792-837): error, depth, and (optionally) the 2x6 nonzero Jacobian
// columns [dP | dR]
DEVI void po_mono(const FrameDesc& d, const double* nav, const double* Rwb, const double* Pw, const double* uv, double* e,
                  double* JP, double* JR) {
    const double dd[3] = {Pw[0] - nav[0], Pw[1] - nav[1], Pw[2] - nav[2]};
    double t1[3], Pa[3];
    mtv3(Rwb, dd, t1);
    mv3(d.Rcb, t1, Pa);
    const double Pc[3] = {Pa[0] + d.tcb[0], Pa[1] + d.tcb[1], Pa[2] + d.tcb[2]};
    const double fx = d.K[0], fy = d.K[1], cx = d.K[2], cy = d.K[3];
    const double iz = 1.0 / Pc[2];
    e[0] = uv[0] - (fx * Pc[0] * iz + cx);
    e[1] = uv[1] - (fy * Pc[1] * iz + cy);
    if (JP) {
        const double Jpi[6] = {fx * iz, 0, -Pc[0] * iz * fx * iz, 0, fy * iz, -Pc[1] * iz * fy * iz};
        double M[9], HA[9], HR[9], RwbT[9];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) RwbT[3 * i + j] = Rwb[3 * j + i];
        mm3(d.Rcb, RwbT, M);
        hat3(Pa, HA);
        mm3(HA, d.Rcb, HR);
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                JP[3 * r + c] = Jpi[3 * r] * M[c] + Jpi[3 * r + 1] * M[3 + c] + Jpi[3 * r + 2] * M[6 + c];
                JR[3 * r + c] = -(Jpi[3 * r] * HR[c] + Jpi[3 * r + 1] * HR[3 + c] + Jpi[3 * r + 2] * HR[6 + c]);
            }
    }
}

// EdgeSE3ProjectXYZOnlyPose (types_six_dof_expmap.cpp): error and the 2x6 Jacobian, rotation columns first
DEVI void po_mono_se3(const FrameDesc& d, const double* T, const double* Pw, const double* uv, double* e, double* J0, double* J1) {
    double Pc[3];
    qrot(T + 3, Pw, Pc);  // SE3Quat::map, se3quat.h:217-220
    Pc[0] += T[0]; Pc[1] += T[1]; Pc[2] += T[2];
    const double fx = d.K[0], fy = d.K[1], cx = d.K[2], cy = d.K[3];
    e[0] = uv[0] - (Pc[0] / Pc[2] * fx + cx);
    e[1] = uv[1] - (Pc[1] / Pc[2] * fy + cy);
    if (J0) {
        const double x = Pc[0], y = Pc[1], z = Pc[2], z_2 = z * z;
        J0[0] = x * y / z_2 * fx; J0[1] = -(1 + (x * x / z_2)) * fx; J0[2] = y / z * fx; J0[3] = -1. / z * fx; J0[4] = 0; J0[5] = x / z_2 * fx;
        J1[0] = (1 + y * y / z_2) * fy; J1[1] = -x * y / z_2 * fy; J1[2] = -x / z * fy; J1[3] = 0; J1[4] = -1. / z * fy; J1[5] = y / z_2 * fy;
    }
}

// EdgeNavStatePVR::computeError (g2otypes.cpp:529-585): rows rP, rV, rPhi
DEVI void po_pvr_error(const FrameDesc& d, const double* ni, const double* nj, double* e) {
    const double* meas = d.meas;
    const double dT = meas[0], dT2 = dT * dT;
    const double *dP = meas + 1, *dV = meas + 4, *dRm = meas + 7;
    const double *JPg = meas + 16, *JPa = meas + 25, *JVg = meas + 34, *JVa = meas + 43, *JRg = meas + 52;
    const double *dbg = ni + 16, *dba = ni + 19;
    double vP[3], vV[3], rv[3], c1[3], c2[3], qiT[4];
    for (int m = 0; m < 3; m++) {
        vP[m] = nj[m] - ni[m] - ni[7 + m] * dT - 0.5 * d.g[m] * dT2;
        vV[m] = nj[7 + m] - ni[7 + m] - d.g[m] * dT;
    }
    so3inv(ni + 3, qiT);
    qrot(qiT, vP, rv);
    mv3(JPg, dbg, c1); mv3(JPa, dba, c2);
    for (int m = 0; m < 3; m++) e[m] = rv[m] - (dP[m] + c1[m] + c2[m]);
    qrot(qiT, vV, rv);
    mv3(JVg, dbg, c1); mv3(JVa, dba, c2);
    for (int m = 0; m < 3; m++) e[3 + m] = rv[m] - (dV[m] + c1[m] + c2[m]);
    double wv[3], qd[4], qR[4], qA[4], qAi[4], qB[4], qC[4];
    mv3(JRg, dbg, wv);
    so3exp(wv, qd);
    R2q(dRm, qR);
    qnorm(qR);
    so3mul(qR, qd, qA);
    so3inv(qA, qAi);
    so3mul(qAi, qiT, qB);
    so3mul(qB, nj + 3, qC);
    so3log(qC, e + 6);
}

// EdgeNavStatePVR::linearizeOplus (:587-701) into J (9 x 30, Hessian column order); lane 0 only
DEVI void po_pvr_jac(const FrameDesc& d, const double* ni, const double* nj, const double* e, double* J, bool last_free) {
    const double* meas = d.meas;
    const double dT = meas[0], dT2 = dT * dT;
    const double *JPg = meas + 16, *JPa = meas + 25, *JVg = meas + 34, *JVa = meas + 43, *JRg = meas + 52;
    const double* dbg = ni + 16;
    for (int q = 0; q < 270; q++) J[q] = 0.0;
    double Ri[9], Rj[9], RiT[9], RjT[9], JrI[9], vP[3], vV[3], mP[3], mV[3], H1[9], H2[9], T1[9], T2[9];
    q2R(ni + 3, Ri);
    q2R(nj + 3, Rj);
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) { RiT[3 * a + b] = Ri[3 * b + a]; RjT[3 * a + b] = Rj[3 * b + a]; }
    so3jrinv(e + 6, JrI);
    for (int m = 0; m < 3; m++) {
        vP[m] = nj[m] - ni[m] - ni[7 + m] * dT - 0.5 * d.g[m] * dT2;
        vV[m] = nj[7 + m] - ni[7 + m] - d.g[m] * dT;
    }
    mv3(RiT, vP, mP);
    mv3(RiT, vV, mV);
    hat3(mP, H1);
    hat3(mV, H2);
    mm3(JrI, RjT, T1);
    mm3(T1, Ri, T2);  // JrInv Rj^T Ri
    double wv[3], qe[4], qei[4], ExpT[9], JrB[9], T3[9], T4[9];
    mv3(JRg, dbg, wv);
    so3exp(e + 6, qe);
    so3inv(qe, qei);
    q2R(qei, ExpT);
    so3jr(wv, JrB);
    mm3(JrI, ExpT, T3);
    mm3(T3, JrB, T4);
    mm3(T4, JRg, T3);  // JrInv Exp(rPhi)^T Jr(JRg dbg) JRg
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) {
            const int q = 3 * a + b;
            // vertex 1 = current frame (columns 0..8: P V R)
            J[(0 + a) * 30 + 0 + b] = RiT[q];
            J[(3 + a) * 30 + 3 + b] = RiT[q];
            J[(6 + a) * 30 + 6 + b] = JrI[q];
            if (last_free) {
                // vertex 0 = last frame PVR (columns 15..23)
                J[(0 + a) * 30 + 15 + b] = -RiT[q];
                J[(0 + a) * 30 + 18 + b] = -RiT[q] * dT;
                J[(0 + a) * 30 + 21 + b] = H1[q];
                J[(3 + a) * 30 + 18 + b] = -RiT[q];
                J[(3 + a) * 30 + 21 + b] = H2[q];
                J[(6 + a) * 30 + 21 + b] = -T2[q];
                // vertex 2 = last frame bias (columns 24..29)
                J[(0 + a) * 30 + 24 + b] = -JPg[q];
                J[(0 + a) * 30 + 27 + b] = -JPa[q];
                J[(3 + a) * 30 + 24 + b] = -JVg[q];
                J[(3 + a) * 30 + 27 + b] = -JVa[q];
                J[(6 + a) * 30 + 24 + b] = -T3[q];
            }
        }
}

// EdgeNavStatePriorPVRBias::computeError (:839-871)
DEVI void po_prior_error(const FrameDesc& d, const double* nl, double* e) {
    const double* pr = d.prior_nav;
    for (int k = 0; k < 3; k++) { e[k] = pr[k] - nl[k]; e[3 + k] = pr[7 + k] - nl[7 + k]; }
    double qi[4], q[4];
    so3inv(pr + 3, qi);
    so3mul(qi, nl + 3, q);
    so3log(q, e + 6);
    for (int k = 0; k < 3; k++) {
        e[9 + k] = (pr[10 + k] + pr[16 + k]) - (nl[10 + k] + nl[16 + k]);
        e[12 + k] = (pr[13 + k] + pr[19 + k]) - (nl[13 + k] + nl[19 + k]);
    }
}

DEVI double po_quad(const double* e, const double* Om, int dd) {
    double s = 0;
    for (int i = 0; i < dd; i++) {
        double t = 0;
        for (int j = 0; j < dd; j++) t += Om[dd * i + j] * e[j];
        s += e[i] * t;
    }
    return s;
}

// VertexNavStatePVR / VertexNavStateBias oplus (NavState.cpp:81-109)
DEVI void po_oplus(double* nav, const double* dpvr, const double* dbias) {
    for (int k = 0; k < 3; k++) { nav[k] += dpvr[k]; nav[7 + k] += dpvr[3 + k]; }
    double dq[4], qn[4];
    so3exp(dpvr + 6, dq);
    so3mul(nav + 3, dq, qn);
    for (int k = 0; k < 4; k++) nav[3 + k] = qn[k];
    for (int k = 0; k < 6; k++) nav[16 + k] += dbias[k];
}

// H += J^T (rw Om) J, b -= J^T (rw Om e) for a dd-row edge whose Jacobian sits in sm[PO_J] (dd x 30); all lanes
DEVI void po_accum(double* sm, const double* Om, int dd, double rw, int n) {
    const int t = threadIdx.x;
    double* J = sm + PO_J;
    double* T = sm + PO_T;
    double* e = sm + PO_E;
    for (int q = t; q < dd * n; q += 64) {
        const int a = q / n, col = q % n;
        double s = 0;
        for (int k = 0; k < dd; k++) s += Om[dd * a + k] * J[k * 30 + col];
        T[a * 30 + col] = s * rw;
    }
    if (t < dd) {
        double s = 0;
        for (int k = 0; k < dd; k++) s += Om[dd * t + k] * e[k];
        e[15 + t] = s * rw;
    }
    __syncthreads();
    for (int q = t; q < n * n; q += 64) {
        const int r = q / n, col = q % n;
        double s = 0;
        for (int k = 0; k < dd; k++) s += J[k * 30 + r] * T[k * 30 + col];
        sm[PO_H + r * n + col] += s;
    }
    if (t < n) {
        double s = 0;
        for (int k = 0; k < dd; k++) s += J[k * 30 + t] * e[15 + k];
        sm[PO_B + t] -= s;
    }
    __syncthreads();
}

// computeActiveErrors + activeRobustChi2 at the state in sm[PO_CUR] / sm[PO_LAST]; returns the robust chi2 (uniform)
DEVI double po_errors(const PoseBatch& B, const FrameDesc& d, double* sm, int vis_robust) {
    const int t = threadIdx.x;
    const double *cur = sm + PO_CUR, *last = sm + PO_LAST;
    double chi = 0.0;
    if (d.last_is_frame == 2) {  // vision only: one VertexSE3Expmap, no IMU edges
        for (int i = t; i < d.n_obs; i += 64) {
            const size_t g = (size_t)d.obs0 + i;
            if (B.lvl[g]) continue;
            double e[2], w;
            po_mono_se3(d, cur, B.pw + 3 * g, B.uv + 2 * g, e, nullptr, nullptr);
            B.err[2 * g] = e[0]; B.err[2 * g + 1] = e[1];
            const double wt = B.w[g];
            const double s = e[0] * (wt * e[0]) + e[1] * (wt * e[1]);
            chi += vis_robust ? huber(s, d.hub_mono, &w) : s;
        }
        return po_wave_sum(chi);
    }
    if (t == 0) {
        double e[15], w;
        if (d.last_is_frame) {
            po_prior_error(d, last, e);
            chi += huber(po_quad(e, d.prior_info, 15), d.hub_prior, &w);
        }
        po_pvr_error(d, last, cur, e);
        chi += huber(po_quad(e, d.info_pvr, 9), d.hub_pvr, &w);
        double eb[6];
        for (int m = 0; m < 3; m++) {
            eb[m] = (cur[10 + m] + cur[16 + m]) - (last[10 + m] + last[16 + m]);
            eb[3 + m] = (cur[13 + m] + cur[19 + m]) - (last[13 + m] + last[19 + m]);
        }
        const double wg = d.inv_bg / d.meas[0], wa = d.inv_ba / d.meas[0];
        chi += huber(wg * (eb[0] * eb[0] + eb[1] * eb[1] + eb[2] * eb[2]) + wa * (eb[3] * eb[3] + eb[4] * eb[4] + eb[5] * eb[5]),
                     d.hub_bias, &w);
    }
    const double dm = d.hub_mono;
    for (int pass = 0; pass < (d.last_is_frame ? 2 : 1); pass++) {
        const int N = pass ? d.n_last : d.n_obs, o0 = pass ? d.last0 : d.obs0;
        const double* nav = pass ? last : cur;
        double Rwb[9];
        q2R(nav + 3, Rwb);
        for (int i = t; i < N; i += 64) {
            const size_t g = (size_t)o0 + i;
            if (B.lvl[g]) continue;
            double e[2], w;
            po_mono(d, nav, Rwb, B.pw + 3 * g, B.uv + 2 * g, e, nullptr, nullptr);
            B.err[2 * g] = e[0]; B.err[2 * g + 1] = e[1];
            const double wt = B.w[g];
            const double s = e[0] * (wt * e[0]) + e[1] * (wt * e[1]);
            chi += vis_robust ? huber(s, dm, &w) : s;
        }
    }
    return po_wave_sum(chi);
}

// buildSystem at the state whose errors were just computed
DEVI void po_build(const PoseBatch& B, const FrameDesc& d, double* sm, int vis_robust) {
    const int t = threadIdx.x, n = (d.last_is_frame == 2) ? 6 : (d.last_is_frame ? 30 : 15);
    double *cur = sm + PO_CUR, *last = sm + PO_LAST;
    for (int q = t; q < 900; q += 64) sm[PO_H + q] = 0.0;
    if (t < 30) sm[PO_B + t] = 0.0;
    __syncthreads();
    if (d.last_is_frame == 2) {
        double acc[21], bb[6];
#pragma unroll
        for (int i = 0; i < 21; i++) acc[i] = 0;
#pragma unroll
        for (int i = 0; i < 6; i++) bb[i] = 0;
        for (int i = t; i < d.n_obs; i += 64) {
            const size_t g = (size_t)d.obs0 + i;
            if (B.lvl[g]) continue;
            double e[2], J0[6], J1[6], rw = 1.0;
            po_mono_se3(d, cur, B.pw + 3 * g, B.uv + 2 * g, e, J0, J1);
            const double wt = B.w[g];
            if (vis_robust) huber(e[0] * (wt * e[0]) + e[1] * (wt * e[1]), d.hub_mono, &rw);
            const double Wt = rw * wt;
            int gi = 0;
#pragma unroll
            for (int a = 0; a < 6; a++) {
                bb[a] -= J0[a] * Wt * e[0] + J1[a] * Wt * e[1];
#pragma unroll
                for (int c = a; c < 6; c++) acc[gi++] += J0[a] * Wt * J0[c] + J1[a] * Wt * J1[c];
            }
        }
#pragma unroll
        for (int i = 0; i < 21; i++) acc[i] = po_wave_sum(acc[i]);
#pragma unroll
        for (int i = 0; i < 6; i++) bb[i] = po_wave_sum(bb[i]);
        if (t == 0) {
            int gi = 0;
            for (int a = 0; a < 6; a++) {
                sm[PO_B + a] = bb[a];
                for (int c = a; c < 6; c++) { sm[PO_H + a * 6 + c] = acc[gi]; sm[PO_H + c * 6 + a] = acc[gi]; gi++; }
            }
        }
        __syncthreads();
        return;
    }
    // vision edges: 21 + 6 sums per frame, lane-strided then reduced in a fixed order
    const double dm = d.hub_mono;
    for (int pass = 0; pass < (d.last_is_frame ? 2 : 1); pass++) {
        const int N = pass ? d.n_last : d.n_obs, o0 = pass ? d.last0 : d.obs0;
        const double* nav = pass ? last : cur;
        const int c0 = pass ? 15 : 0;
        double Rwb[9];
        q2R(nav + 3, Rwb);
        double acc[21], bb[6];
#pragma unroll
        for (int i = 0; i < 21; i++) acc[i] = 0;
#pragma unroll
        for (int i = 0; i < 6; i++) bb[i] = 0;
        for (int i = t; i < N; i += 64) {
            const size_t g = (size_t)o0 + i;
            if (B.lvl[g]) continue;
            double e[2], JP[6], JR[6], rw = 1.0;
            po_mono(d, nav, Rwb, B.pw + 3 * g, B.uv + 2 * g, e, JP, JR);
            const double wt = B.w[g];
            if (vis_robust) huber(e[0] * (wt * e[0]) + e[1] * (wt * e[1]), dm, &rw);
            const double Wt = rw * wt;
            double J0[6], J1[6];
#pragma unroll
            for (int c = 0; c < 3; c++) { J0[c] = JP[c]; J0[3 + c] = JR[c]; J1[c] = JP[3 + c]; J1[3 + c] = JR[3 + c]; }
            int gi = 0;
#pragma unroll
            for (int a = 0; a < 6; a++) {
                bb[a] -= J0[a] * Wt * e[0] + J1[a] * Wt * e[1];
#pragma unroll
                for (int c = a; c < 6; c++) acc[gi++] += J0[a] * Wt * J0[c] + J1[a] * Wt * J1[c];
            }
        }
#pragma unroll
        for (int i = 0; i < 21; i++) acc[i] = po_wave_sum(acc[i]);
#pragma unroll
        for (int i = 0; i < 6; i++) bb[i] = po_wave_sum(bb[i]);
        if (t == 0) {
            int gi = 0;
            for (int a = 0; a < 6; a++) {
                const int ra = c0 + (a < 3 ? a : a + 3);   // P -> 0..2, R -> 6..8
                sm[PO_B + ra] += bb[a];
                for (int c = a; c < 6; c++) {
                    const int rc = c0 + (c < 3 ? c : c + 3);
                    sm[PO_H + ra * n + rc] += acc[gi];
                    if (c != a) sm[PO_H + rc * n + ra] += acc[gi];
                    gi++;
                }
            }
        }
        __syncthreads();
    }
    // prior edge on the last frame (:1735-1747)
    if (d.last_is_frame) {
        if (t == 0) {
            double* J = sm + PO_J;
            double* e = sm + PO_E;
            for (int q = 0; q < 450; q++) J[q] = 0.0;
            po_prior_error(d, last, e);
            double JrI[9], w;
            so3jrinv(e + 6, JrI);
            for (int k = 0; k < 3; k++) { J[k * 30 + 15 + k] = -1.0; J[(3 + k) * 30 + 18 + k] = -1.0; }
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) J[(6 + r) * 30 + 21 + c] = JrI[3 * r + c];
            for (int k = 0; k < 6; k++) J[(9 + k) * 30 + 24 + k] = -1.0;
            huber(po_quad(e, d.prior_info, 15), d.hub_prior, &w);
            sm[PO_SC + 1] = w;
        }
        __syncthreads();
        po_accum(sm, d.prior_info, 15, sm[PO_SC + 1], n);
    }
    // PVR edge
    if (t == 0) {
        double* e = sm + PO_E;
        double w;
        po_pvr_error(d, last, cur, e);
        po_pvr_jac(d, last, cur, e, sm + PO_J, d.last_is_frame != 0);
        huber(po_quad(e, d.info_pvr, 9), d.hub_pvr, &w);
        sm[PO_SC + 1] = w;
    }
    __syncthreads();
    po_accum(sm, d.info_pvr, 9, sm[PO_SC + 1], n);
    // bias edge: J = -I (last), +I (current), Omega diagonal
    if (t == 0) {
        double eb[6], w;
        for (int m = 0; m < 3; m++) {
            eb[m] = (cur[10 + m] + cur[16 + m]) - (last[10 + m] + last[16 + m]);
            eb[3 + m] = (cur[13 + m] + cur[19 + m]) - (last[13 + m] + last[19 + m]);
        }
        const double wg = d.inv_bg / d.meas[0], wa = d.inv_ba / d.meas[0];
        huber(wg * (eb[0] * eb[0] + eb[1] * eb[1] + eb[2] * eb[2]) + wa * (eb[3] * eb[3] + eb[4] * eb[4] + eb[5] * eb[5]),
              d.hub_bias, &w);
        for (int a = 0; a < 6; a++) {
            const double om = w * (a < 3 ? wg : wa);
            sm[PO_H + (9 + a) * n + 9 + a] += om;
            sm[PO_B + 9 + a] -= om * eb[a];
            if (d.last_is_frame) {
                sm[PO_H + (24 + a) * n + 24 + a] += om;
                sm[PO_H + (24 + a) * n + 9 + a] -= om;
                sm[PO_H + (9 + a) * n + 24 + a] -= om;
                sm[PO_B + 24 + a] += om * eb[a];
            }
        }
    }
    __syncthreads();
}

// (H + lambda I) x = b by L L^T (LinearSolverCholmod::solve; false when not positive definite); uniform result
DEVI bool po_solve(double* sm, int n, double lambda) {
    const int t = threadIdx.x;
    double* A = sm + PO_A;
    for (int q = t; q < n * n; q += 64) A[q] = sm[PO_H + q] + ((q / n == q % n) ? lambda : 0.0);
    if (t < n) sm[PO_Y + t] = sm[PO_B + t];
    __syncthreads();
    bool ok = true;
    for (int j = 0; j < n; j++) {
        const double dj = A[j * n + j];
        if (!(dj > 0.0) || !isfinite(dj)) { ok = false; break; }  // uniform: every lane reads the same LDS word
        const double sd = sqrt(dj);
        __syncthreads();
        if (t > j && t < n) A[t * n + j] /= sd;
        if (t == j) A[j * n + j] = sd;
        __syncthreads();
        const int m = n - j - 1;  // trailing update of the lower triangle
        for (int q = t; q < m * m; q += 64) {
            const int i = j + 1 + q / m, k = j + 1 + q % m;
            if (k <= i) A[i * n + k] -= A[i * n + j] * A[k * n + j];
        }
        __syncthreads();
    }
    if (!ok) return false;
    double* y = sm + PO_Y;
    for (int i = 0; i < n; i++) {  // L y = b
        if (t == 0) y[i] /= A[i * n + i];
        __syncthreads();
        if (t > i && t < n) y[t] -= A[t * n + i] * y[i];
        __syncthreads();
    }
    for (int i = n - 1; i >= 0; i--) {  // L^T x = y
        if (t == 0) y[i] /= A[i * n + i];
        __syncthreads();
        if (t < i) y[t] -= A[i * n + t] * y[i];
        __syncthreads();
    }
    if (t < n) sm[PO_X + t] = y[t];
    __syncthreads();
    return true;
}

// inverse of the n x n matrix src (row stride ls) into dst (row stride ld) by Gauss-Jordan with partial pivoting, the
// rows of every elimination step spread over the lanes; workspace sm[PO_A .. PO_A + 60 n): for n = 30 that is PO_A and PO_H.
// src may equal dst (the source is copied into the workspace first)
DEVI void po_inverse(double* sm, const double* src, int ls, int n, double* dst, int ld) {
    const int t = threadIdx.x;
    double* M = sm + PO_A;  // [n][2n], row stride 60
    for (int q = t; q < n * n; q += 64) {
        const int i = q / n, j = q % n;
        M[i * 60 + j] = src[i * ls + j];
        M[i * 60 + n + j] = (i == j) ? 1.0 : 0.0;
    }
    __syncthreads();
    for (int c = 0; c < n; c++) {
        int p = c;
        for (int r = c + 1; r < n; r++)
            if (fabs(M[r * 60 + c]) > fabs(M[p * 60 + c])) p = r;  // uniform
        __syncthreads();
        if (p != c && t < 2 * n) { const double tmp = M[c * 60 + t]; M[c * 60 + t] = M[p * 60 + t]; M[p * 60 + t] = tmp; }
        __syncthreads();
        const double inv = 1.0 / M[c * 60 + c];
        __syncthreads();
        if (t < 2 * n) M[c * 60 + t] *= inv;
        __syncthreads();
        if (t < n && t != c) {
            const double f = M[t * 60 + c];
            if (f != 0.0)
                for (int j = 0; j < 2 * n; j++) M[t * 60 + j] -= f * M[c * 60 + j];
        }
        __syncthreads();
    }
    for (int q = t; q < n * n; q += 64) dst[(q / n) * ld + q % n] = M[(q / n) * 60 + n + q % n];
    __syncthreads();
}

__global__ void __launch_bounds__(64) k_pose_opt(PoseBatch B) {
    __shared__ double sm[PO_TOTAL];
    const int f = blockIdx.x, t = threadIdx.x;
    if (f >= B.n_frames) return;
    const FrameDesc& d = B.desc[f];
    FrameOut& out = B.out[f];
    const bool vision = d.last_is_frame == 2, lif = d.last_is_frame == 1;
    const int n = vision ? 6 : (lif ? 30 : 15);
    for (int i = t; i < d.n_obs; i += 64) B.lvl[(size_t)d.obs0 + i] = 0;  // pFrame->mvbOutlier[i] = false, :2147
    if (lif)
        for (int i = t; i < d.n_last; i += 64) B.lvl[(size_t)d.last0 + i] = 0;
    if (t == 0) {
        out.n_inliers = 0; out.status = 0;
        for (int k = 0; k < 4; k++) { out.its[k] = 0; out.chi2_round[k] = 0.0; }
        for (int k = 0; k < 22; k++) out.nav[k] = d.nav[k];
    }
    for (int q = t; q < 225; q += 64) out.marg[q] = 0.0;
    if (d.n_obs < 3) return;  // nInitialCorrespondences < 3, :2178
    int vis_robust = 1, nBad = 0;
    const int n_edges = d.n_obs + (vision ? 0 : 2) + (lif ? d.n_last + 1 : 0);
    for (int round = 0; round < 4; round++) {
        if (t < 22) { sm[PO_CUR + t] = d.nav[t]; sm[PO_LAST + t] = d.nav_last[t]; }  // setEstimate(...) before every round
        __syncthreads();
        // ---- OptimizationAlgorithmLevenberg::solve x 10 (levenberg.cpp:61-164) ----
        double lambda = 0, ni = 2, cur = 0;
        int cj = 0, nb = 0;
        for (int it = 0; it < 10; it++) {
            cur = po_errors(B, d, sm, vis_robust);
            const double iniChi = cur;
            po_build(B, d, sm, vis_robust);
            for (int q = t; q < n * n; q += 64) sm[PO_HL + q] = sm[PO_H + q];
            if (it == 0) {
                double mx = 0;
                for (int i = 0; i < n; i++) mx = fmax(fabs(sm[PO_H + i * n + i]), mx);
                lambda = 1e-5 * mx;
                ni = 2;
                nb = 0;
            }
            double rho = 0;
            int qmax = 0;
            do {
                if (t < 22) { sm[PO_CURBK + t] = sm[PO_CUR + t]; sm[PO_LASTBK + t] = sm[PO_LAST + t]; }
                __syncthreads();
                const bool ok2 = po_solve(sm, n, lambda);
                if (ok2 && t == 0) {
                    if (vision) se3_oplus(sm + PO_CUR, sm + PO_X);  // VertexSE3Expmap::oplusImpl
                    else {
                        po_oplus(sm + PO_CUR, sm + PO_X, sm + PO_X + 9);
                        if (n == 30) po_oplus(sm + PO_LAST, sm + PO_X + 15, sm + PO_X + 24);
                    }
                }
                __syncthreads();
                double tempChi = po_errors(B, d, sm, vis_robust);
                if (!ok2) tempChi = 1.7976931348623157e308;
                rho = cur - tempChi;
                double scale = 0;
                for (int j = 0; j < n; j++) scale += sm[PO_X + j] * (lambda * sm[PO_X + j] + sm[PO_B + j]);
                scale += 1e-3;
                rho /= scale;
                if (rho > 0 && isfinite(tempChi)) {
                    double alpha = 1. - pow((2 * rho - 1), 3);
                    alpha = fmin(alpha, 2. / 3.);
                    lambda *= fmax(1. / 3., alpha);
                    ni = 2;
                    cur = tempChi;
                } else {
                    lambda *= ni;
                    ni *= 2;
                    __syncthreads();
                    if (t < 22) { sm[PO_CUR + t] = sm[PO_CURBK + t]; sm[PO_LAST + t] = sm[PO_LASTBK + t]; }
                    __syncthreads();
                }
                qmax++;
            } while (rho < 0 && qmax < 10);
            ++cj;
            if (qmax == 10 || rho == 0) break;
            if ((iniChi - cur) * 1e3 < iniChi) nb++;
            else nb = 0;
            if (nb >= 3) break;
        }
        if (t == 0) { out.its[round] = cj; out.chi2_round[round] = cur; }
        // ---- reclassification (:2193-2219): chi2 from the stored error, recomputed for the edges that sat out ----
        for (int pass = 0; pass < (lif ? 2 : 1); pass++) {
            const int N = pass ? d.n_last : d.n_obs, o0 = pass ? d.last0 : d.obs0;
            const double* nav = sm + (pass ? PO_LAST : PO_CUR);
            double Rwb[9];
            q2R(nav + 3, Rwb);
            double bad = 0.0;
            for (int i = t; i < N; i += 64) {
                const size_t g = (size_t)o0 + i;
                double e[2] = {B.err[2 * g], B.err[2 * g + 1]};
                if (B.lvl[g]) {
                    if (vision) po_mono_se3(d, nav, B.pw + 3 * g, B.uv + 2 * g, e, nullptr, nullptr);
                    else po_mono(d, nav, Rwb, B.pw + 3 * g, B.uv + 2 * g, e, nullptr, nullptr);
                    B.err[2 * g] = e[0]; B.err[2 * g + 1] = e[1];
                }
                const double wt = B.w[g];
                const float chi2 = (float)(e[0] * (wt * e[0]) + e[1] * (wt * e[1]));  // const float chi2 = e->chi2()
                const bool b = chi2 > 5.991f;
                B.lvl[g] = b ? 1 : 0;
                bad += b ? 1.0 : 0.0;
            }
            bad = po_wave_sum(bad);
            if (!pass) nBad = (int)(bad + 0.5);
        }
        __syncthreads();
        if (round == 2) vis_robust = 0;  // e->setRobustKernel(0)
        if (n_edges < 10) break;         // optimizer.edges().size() < 10
    }
    if (t == 0) {
        for (int k = 0; k < (vision ? 7 : 10); k++) out.nav[k] = sm[PO_CUR + k];
        if (!vision)
            for (int k = 16; k < 22; k++) out.nav[k] = sm[PO_CUR + k];
        out.n_inliers = d.n_obs - nBad;
    }
    if (d.compute_marg && !vision) {
        // computeMarginals on the Hessian of the last linearisation (:2244-2254 / :2005-2019)
        __syncthreads();
        double* Hi = sm + PO_HL;  // in place: po_inverse copies its source into the workspace before it writes
        po_inverse(sm, sm + PO_HL, n, n, Hi, n);
        double* tmp = sm + PO_H;  // the smaller inverses (n <= 15) only use the PO_A half of the workspace
        if (!lif) {
            po_inverse(sm, Hi, n, 9, tmp, 9);
            for (int q = t; q < 81; q += 64) out.marg[15 * (q / 9) + q % 9] = tmp[q];
            __syncthreads();
            po_inverse(sm, Hi + 9 * n + 9, n, 6, tmp, 6);
            for (int q = t; q < 36; q += 64) out.marg[15 * (9 + q / 6) + 9 + q % 6] = tmp[q];
        } else {
            po_inverse(sm, Hi, n, 15, tmp, 15);
            for (int q = t; q < 225; q += 64) out.marg[q] = tmp[q];
        }
    }
}
