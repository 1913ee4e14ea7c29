// vba_kernels_lm.h -- XYZ-landmark variants (0: VertexSE3Expmap + EdgeSE3ProjectXYZ, 1: VertexNavStatePR +
// EdgeNavStatePRPointXYZ) and the Levenberg-Marquardt outer loop (optimization_algorithm_levenberg.cpp:61-164).
//
// LM re-solves the SAME linearisation with a different damping per trial: k_lin_xyz stores the
// undamped products once per outer iteration -- slot record W = Bi^T A (the H_pl block), edge record Bi and g, point
// record H_ll and b_l -- and a trial only recomputes, per LANDMARK, Sigma = (H_ll + lambda I)^-1 and t = Sigma b_l
// (k_dinv: 150 B per landmark).  The Schur term of a keyframe pair is -W_a Sigma W_b^T, formed in the gather from the
// two slot records and the landmark's Sigma, as g2o does (block_solver.hpp:373-430).  (Round 1 rebuilt damped slot
// records U = W chol(Sigma) for every observation and trial: 450 B of HBM traffic per observation, the largest single
// kernel of a vision-only solve.)
#pragma once
#include "vba_kernels.h"

#define LIN_ERR_TRIAL 2

// point record (XYZ): [0..5] H_ll (xx xy xz yy yz zz)  [6..8] b_l  [9] #active edges  [10..15] Sigma = (H_ll + lambda I)^-1
//                     (xx xy xz yy yz zz)  [16..18] t = Sigma b_l
// edge record (XYZ):  [0..11] Bi (2x6 pose Jacobian)  [12..17] g = -Bi^T r   (all pre-scaled)
// slot record (XYZ):  [0..17] W = Bi^T A (6x3, row-major)

// One reprojection edge of an XYZ landmark (EdgeSE3ProjectXYZ / EdgeNavStatePRPointXYZ): depth, chi2, and in LIN_FULL mode its
// edge record (Bi, g) and slot record (W = Bi^T A); returns through A / r the point Jacobian and weighted residual the
// landmark sums need.  `act`: the edge is at level 0 and contributes.
DEVI void xyz_edge(const Batch& B, const WinDesc& d, const WinCtrl& c, size_t go, const double* X, int mode, double& chi, bool& act,
                   double* A, double& r0, double& r1) {
    const double fx = d.K[0], fy = d.K[1], cx = d.K[2], cy = d.K[3];
    act = false;
    const int kf = B.obs_kf[go];
    const double* Ci = B.kfR + 12 * (size_t)(d.kf0 + kf);
    double R[9];
#pragma unroll
    for (int i = 0; i < 9; i++) R[i] = Ci[i];
    double Pc[3], ta[3] = {0, 0, 0};
    if (d.variant == 0) {  // SE3Quat::map: R_cw X + t_cw (se3quat.h:217-220)
        mv3(R, X, Pc);
        Pc[0] += Ci[9]; Pc[1] += Ci[10]; Pc[2] += Ci[11];
    } else {               // Rcb Rwb^T (Pw - Pwb) + tcb (g2otypes.h:289-308)
        const double v[3] = {X[0] - Ci[9], X[1] - Ci[10], X[2] - Ci[11]};
        mtv3(R, v, ta);
        mv3(d.Rcb, ta, Pc);
        Pc[0] += d.tcb[0]; Pc[1] += d.tcb[1]; Pc[2] += d.tcb[2];
    }
    B.depth_e[go] = Pc[2];
    const size_t pe = B.slot_perm[go];                                   // records live keyframe-major (slot_perm)
    double* rec = B.erec + VBA_EREC * (size_t)(d.obs0 + pe);
    double* sl = B.slot + VBA_SLOT3 * (size_t)(d.obs0 + d.pt0 + pe);
    if (B.lvl[go]) {
        if (mode == LIN_FULL)
            for (int i = 0; i < 18; i++) { rec[i] = 0.0; sl[i] = 0.0; }
        return;
    }
    const double iz = 1.0 / Pc[2];
    const double ex = B.obs_uv[2 * go] - (Pc[0] * iz * fx + cx);
    const double ey = B.obs_uv[2 * go + 1] - (Pc[1] * iz * fy + cy);
    const double wgt = B.obs_w[go];
    const double s = ex * (wgt * ex) + ey * (wgt * ey);
    B.chi2_e[go] = s;
    double rw = 1.0;
    if (c.robust_vis) chi += huber(s, d.hub_vis, &rw);
    else chi += s;
    if (mode != LIN_FULL) return;
    act = true;
    const double sc = sqrt(rw * wgt);
    const double x = Pc[0], y = Pc[1], z = Pc[2];
    const double Jp[6] = {fx * iz, 0.0, -x * iz * fx * iz, 0.0, fy * iz, -y * iz * fy * iz};
    double Bi[12];
    const bool of = kf_free(B, d, kf) & 1;
    if (d.variant == 0) {
        // types_six_dof_expmap.cpp:124-138: J_point = -(1/z) tmp R ; J_pose closed form (rotation, then translation)
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int k = 0; k < 3; k++)
                A[3 * r + k] = -sc * (Jp[3 * r] * R[k] + Jp[3 * r + 1] * R[3 + k] + Jp[3 * r + 2] * R[6 + k]);
        const double z2 = z * z;
        Bi[0] = x * y / z2 * fx; Bi[1] = -(1 + (x * x / z2)) * fx; Bi[2] = y / z * fx;
        Bi[3] = -1. / z * fx;    Bi[4] = 0;                        Bi[5] = x / z2 * fx;
        Bi[6] = (1 + y * y / z2) * fy; Bi[7] = -x * y / z2 * fy;   Bi[8] = -x / z * fy;
        Bi[9] = 0;               Bi[10] = -1. / z * fy;            Bi[11] = y / z2 * fy;
#pragma unroll
        for (int i = 0; i < 12; i++) Bi[i] = of ? sc * Bi[i] : 0.0;
    } else {
        double Jc[6], JA[6];
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int k = 0; k < 3; k++)
                Jc[3 * r + k] = Jp[3 * r] * d.Rcb[k] + Jp[3 * r + 1] * d.Rcb[3 + k] + Jp[3 * r + 2] * d.Rcb[6 + k];
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int k = 0; k < 3; k++)
                JA[3 * r + k] = Jc[3 * r] * R[3 * k] + Jc[3 * r + 1] * R[3 * k + 1] + Jc[3 * r + 2] * R[3 * k + 2];
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const double j0 = Jc[3 * r], j1 = Jc[3 * r + 1], j2 = Jc[3 * r + 2];
            const double h0 = j1 * ta[2] - j2 * ta[1], h1 = j2 * ta[0] - j0 * ta[2], h2 = j0 * ta[1] - j1 * ta[0];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                A[3 * r + k] = -sc * JA[3 * r + k];                 // g2otypes.cpp:406
                Bi[6 * r + k] = of ? sc * JA[3 * r + k] : 0.0;      // :409
            }
            Bi[6 * r + 3] = of ? -sc * h0 : 0.0;                    // :412
            Bi[6 * r + 4] = of ? -sc * h1 : 0.0;
            Bi[6 * r + 5] = of ? -sc * h2 : 0.0;
        }
    }
    r0 = sc * ex; r1 = sc * ey;
#pragma unroll
    for (int i = 0; i < 12; i++) rec[i] = Bi[i];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        rec[12 + i] = -(Bi[i] * r0 + Bi[6 + i] * r1);
#pragma unroll
        for (int k = 0; k < 3; k++) sl[3 * i + k] = Bi[i] * A[k] + Bi[6 + i] * A[3 + k];   // W = Bi^T A
    }
}

// a landmark's sums over its edges, in edge order: H_ll += A^T A, b_l -= A^T r
DEVI void xyz_accum(double* Hll, double* bl, const double* A, double r0, double r1) {
    Hll[0] += A[0] * A[0] + A[3] * A[3]; Hll[1] += A[0] * A[1] + A[3] * A[4]; Hll[2] += A[0] * A[2] + A[3] * A[5];
    Hll[3] += A[1] * A[1] + A[4] * A[4]; Hll[4] += A[1] * A[2] + A[4] * A[5]; Hll[5] += A[2] * A[2] + A[5] * A[5];
#pragma unroll
    for (int k = 0; k < 3; k++) bl[k] -= A[k] * r0 + A[3 + k] * r1;
}

// thread per landmark (the fallback for windows with a landmark of more than 256 observations)
DEVI void lin_point_xyz(const Batch& B, const WinDesc& d, const WinCtrl& c, int w, int p, int mode, double& chi,
                        double& maxdiag) {
    const size_t gp = d.pt0 + p;
    const double X[3] = {B.pt[3 * gp], B.pt[3 * gp + 1], B.pt[3 * gp + 2]};
    double Hll[6] = {0, 0, 0, 0, 0, 0}, bl[3] = {0, 0, 0};
    int nact = 0;
    const int* ob = B.pt_obs_begin + d.pt0 + d.win;
    for (int o = ob[p]; o < ob[p + 1]; o++) {
        double A[6] = {0, 0, 0, 0, 0, 0}, r0 = 0, r1 = 0;
        bool act;
        xyz_edge(B, d, c, d.obs0 + o, X, mode, chi, act, A, r0, r1);
        if (!act) continue;
        nact++;
        xyz_accum(Hll, bl, A, r0, r1);
    }
    if (mode == LIN_FULL) {
        double* pr = B.prec + VBA_PREC * gp;
#pragma unroll
        for (int i = 0; i < 6; i++) pr[i] = Hll[i];
        pr[6] = bl[0]; pr[7] = bl[1]; pr[8] = bl[2];
        pr[9] = (double)nact;
        if (nact) maxdiag = fmax(fmax(fabs(Hll[0]), fabs(Hll[3])), fabs(Hll[5]));
    }
}

template <int NT>
DEVI double block_max(double v, double* sm) {
    const int t = threadIdx.x;
    sm[t] = v;
    __syncthreads();
#pragma unroll
    for (int s = NT / 2; s > 0; s >>= 1) {
        if (t < s) sm[t] = fmax(sm[t], sm[t + s]);
        __syncthreads();
    }
    const double r = sm[0];
    __syncthreads();
    return r;
}

__global__ void __launch_bounds__(64) k_lin_xyz(Batch B, int nblk_pt, int mode) {
    __shared__ double sm[64];
    const int w = blockIdx.y;
    const WinDesc& d = B.desc[w];
    const WinCtrl& c = B.ctrl[w];
    if (!c.active) return;
    if (mode == LIN_ERR_TRIAL && !win_on(d, c)) return;
    if (mode == LIN_FULL && d.algo == 1 && c.lm_need_trial) return;   // "outer" slot of the LM schedule: not for a window that still owes a trial
    const int m = (mode == LIN_FULL) ? LIN_FULL : LIN_ERR;
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (d.lin_runs || (int)blockIdx.x * 64 >= d.n_pt) return;   // (windows with a work split are linearised by k_lin_xyz_e)
    double chi = 0.0, mx = 0.0;
    if (p < d.n_pt) lin_point_xyz(B, d, c, w, p, m, chi, mx);
    const double tot = block_sum<64>(chi, sm);
    const double tmx = block_max<64>(mx, sm);
    if (threadIdx.x == 0) {
        B.part[d.part0 + blockIdx.x] = tot;
        if (m == LIN_FULL) B.part[d.part0 + d.n_part_lin + blockIdx.x] = tmx;
    }
}

// The same linearisation, EDGE-parallel (the default): a 256-thread workgroup owns a run of consecutive landmarks with <= 256
// edges (the k_lin2 work split), a lane per edge evaluates residual and Jacobians and writes the edge's two records, a lane per
// landmark then adds up its edges' A^T A and A^T r from LDS in edge order (the sums of the thread-per-landmark form, bit for
// bit).  Six times the parallelism of a thread per landmark and no serial walk over a landmark's observations.
#define LINX_ES 9   // LDS doubles per edge: A (6), r (2), active
__global__ void __launch_bounds__(256) k_lin_xyz_e(Batch B, int mode) {
    __shared__ double ER[256 * LINX_ES];
    __shared__ double red[4];
    const int w = blockIdx.y;
    const WinDesc& d = B.desc[w];
    const WinCtrl& c = B.ctrl[w];
    if (!c.active) return;
    if (mode == LIN_ERR_TRIAL && !win_on(d, c)) return;
    if (mode == LIN_FULL && d.algo == 1 && c.lm_need_trial) return;   // "outer" slot of the LM schedule: not for a window that still owes a trial
    const int m = (mode == LIN_FULL) ? LIN_FULL : LIN_ERR;
    const int lb = blockIdx.x, t = threadIdx.x;
    if (!d.lin_runs || lb >= d.n_part_lin) return;
    const int4 run = reinterpret_cast<const int4*>(B.lin_blk)[d.lb0 + lb];
    const int p0 = run.x, p1 = run.y, e0 = run.z, e1 = run.w;
    const int ne = e1 - e0, npb = p1 - p0;
    double chi = 0.0;
    if (t < ne) {
        const size_t go = d.obs0 + e0 + t;
        const size_t gp = d.pt0 + B.obs_pt[go];
        const double X[3] = {B.pt[3 * gp], B.pt[3 * gp + 1], B.pt[3 * gp + 2]};
        double A[6] = {0, 0, 0, 0, 0, 0}, r0 = 0, r1 = 0;
        bool act;
        xyz_edge(B, d, c, go, X, m, chi, act, A, r0, r1);
        if (m == LIN_FULL) {
            double* er = ER + t * LINX_ES;
#pragma unroll
            for (int i = 0; i < 6; i++) er[i] = A[i];
            er[6] = r0; er[7] = r1; er[8] = act ? 1.0 : 0.0;
        }
    }
    double mx = 0.0;
    if (m == LIN_FULL) {
        __syncthreads();
        if (t < npb) {
            const int* ob = B.pt_obs_begin + d.pt0 + d.win;
            const int p = p0 + t;
            double Hll[6] = {0, 0, 0, 0, 0, 0}, bl[3] = {0, 0, 0};
            int nact = 0;
            for (int o = ob[p] - e0; o < ob[p + 1] - e0; o++) {
                const double* er = ER + o * LINX_ES;
                if (er[8] == 0.0) continue;
                nact++;
                xyz_accum(Hll, bl, er, er[6], er[7]);
            }
            double* pr = B.prec + VBA_PREC * (size_t)(d.pt0 + p);
#pragma unroll
            for (int i = 0; i < 6; i++) pr[i] = Hll[i];
            pr[6] = bl[0]; pr[7] = bl[1]; pr[8] = bl[2];
            pr[9] = (double)nact;
            if (nact) mx = fmax(fmax(fabs(Hll[0]), fabs(Hll[3])), fabs(Hll[5]));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
    }
    const double tot = block_sum256(chi, red);
    if (m == LIN_FULL) {   // (npb <= 64: the landmark lanes are wave 0)
        if (t == 0) B.part[d.part0 + d.n_part_lin + lb] = mx;
    }
    if (t == 0) B.part[d.part0 + lb] = tot;
}

// depth of every observation at the current estimates, nothing else (isDepthPositive() of the erase loops is
// evaluated at the final state even when the stored _error is stale)
__global__ void __launch_bounds__(64) k_depth_xyz(Batch B) {
    const int w = blockIdx.y;
    const WinDesc& d = B.desc[w];
    const int o = blockIdx.x * 64 + threadIdx.x;
    if (o >= d.n_obs || B.ctrl[w].status == 2) return;
    const size_t go = d.obs0 + o;
    const size_t gp = d.pt0 + B.obs_pt[go];
    const double X[3] = {B.pt[3 * gp], B.pt[3 * gp + 1], B.pt[3 * gp + 2]};
    const double* Ci = B.kfR + 12 * (size_t)(d.kf0 + B.obs_kf[go]);
    double z;
    if (d.variant == 0) z = Ci[6] * X[0] + Ci[7] * X[1] + Ci[8] * X[2] + Ci[11];
    else {
        const double v[3] = {X[0] - Ci[9], X[1] - Ci[10], X[2] - Ci[11]};
        double ta[3], Pc[3];
        mtv3(Ci, v, ta);
        mv3(d.Rcb, ta, Pc);
        z = Pc[2] + d.tcb[2];
    }
    B.depth_e[go] = z;
}

// fresh chi2 of every level-0 edge at the final estimates (the parity chi2 of SURVEY 8a N3)
__global__ void __launch_bounds__(64) k_chi2_fresh_xyz(Batch B) {
    const int w = blockIdx.y;
    const WinDesc& d = B.desc[w];
    const int o = blockIdx.x * 64 + threadIdx.x;
    if (o >= d.n_obs || B.ctrl[w].status == 2) return;
    const size_t go = d.obs0 + o;
    const size_t gp = d.pt0 + B.obs_pt[go];
    const double X[3] = {B.pt[3 * gp], B.pt[3 * gp + 1], B.pt[3 * gp + 2]};
    const double* Ci = B.kfR + 12 * (size_t)(d.kf0 + B.obs_kf[go]);
    double Pc[3];
    if (d.variant == 0) {
        mv3(Ci, X, Pc);
        Pc[0] += Ci[9]; Pc[1] += Ci[10]; Pc[2] += Ci[11];
    } else {
        const double v[3] = {X[0] - Ci[9], X[1] - Ci[10], X[2] - Ci[11]};
        double ta[3];
        mtv3(Ci, v, ta);
        mv3(d.Rcb, ta, Pc);
        Pc[0] += d.tcb[0]; Pc[1] += d.tcb[1]; Pc[2] += d.tcb[2];
    }
    const double ex = B.obs_uv[2 * go] - (Pc[0] / Pc[2] * d.K[0] + d.K[2]);
    const double ey = B.obs_uv[2 * go + 1] - (Pc[1] / Pc[2] * d.K[1] + d.K[3]);
    const double wgt = B.obs_w[go];
    B.chi2_f[go] = ex * (wgt * ex) + ey * (wgt * ey);
}

// K_dinv: the damped landmark blocks of one trial, Sigma = (H_ll + lambda I)^-1 and t = Sigma b_l (one thread per landmark)
__global__ void __launch_bounds__(64) k_dinv(Batch B) {
    const int w = blockIdx.y;
    const WinDesc& d = B.desc[w];
    const WinCtrl& c = B.ctrl[w];
    if (!win_on(d, c)) return;
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= d.n_pt) return;
    const size_t gp = d.pt0 + p;
    double* pr = B.prec + VBA_PREC * gp;
    const double lam = (d.algo == 1) ? c.lambda : 0.0;
    const bool on = pr[9] > 0.0;
    double i00 = 0, i01 = 0, i02 = 0, i11 = 0, i12 = 0, i22 = 0;
    if (on) {
        const double h00 = pr[0] + lam, h01 = pr[1], h02 = pr[2], h11 = pr[3] + lam, h12 = pr[4], h22 = pr[5] + lam;
        // Matrix3d::inverse() (cofactors), block_solver.hpp:389
        const double c00 = h11 * h22 - h12 * h12, c01 = h02 * h12 - h01 * h22, c02 = h01 * h12 - h02 * h11;
        const double idet = 1.0 / (h00 * c00 + h01 * c01 + h02 * c02);
        i00 = c00 * idet; i01 = c01 * idet; i02 = c02 * idet;
        i11 = (h00 * h22 - h02 * h02) * idet; i12 = (h01 * h02 - h00 * h12) * idet;
        i22 = (h00 * h11 - h01 * h01) * idet;
    }
    const double b0 = pr[6], b1 = pr[7], b2 = pr[8];
    pr[10] = i00; pr[11] = i01; pr[12] = i02; pr[13] = i11; pr[14] = i12; pr[15] = i22;
    pr[16] = i00 * b0 + i01 * b1 + i02 * b2;
    pr[17] = i01 * b0 + i11 * b1 + i12 * b2;
    pr[18] = i02 * b0 + i12 * b1 + i22 * b2;
}

// SE3Quat::exp(update) * T  (se3quat.h:223-257, :103-109; VertexSE3Expmap::oplusImpl types_six_dof_expmap.h:73-76)
DEVI void se3_oplus(double* T, const double* dx) {
    const double* om = dx;
    const double* up = dx + 3;
    const double th = nrm3(om);
    double Om[9], Om2[9], R[9], V[9];
    hat3(om, Om);
    mm3(Om, Om, Om2);
    if (th < 0.00001) {
#pragma unroll
        for (int i = 0; i < 9; i++) { R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + Om[i] + Om2[i]; V[i] = R[i]; }
    } else {
        const double a = sin(th) / th, b = (1 - cos(th)) / (th * th), cc = (th - sin(th)) / (th * th * th);
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const double I = (i % 4 == 0) ? 1.0 : 0.0;
            R[i] = I + a * Om[i] + b * Om2[i];
            V[i] = I + b * Om[i] + cc * Om2[i];
        }
    }
    double qe[4], te[3];
    R2q(R, qe);
    if (qe[3] < 0) { qe[0] = -qe[0]; qe[1] = -qe[1]; qe[2] = -qe[2]; qe[3] = -qe[3]; }
    qnorm(qe);
    mv3(V, up, te);
    double tn[3], qn[4];
    qrot(qe, T, tn);
    tn[0] += te[0]; tn[1] += te[1]; tn[2] += te[2];
    qmul(qe, T + 3, qn);
    if (qn[3] < 0) { qn[0] = -qn[0]; qn[1] = -qn[1]; qn[2] = -qn[2]; qn[3] = -qn[3]; }
    qnorm(qn);
    T[0] = tn[0]; T[1] = tn[1]; T[2] = tn[2];
    T[3] = qn[0]; T[4] = qn[1]; T[5] = qn[2]; T[6] = qn[3];
}

__global__ void __launch_bounds__(64) k_update_xyz(Batch B, int nblk_pt) {
    __shared__ double sm[64];
    const int w = blockIdx.y;
    const WinDesc& d = B.desc[w];
    const WinCtrl& c = B.ctrl[w];
    if (!win_on(d, c) || c.chol_fail) return;
    const double* x = B.vec + d.vec0;
    const int P = d.pdim;
    if ((int)blockIdx.x < nblk_pt) {
        const int p = blockIdx.x * 64 + threadIdx.x;
        if ((int)blockIdx.x * 64 >= d.n_pt) return;
        double sc = 0.0;
        if (p < d.n_pt) {
            const size_t gp = d.pt0 + p;
            const double* pr = B.prec + VBA_PREC * gp;
            if (pr[9] > 0.0) {
                double v0 = pr[6], v1 = pr[7], v2 = pr[8];   // b_l - sum_k W_kl^T x_k   (block_solver.hpp:461-481)
                const int* ob = B.pt_obs_begin + d.pt0 + d.win;
                const double* slots = B.slot + VBA_SLOT3 * (size_t)(d.obs0 + d.pt0);
                for (int o = ob[p]; o < ob[p + 1]; o++) {
                    const int kf = B.obs_kf[d.obs0 + o];
                    if (kf >= d.n_free) continue;
                    const double* sl = slots + VBA_SLOT3 * (size_t)B.slot_perm[d.obs0 + o];
#pragma unroll
                    for (int i = 0; i < 6; i++) {
                        const double xi = x[vpos(d, kf, i)];
                        v0 -= sl[3 * i] * xi; v1 -= sl[3 * i + 1] * xi; v2 -= sl[3 * i + 2] * xi;
                    }
                }
                const double dl0 = pr[10] * v0 + pr[11] * v1 + pr[12] * v2, dl1 = pr[11] * v0 + pr[13] * v1 + pr[14] * v2,
                             dl2 = pr[12] * v0 + pr[14] * v1 + pr[15] * v2;   // Sigma v
                B.pt[3 * gp] += dl0; B.pt[3 * gp + 1] += dl1; B.pt[3 * gp + 2] += dl2;  // VertexSBAPointXYZ::oplusImpl
                const double lam = (d.algo == 1) ? c.lambda : 0.0;
                sc = dl0 * (lam * dl0 + pr[6]) + dl1 * (lam * dl1 + pr[7]) + dl2 * (lam * dl2 + pr[8]);  // computeScale
            }
        }
        const double tot = block_sum<64>(sc, sm);
        if (threadIdx.x == 0) B.part[d.part0 + 2 * d.n_part_lin + blockIdx.x] = tot;
    } else {
        const int a = (blockIdx.x - nblk_pt) * 64 + threadIdx.x;
        if (a >= d.n_free) return;
        const int* va = B.var_act + d.vec0;
        double dx[15];   // (constant indices only: a loop up to the run-time P put the array into scratch memory)
#pragma unroll
        for (int i = 0; i < 6; i++) dx[i] = x[vpos(d, a, i)];
#pragma unroll
        for (int i = 6; i < 15; i++) dx[i] = (P == 15) ? x[vpos(d, a, i)] : 0.0;
        const size_t gk = d.kf0 + a;
        if (va[vpos(d, a, 0)]) {
            double* T = B.pose + 7 * gk;
            if (d.variant == 0) se3_oplus(T, dx);
            else {
                T[0] += dx[0]; T[1] += dx[1]; T[2] += dx[2];
                double dq[4], qn[4];
                so3exp(dx + 3, dq);
                so3mul(T + 3, dq, qn);
                T[3] = qn[0]; T[4] = qn[1]; T[5] = qn[2]; T[6] = qn[3];
            }
            kf_cache(B, d, a);
        }
        if (P == 15) {
            if (va[vpos(d, a, 6)])
                for (int i = 0; i < 3; i++) B.vel[3 * gk + i] += dx[6 + i];
            if (va[vpos(d, a, 9)])
                for (int i = 0; i < 6; i++) B.bias[12 * gk + 6 + i] += dx[9 + i];
        }
    }
}

// push / pop of the estimates around an LM trial (SparseOptimizer::push / pop / discardTop)
__global__ void __launch_bounds__(64) k_backup(Batch B) {
    const int w = blockIdx.y;
    const WinDesc& d = B.desc[w];
    WinCtrl& c = B.ctrl[w];
    if (!win_on(d, c)) return;
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t < d.n_free) {
        const size_t k = d.kf0 + t;
        for (int i = 0; i < 7; i++) B.pose_bk[7 * k + i] = B.pose[7 * k + i];
        for (int i = 0; i < 3; i++) B.vel_bk[3 * k + i] = B.vel[3 * k + i];
        for (int i = 0; i < 12; i++) B.bias_bk[12 * k + i] = B.bias[12 * k + i];
    }
    if (t < d.n_pt) {
        const size_t p = d.pt0 + t;
        for (int i = 0; i < 3; i++) B.pt_bk[3 * p + i] = B.pt[3 * p + i];
    }
}
__global__ void __launch_bounds__(64) k_restore(Batch B) {
    const int w = blockIdx.y;
    const WinDesc& d = B.desc[w];
    const WinCtrl& c = B.ctrl[w];
    if (!c.lm_restore) return;
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t < d.n_free) {
        const size_t k = d.kf0 + t;
        for (int i = 0; i < 7; i++) B.pose[7 * k + i] = B.pose_bk[7 * k + i];
        for (int i = 0; i < 3; i++) B.vel[3 * k + i] = B.vel_bk[3 * k + i];
        for (int i = 0; i < 12; i++) B.bias[12 * k + i] = B.bias_bk[12 * k + i];
        kf_cache(B, d, t);
    }
    if (t < d.n_pt) {
        const size_t p = d.pt0 + t;
        for (int i = 0; i < 3; i++) B.pt[3 * p + i] = B.pt_bk[3 * p + i];
    }
}

// outer LM iteration begin: currentChi, lambda init on iteration 0 (computeLambdaInit :166-180)
__global__ void __launch_bounds__(64) k_ctrl_lm_outer(Batch B) {
    __shared__ double sm[64];
    const int w = blockIdx.x;
    const WinDesc& d = B.desc[w];
    WinCtrl& c = B.ctrl[w];
    if (!c.active || c.lm_need_trial) return;   // the window is still inside the trials of its current outer iteration
    const int t = threadIdx.x;
    const double cur = window_chi2(B, d, sm);
    double mx = 0.0;
    if (c.it == 0) {
        const int* va = B.var_act + d.vec0;
        const double* hd = B.bpose + 2 * (size_t)d.vec0 + d.nS;
        for (int i = t; i < d.nS; i += 64)
            if (va[i]) mx = fmax(mx, fabs(hd[i]));
        for (int k = t; k < d.n_part_lin; k += 64) mx = fmax(mx, B.part[d.part0 + d.n_part_lin + k]);
        mx = block_max<64>(mx, sm);
    }
    if (t != 0) return;
    if (poll_stop(B, c)) {  // terminate() before the iteration
        c.aborted = 1;
        if (c.stage == 0) c.status = 1;
        c.active = 0;
        c.lm_need_trial = 0;
        return;
    }
    if (c.it == 0) {
        if (c.n_trace < VBA_TRACE) c.trace[c.n_trace++] = cur;
        c.lambda = 1e-5 * mx;
        c.ni = 2;
        c.nbad = 0;
    }
    c.chi_prev = cur;
    c.chi_ini = cur;
    c.lm_trial = 0;
    c.lm_need_trial = 1;
    c.lm_restore = 0;
    c.chol_fail = 0;
}

// one LM trial has been solved, applied and re-evaluated: rho test, lambda update, accept / reject, and when
// the do-while ends the per-iteration stop rules (levenberg.cpp:120-161)
// alive: pinned host word of this slot group; set when the window goes on (another trial or another outer iteration)
__global__ void __launch_bounds__(64) k_ctrl_lm_trial(Batch B, int* alive, int* alive_mirror) {
    __shared__ double sm[64];
    const int w = blockIdx.x;
    const WinDesc& d = B.desc[w];
    WinCtrl& c = B.ctrl[w];
    if (!c.active || !c.lm_need_trial) return;
    const int t = threadIdx.x;
    double tempChi = window_chi2(B, d, sm);
    double sc = 0.0;
    {
        const int* va = B.var_act + d.vec0;
        const double* x = B.vec + d.vec0;
        const double* bp = B.bpose + 2 * (size_t)d.vec0;
        for (int i = t; i < d.nS; i += 64)
            if (va[i]) sc += x[i] * (c.lambda * x[i] + bp[i]);
        for (int k = t; k < d.n_part_pt; k += 64) sc += B.part[d.part0 + 2 * d.n_part_lin + k];
        sc = block_sum<64>(sc, sm);
    }
    if (t != 0) return;
    if (c.chol_fail) { tempChi = 1.7976931348623157e308; sc = 0.0; }
    double cur = c.chi_prev;
    double rho = (cur - tempChi) / (sc + 1e-3);
    if (rho > 0 && isfinite(tempChi)) {
        double alpha = 1. - (2 * rho - 1) * (2 * rho - 1) * (2 * rho - 1);
        alpha = fmin(alpha, 2. / 3.);
        c.lambda *= fmax(1. / 3., alpha);
        c.ni = 2;
        cur = tempChi;
        c.lm_restore = 0;
    } else {
        c.lambda *= c.ni;
        c.ni *= 2;
        c.lm_restore = 1;
    }
    c.chi_prev = cur;
    c.chol_fail = 0;
    const int qmax = ++c.lm_trial;
    if (rho < 0 && qmax < 10 && !poll_stop(B, c)) {   // (short-circuit: terminate() is only called when a retry is due, levenberg.cpp:149)
        c.lm_need_trial = 1;
        if (atomicExch(alive_mirror, 1) == 0) __hip_atomic_store(alive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // a posted store, not a PCIe atomic
        return;
    }
    c.lm_need_trial = 0;
    if (c.n_trace < VBA_TRACE) c.trace[c.n_trace++] = cur;
    const int st = c.stage;
    c.it += 1;
    c.its_done[st] = c.it;
    bool term = (qmax == 10 || rho == 0);
    if (!term) {
        if ((c.chi_ini - cur) * 1e3 < c.chi_ini) c.nbad++;
        else c.nbad = 0;
        if (c.nbad >= 3) term = true;
    }
    if (term || c.it >= d.its[st]) c.active = 0;
    else if (atomicExch(alive_mirror, 1) == 0) __hip_atomic_store(alive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // goes on to another outer iteration
}
