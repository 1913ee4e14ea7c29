/*
 * vba_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Plain-C, double-precision restatement of the reference's local-BA algorithm (mc275/MC_SLAM):
 * g2o's sparse optimiser + the VI edge types, as called from src/Optimizer.cpp.  It exists only to
 * check the HIP backend (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).  The product
 * path (mc_slam_amd/csrc) never links or calls anything in this file.
 *
 * PARITY UNPINNED: the reference ships no tests, fixtures or golden vectors for this path
 *  and cannot be compiled here (needs Eigen3/OpenCV, absent from the image),
 * so this restatement is pinned only by its own known-answer tests (tests/test_oracle_*.py):
 * central-difference Jacobians through the same retractions, closed-form preintegration,
 * Schur-vs-full-system solves against numpy.  Third-party arithmetic it restates: Eigen3 (>=3.1.0,
 * unpinned by the reference): Quaterniond <-> Matrix3d, Matrix3d::inverse(), SimplicialLDLT (here:
 * LDL^T without pivoting on the dense reduced system; same solution up to rounding).
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 */
#include "../include/vislam_ba.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------------------------------
 * small dense helpers (row-major 3x3)
 * ---------------------------------------------------------------------------------------------- */
static void m3_mul(const double *A, const double *B, double *C) {
    double t[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    memcpy(C, t, sizeof t);
}
static void m3_T(const double *A, double *At) {
    double t[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * j + i];
    memcpy(At, t, sizeof t);
}
static void m3_vec(const double *A, const double *v, double *o) {
    double t[3];
    for (int i = 0; i < 3; i++) t[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
    o[0] = t[0]; o[1] = t[1]; o[2] = t[2];
}
__attribute__((unused)) static void m3T_vec(const double *A, const double *v, double *o) {
    double t[3];
    for (int i = 0; i < 3; i++) t[i] = A[i] * v[0] + A[3 + i] * v[1] + A[6 + i] * v[2];
    o[0] = t[0]; o[1] = t[1]; o[2] = t[2];
}
static void m3_id(double *A) { memset(A, 0, 9 * sizeof(double)); A[0] = A[4] = A[8] = 1.0; }
static double v3_norm(const double *v) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

/* Sophus::SO3::hat, src/IMU/so3.cpp:263-271 (== g2o skew, Thirdparty/g2o/g2o/types/se3_ops.hpp:27-38) */
static void hat(const double *v, double *M) {
    M[0] = 0; M[1] = -v[2]; M[2] = v[1];
    M[3] = v[2]; M[4] = 0; M[5] = -v[0];
    M[6] = -v[1]; M[7] = v[0]; M[8] = 0;
}

/* ---- Eigen::Quaterniond restated (coefficient order x,y,z,w as Eigen stores it) ---- */
/* Eigen QuaternionBase::toRotationMatrix */
static void quat_to_R(const double *q, double *R) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
/* Eigen quaternion product a*b */
static void quat_mul(const double *a, const double *b, double *o) {
    double t[4];
    t[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    t[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    t[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    t[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    memcpy(o, t, sizeof t);
}
static void quat_normalize(double *q) {
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
static void quat_conj(const double *q, double *o) { o[0] = -q[0]; o[1] = -q[1]; o[2] = -q[2]; o[3] = q[3]; }
/* Eigen QuaternionBase::_transformVector */
static void quat_rot(const double *q, const double *v, double *o) {
    double uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    const double c[3] = {q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0]};
    const double r0 = v[0] + q[3] * uv[0] + c[0], r1 = v[1] + q[3] * uv[1] + c[1], r2 = v[2] + q[3] * uv[2] + c[2];
    o[0] = r0; o[1] = r1; o[2] = r2;
}
/* Eigen quaternion-from-rotation-matrix (internal::quaternionbase_assign_impl<Other,3,3>) */
static void R_to_quat(const double *m, double *q) {
    double t = m[0] + m[4] + m[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m[7] - m[5]) * t;
        q[1] = (m[2] - m[6]) * t;
        q[2] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (m[3 * k + j] - m[3 * j + k]) * t;
        q[j] = (m[3 * j + i] + m[3 * i + j]) * t;
        q[k] = (m[3 * k + i] + m[3 * i + k]) * t;
    }
}

/* ------------------------------------------------------------------------------------------------
 * A8  Sophus SO3 restated (src/IMU/so3.cpp)
 * ---------------------------------------------------------------------------------------------- */
/* SO3::expAndTheta, so3.cpp:237-261 (SMALL_EPS 1e-10, so3.h:36); result normalised by SO3(Quaterniond), :105-109 */
void vbo_so3_exp(const double *omega, double *q) {
    const double theta = v3_norm(omega);
    const double half = 0.5 * theta;
    double imag;
    const double real = cos(half);
    if (theta < 1e-10) {
        const double t2 = theta * theta, t4 = t2 * t2;
        imag = 0.5 - 0.0208333 * t2 + 0.000260417 * t4;
    } else {
        imag = sin(half) / theta;
    }
    q[3] = real; q[0] = imag * omega[0]; q[1] = imag * omega[1]; q[2] = imag * omega[2];
    quat_normalize(q);
}
/* SO3::logAndTheta, so3.cpp:190-228.  The |w|<eps branch is dead (missing else, :211-222). */
void vbo_so3_log(const double *q, double *omega) {
    const double n = v3_norm(q);
    const double w = q[3];
    double f;
    if (n < 1e-10) f = 2. / w - 2. * (n * n) / (w * w * w);
    else f = 2 * atan(n / w) / n;
    omega[0] = f * q[0]; omega[1] = f * q[1]; omega[2] = f * q[2];
}
/* SO3::JacobianR, so3.cpp:33-50 */
void vbo_so3_jr(const double *w, double *J) {
    m3_id(J);
    const double theta = v3_norm(w);
    if (theta < 0.00001) return;
    const double k[3] = {w[0] / theta, w[1] / theta, w[2] / theta};
    double K[9], K2[9];
    hat(k, K);
    m3_mul(K, K, K2);
    const double a = (1 - cos(theta)) / theta, b = 1 - sin(theta) / theta;
    for (int i = 0; i < 9; i++) J[i] = J[i] - a * K[i] + b * K2[i];
}
/* SO3::JacobianRInv, so3.cpp:53-72 */
void vbo_so3_jrinv(const double *w, double *J) {
    m3_id(J);
    const double theta = v3_norm(w);
    if (theta < 0.00001) return;
    const double k[3] = {w[0] / theta, w[1] / theta, w[2] / theta};
    double K[9], K2[9], W[9];
    hat(k, K);
    hat(w, W);
    m3_mul(K, K, K2);
    const double c = 1.0 - (1.0 + cos(theta)) * theta / (2.0 * sin(theta));
    for (int i = 0; i < 9; i++) J[i] = J[i] + 0.5 * W[i] + c * K2[i];
}
/* SO3::operator*, so3.cpp:127-133 (copy-constructor normalises :93-96, product normalised) */
static void so3_mul(const double *a, const double *b, double *o) {
    double t[4] = {a[0], a[1], a[2], a[3]};
    quat_normalize(t);
    quat_mul(t, b, t);
    quat_normalize(t);
    memcpy(o, t, sizeof t);
}
/* SO3::inverse, so3.cpp:149-152 */
static void so3_inv(const double *a, double *o) { quat_conj(a, o); quat_normalize(o); }

/* ------------------------------------------------------------------------------------------------
 * A5  g2o SE3Quat restated (Thirdparty/g2o/g2o/types/se3quat.h)
 * ---------------------------------------------------------------------------------------------- */
/* SE3Quat::normalizeRotation, se3quat.h:273-278 */
static void se3_normrot(double *q) {
    if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    quat_normalize(q);
}
/* SE3Quat::exp, se3quat.h:223-257; update = [omega, upsilon]; out = t(3) q(4) */
void vbo_se3_exp(const double *upd, double *out7) {
    const double *omega = upd, *ups = upd + 3;
    const double theta = v3_norm(omega);
    double Om[9], Om2[9], R[9], V[9];
    hat(omega, Om);
    m3_mul(Om, Om, Om2);
    if (theta < 0.00001) {
        for (int i = 0; i < 9; i++) R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + Om[i] + Om2[i];
        memcpy(V, R, sizeof R);
    } else {
        const double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta);
        const double c = (theta - sin(theta)) / pow(theta, 3);
        for (int i = 0; i < 9; i++) {
            const double I = (i % 4 == 0) ? 1.0 : 0.0;
            R[i] = I + a * Om[i] + b * Om2[i];
            V[i] = I + b * Om[i] + c * Om2[i];
        }
    }
    R_to_quat(R, out7 + 3);
    m3_vec(V, ups, out7);
    se3_normrot(out7 + 3); /* SE3Quat(q,t) constructor, se3quat.h:59-61 */
}
/* SE3Quat::operator*, se3quat.h:103-109: out = a * b */
static void se3_mul(const double *a, const double *b, double *out7) {
    double t[3], q[4];
    quat_rot(a + 3, b, t);
    t[0] += a[0]; t[1] += a[1]; t[2] += a[2];
    quat_mul(a + 3, b + 3, q);
    se3_normrot(q);
    out7[0] = t[0]; out7[1] = t[1]; out7[2] = t[2];
    memcpy(out7 + 3, q, sizeof q);
}

/* ------------------------------------------------------------------------------------------------
 * A9  Huber kernel, Thirdparty/g2o/g2o/core/robust_kernel_impl.cpp:78-91
 * ---------------------------------------------------------------------------------------------- */
void vbo_huber(double e, double delta, double *rho) {
    const double dsqr = delta * delta;
    if (e <= dsqr) {
        rho[0] = e; rho[1] = 1.; rho[2] = 0.;
    } else {
        const double sqrte = sqrt(e);
        rho[0] = 2 * sqrte * delta - dsqr;
        rho[1] = delta / sqrte;
        rho[2] = -0.5 * rho[1] / e;
    }
}

/* ------------------------------------------------------------------------------------------------
 * A1  EdgePRIDP, src/IMU/g2otypes.cpp:17-158
 *   pt3 = rho,xbar,ybar; ref7 / obs7 = NavState P,q of reference / observing KF; Tcb7 = t_cb,q_cb
 *   out: e(2) = z - K pi(P_i); Pc(3); Jrho(2), Jref(2x6 row-major), Jobs(2x6) when Jrho != NULL
 * ---------------------------------------------------------------------------------------------- */
void vbo_edge_idp(const double *pt3, const double *ref7, const double *obs7, const double *Tcb7, const double *K,
                  const double *uv, double *e, double *Pc, double *Jrho, double *Jref, double *Jobs) {
    double R0[9], Ri[9], Rcb[9], RiT[9], RcbT[9];
    quat_to_R(ref7 + 3, R0);
    quat_to_R(obs7 + 3, Ri);
    quat_to_R(Tcb7 + 3, Rcb);
    m3_T(Ri, RiT);
    m3_T(Rcb, RcbT);
    const double *t0 = ref7, *ti = obs7, *tcb = Tcb7;
    double rho = pt3[0];
    if (rho < 1e-6) rho = 1e-6; /* g2otypes.cpp:42-47 / :86-91 */
    const double d = 1.0 / rho;
    const double P0[3] = {pt3[1] * d, pt3[2] * d, d};
    /* Rcic0 = Rcb Ri^T R0 Rcb^T ; Pi = Rcic0 P0 + tcb - Rcic0 tcb + Rcb Ri^T (t0 - ti)   :63-64 */
    double RcbRiT[9], Rcic0[9], tmp[9];
    m3_mul(Rcb, RiT, RcbRiT);
    m3_mul(RcbRiT, R0, tmp);
    m3_mul(tmp, RcbT, Rcic0);
    double a[3], b[3], c[3];
    m3_vec(Rcic0, P0, a);
    m3_vec(Rcic0, tcb, b);
    const double dt[3] = {t0[0] - ti[0], t0[1] - ti[1], t0[2] - ti[2]};
    m3_vec(RcbRiT, dt, c);
    const double Pi[3] = {a[0] + tcb[0] - b[0] + c[0], a[1] + tcb[1] - b[1] + c[1], a[2] + tcb[2] - b[2] + c[2]};
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
    e[0] = uv[0] - (Pi[0] / Pi[2] * fx + cx); /* cam_project, g2otypes.h:102-120 */
    e[1] = uv[1] - (Pi[1] / Pi[2] * fy + cy);
    if (Pc) { Pc[0] = Pi[0]; Pc[1] = Pi[1]; Pc[2] = Pi[2]; }
    if (!Jrho) return;
    const double x = Pi[0], y = Pi[1], z = Pi[2];
    /* Jpi = Maux / z, :112-121 */
    const double Jpi[6] = {fx / z, 0, -x / z * fx / z, 0, fy / z, -y / z * fy / z};
    /* vertex 0: J_pi_rho = Rcic0 * (-d * P0), :124-125 */
    const double mdP0[3] = {-d * P0[0], -d * P0[1], -d * P0[2]};
    double Jpr[3];
    m3_vec(Rcic0, mdP0, Jpr);
    Jrho[0] = -(Jpi[0] * Jpr[0] + Jpi[1] * Jpr[1] + Jpi[2] * Jpr[2]);
    Jrho[1] = -(Jpi[3] * Jpr[0] + Jpi[4] * Jpr[1] + Jpi[5] * Jpr[2]);
    /* vertex 1: [Rcb RiT | -Rcic0 hat(P0 - tcb) Rcb], :128-134 */
    const double P0mt[3] = {P0[0] - tcb[0], P0[1] - tcb[1], P0[2] - tcb[2]};
    double H0[9], Jr0[9];
    hat(P0mt, H0);
    m3_mul(Rcic0, H0, tmp);
    m3_mul(tmp, Rcb, Jr0);
    for (int r = 0; r < 2; r++)
        for (int cc = 0; cc < 3; cc++) {
            double s1 = 0, s2 = 0;
            for (int k = 0; k < 3; k++) {
                s1 += Jpi[3 * r + k] * RcbRiT[3 * k + cc];
                s2 += Jpi[3 * r + k] * (-Jr0[3 * k + cc]);
            }
            Jref[6 * r + cc] = -s1;
            Jref[6 * r + 3 + cc] = -s2;
        }
    /* vertex 2: [-Rcb RiT | Rcb hat(RiT (R0 Rcb^T (P0 - tcb) + t0 - ti))], :139-145 */
    double u[3], taux[3], Hi[9], Jri[9];
    m3_mul(R0, RcbT, tmp);
    m3_vec(tmp, P0mt, u);
    u[0] += dt[0]; u[1] += dt[1]; u[2] += dt[2];
    m3_vec(RiT, u, taux);
    hat(taux, Hi);
    m3_mul(Rcb, Hi, Jri);
    for (int r = 0; r < 2; r++)
        for (int cc = 0; cc < 3; cc++) {
            double s1 = 0, s2 = 0;
            for (int k = 0; k < 3; k++) {
                s1 += Jpi[3 * r + k] * (-RcbRiT[3 * k + cc]);
                s2 += Jpi[3 * r + k] * Jri[3 * k + cc];
            }
            Jobs[6 * r + cc] = -s1;
            Jobs[6 * r + 3 + cc] = -s2;
        }
}

/* ------------------------------------------------------------------------------------------------
 * A4  EdgeNavStatePRPointXYZ, src/IMU/g2otypes.h:275-308, g2otypes.cpp:371-420
 *   Rcb, tcb come from T_cb (the reference holds Rbc,Pbc: Rcb = Rbc^T, -Rcb*Pbc = t_cb)
 * ---------------------------------------------------------------------------------------------- */
void vbo_edge_prxyz(const double *Pw, const double *kf7, const double *Tcb7, const double *K, const double *uv,
                    double *e, double *Pc_out, double *Jp, double *Jkf) {
    double Rwb[9], RwbT[9], Rcb[9], M[9];
    quat_to_R(kf7 + 3, Rwb);
    m3_T(Rwb, RwbT);
    quat_to_R(Tcb7 + 3, Rcb);
    m3_mul(Rcb, RwbT, M);
    const double dP[3] = {Pw[0] - kf7[0], Pw[1] - kf7[1], Pw[2] - kf7[2]};
    double Paux[3];
    m3_vec(M, dP, Paux);
    const double Pc[3] = {Paux[0] + Tcb7[0], Paux[1] + Tcb7[1], Paux[2] + Tcb7[2]};
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
    e[0] = uv[0] - (Pc[0] / Pc[2] * fx + cx);
    e[1] = uv[1] - (Pc[1] / Pc[2] * fy + cy);
    if (Pc_out) { Pc_out[0] = Pc[0]; Pc_out[1] = Pc[1]; Pc_out[2] = Pc[2]; }
    if (!Jp) return;
    const double x = Pc[0], y = Pc[1], z = Pc[2];
    const double Jpi[6] = {fx / z, 0, -x / z * fx / z, 0, fy / z, -y / z * fy / z};
    double Hh[9], HR[9];
    hat(Paux, Hh);
    m3_mul(Hh, Rcb, HR);
    for (int r = 0; r < 2; r++)
        for (int c = 0; c < 3; c++) {
            double s1 = 0, s2 = 0;
            for (int k = 0; k < 3; k++) {
                s1 += Jpi[3 * r + k] * M[3 * k + c];
                s2 += Jpi[3 * r + k] * HR[3 * k + c];
            }
            Jp[3 * r + c] = -s1;       /* :406 */
            Jkf[6 * r + c] = s1;       /* :409  -Jpi * (-Rcb RwbT) */
            Jkf[6 * r + 3 + c] = -s2;  /* :412 */
        }
}

/* ------------------------------------------------------------------------------------------------
 * A5  EdgeSE3ProjectXYZ, Thirdparty/g2o/g2o/types/types_six_dof_expmap.h:80-109, .cpp:103-147
 * ---------------------------------------------------------------------------------------------- */
void vbo_edge_se3xyz(const double *Pw, const double *T7, const double *K, const double *uv, double *e, double *Pc_out,
                     double *Jp, double *Jkf) {
    double Pc[3];
    quat_rot(T7 + 3, Pw, Pc); /* SE3Quat::map, se3quat.h:217-220 */
    Pc[0] += T7[0]; Pc[1] += T7[1]; Pc[2] += T7[2];
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
    e[0] = uv[0] - (Pc[0] / Pc[2] * fx + cx);
    e[1] = uv[1] - (Pc[1] / Pc[2] * fy + cy);
    if (Pc_out) { Pc_out[0] = Pc[0]; Pc_out[1] = Pc[1]; Pc_out[2] = Pc[2]; }
    if (!Jp) return;
    const double x = Pc[0], y = Pc[1], z = Pc[2], z_2 = z * z;
    const double tmp[6] = {fx, 0, -x / z * fx, 0, fy, -y / z * fy};
    double R[9];
    quat_to_R(T7 + 3, R);
    for (int r = 0; r < 2; r++)
        for (int c = 0; c < 3; c++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += tmp[3 * r + k] * R[3 * k + c];
            Jp[3 * r + c] = -1. / z * s; /* :124 */
        }
    Jkf[0] = x * y / z_2 * fx;
    Jkf[1] = -(1 + (x * x / z_2)) * fx;
    Jkf[2] = y / z * fx;
    Jkf[3] = -1. / z * fx;
    Jkf[4] = 0;
    Jkf[5] = x / z_2 * fx;
    Jkf[6] = (1 + y * y / z_2) * fy;
    Jkf[7] = -x * y / z_2 * fy;
    Jkf[8] = -x / z * fy;
    Jkf[9] = 0;
    Jkf[10] = -1. / z * fy;
    Jkf[11] = y / z_2 * fy;
}

/* ------------------------------------------------------------------------------------------------
 * A2  EdgeNavStatePRV, src/IMU/g2otypes.cpp:163-367
 *   posei/posej: P,q ; veli/velj ; biasi: bg ba dbg dba ; meas: VBA_IMU_MEAS_STRIDE ; g: gravity
 *   err(9) order P, phi, V.  If J != NULL: J[0]=dPRi(9x6) J[1]=dPRj(9x6) J[2]=dVi(9x3) J[3]=dVj(9x3) J[4]=dBi(9x6)
 * ---------------------------------------------------------------------------------------------- */
void vbo_edge_prv_error(const double *posei, const double *posej, const double *veli, const double *velj,
                        const double *biasi, const double *meas, const double *g, double *err) {
    const double dT = meas[0], dT2 = dT * dT;
    const double *dP = meas + 1, *dV = meas + 4, *dRm = meas + 7;
    const double *JPg = meas + 16, *JPa = meas + 25, *JVg = meas + 34, *JVa = meas + 43, *JRg = meas + 52;
    const double *dbg = biasi + 6, *dba = biasi + 9;
    double dRq[4], RiTq[4];
    R_to_quat(dRm, dRq); /* Sophus::SO3(Matrix3d), so3.cpp:99-102 */
    quat_normalize(dRq);
    so3_inv(posei + 3, RiTq); /* :203 */
    double v[3], rv[3], c1[3], c2[3];
    for (int k = 0; k < 3; k++) v[k] = posej[k] - posei[k] - veli[k] * dT - 0.5 * g[k] * dT2;
    quat_rot(RiTq, v, rv);
    m3_vec(JPg, dbg, c1);
    m3_vec(JPa, dba, c2);
    for (int k = 0; k < 3; k++) err[k] = rv[k] - (dP[k] + c1[k] + c2[k]); /* :207-208 */
    for (int k = 0; k < 3; k++) v[k] = velj[k] - veli[k] - g[k] * dT;
    quat_rot(RiTq, v, rv);
    m3_vec(JVg, dbg, c1);
    m3_vec(JVa, dba, c2);
    for (int k = 0; k < 3; k++) err[6 + k] = rv[k] - (dV[k] + c1[k] + c2[k]); /* :210-211 */
    double w[3], dRdbg[4], A[4], Ainv[4], B[4], C[4];
    m3_vec(JRg, dbg, w);
    vbo_so3_exp(w, dRdbg);      /* :213 */
    so3_mul(dRq, dRdbg, A);     /* dRij * dR_dbg */
    so3_inv(A, Ainv);
    so3_mul(Ainv, RiTq, B);
    so3_mul(B, posej + 3, C);   /* :214 */
    vbo_so3_log(C, err + 3);    /* :215 */
}

void vbo_edge_prv_jac(const double *posei, const double *posej, const double *veli, const double *velj,
                      const double *biasi, const double *meas, const double *g, const double *err, double *JPRi,
                      double *JPRj, double *JVi, double *JVj, double *JBi) {
    const double dT = meas[0], dT2 = dT * dT;
    const double *JPg = meas + 16, *JPa = meas + 25, *JVg = meas + 34, *JVa = meas + 43, *JRg = meas + 52;
    const double *dbg = biasi + 6;
    double Ri[9], Rj[9], RiT[9], RjT[9];
    quat_to_R(posei + 3, Ri);
    quat_to_R(posej + 3, Rj);
    m3_T(Ri, RiT);
    m3_T(Rj, RjT);
    const double *rPhi = err + 3; /* :265 */
    double JrInv[9];
    vbo_so3_jrinv(rPhi, JrInv);
    memset(JPRi, 0, 54 * sizeof(double));
    memset(JPRj, 0, 54 * sizeof(double));
    memset(JVi, 0, 27 * sizeof(double));
    memset(JVj, 0, 27 * sizeof(double));
    memset(JBi, 0, 54 * sizeof(double));
    double v[3], rv[3], H[9], T1[9], T2[9];
#define SETB(J, ld, r0, c0, M, sgn)                                                    \
    for (int _r = 0; _r < 3; _r++)                                                      \
        for (int _c = 0; _c < 3; _c++) (J)[((r0) + _r) * (ld) + (c0) + _c] = (sgn) * (M)[3 * _r + _c];
    /* JPRi :300-312 */
    SETB(JPRi, 6, 0, 0, RiT, -1.0);
    for (int k = 0; k < 3; k++) v[k] = posej[k] - posei[k] - veli[k] * dT - 0.5 * g[k] * dT2;
    m3_vec(RiT, v, rv);
    hat(rv, H);
    SETB(JPRi, 6, 0, 3, H, 1.0);
    m3_mul(JrInv, RjT, T1);
    m3_mul(T1, Ri, T2);
    SETB(JPRi, 6, 3, 3, T2, -1.0);
    for (int k = 0; k < 3; k++) v[k] = velj[k] - veli[k] - g[k] * dT;
    m3_vec(RiT, v, rv);
    hat(rv, H);
    SETB(JPRi, 6, 6, 3, H, 1.0);
    /* JVi :315-319 */
    SETB(JVi, 3, 0, 0, RiT, -dT);
    SETB(JVi, 3, 6, 0, RiT, -1.0);
    /* JPRj :323-336 */
    SETB(JPRj, 6, 0, 0, RiT, 1.0);
    SETB(JPRj, 6, 3, 3, JrInv, 1.0);
    /* JVj :339-343 */
    SETB(JVj, 3, 6, 0, RiT, 1.0);
    /* JBiasi :347-359 */
    SETB(JBi, 6, 0, 0, JPg, -1.0);
    SETB(JBi, 6, 0, 3, JPa, -1.0);
    double q[4], qi[4], ExpT[9], w[3], JrB[9];
    vbo_so3_exp(rPhi, q);
    so3_inv(q, qi);
    quat_to_R(qi, ExpT); /* :353 */
    m3_vec(JRg, dbg, w);
    vbo_so3_jr(w, JrB); /* :354 */
    m3_mul(JrInv, ExpT, T1);
    m3_mul(T1, JrB, T2);
    m3_mul(T2, JRg, T1);
    SETB(JBi, 6, 3, 0, T1, -1.0);
    SETB(JBi, 6, 6, 0, JVg, -1.0);
    SETB(JBi, 6, 6, 3, JVa, -1.0);
#undef SETB
}

/* A3  EdgeNavStateBias::computeError, src/IMU/g2otypes.cpp:703-726 */
void vbo_edge_bias_error(const double *biasi, const double *biasj, double *err) {
    for (int k = 0; k < 3; k++) {
        err[k] = (biasj[k] + biasj[6 + k]) - (biasi[k] + biasi[6 + k]);
        err[3 + k] = (biasj[3 + k] + biasj[9 + k]) - (biasi[3 + k] + biasi[9 + k]);
    }
}

/* ------------------------------------------------------------------------------------------------
 * A6  EdgeNavState (15-D, two VertexNavState), src/IMU/g2otypes.cpp:989-1168, and VertexNavState::oplusImpl
 *     (:961-966 -> NavState::IncSmall, src/IMU/NavState.cpp:31-59).  The reference no longer instantiates this
 *     edge (LocalMapping calls the PR/V/Bias-split optimisers); it is restated so that the split factor the backend
 *     fuses (A2 EdgeNavStatePRV + A3 EdgeNavStateBias) can be checked against the 15-D factor it was derived from
 *     (SURVEY 8c item 5).  Written from the cited lines, sharing no code with A2/A3 above.
 *   nav (22): P(3) q(4, xyzw) V(3) bg(3) ba(3) dbg(3) dba(3)        src/IMU/NavState.h:124-138
 *   err (15): rP, rV, rPhi, rBiasG, rBiasA                          :1040-1046
 *   Ji, Jj (15x15 row-major), columns P V Phi dBg dBa               :1085-1166
 * ---------------------------------------------------------------------------------------------- */
void vbo_edge_navstate_error(const double *navi, const double *navj, const double *meas, const double *g, double *err) {
    const double *Pi = navi, *Vi = navi + 7, *dBgi = navi + 16, *dBai = navi + 19;   /* :995-1000 */
    const double *Pj = navj, *Vj = navj + 7;                                        /* :1003-1006 */
    const double dTij = meas[0], dT2 = dTij * dTij;                                 /* :1010-1011 */
    const double *dPij = meas + 1, *dVij = meas + 4;
    double dRij[4];
    R_to_quat(meas + 7, dRij);   /* Sophus::SO3(M.getDeltaR()), :1014 */
    quat_normalize(dRij);
    double RiT[4];
    so3_inv(navi + 3, RiT);      /* :1016 */
    double a[3], ra[3], t1[3], t2[3];
    for (int k = 0; k < 3; k++) a[k] = Pj[k] - Pi[k] - Vi[k] * dTij - 0.5 * g[k] * dT2;
    quat_rot(RiT, a, ra);
    m3_vec(meas + 16, dBgi, t1);   /* JPBiasg */
    m3_vec(meas + 25, dBai, t2);   /* JPBiasa */
    for (int k = 0; k < 3; k++) err[k] = ra[k] - (dPij[k] + t1[k] + t2[k]);       /* rPij :1020-1021 */
    for (int k = 0; k < 3; k++) a[k] = Vj[k] - Vi[k] - g[k] * dTij;
    quat_rot(RiT, a, ra);
    m3_vec(meas + 34, dBgi, t1);   /* JVBiasg */
    m3_vec(meas + 43, dBai, t2);   /* JVBiasa */
    for (int k = 0; k < 3; k++) err[3 + k] = ra[k] - (dVij[k] + t1[k] + t2[k]);   /* rVij :1024-1025 */
    double w[3], dR_dbg[4], prod[4], inv[4], u[4], rR[4];
    m3_vec(meas + 52, dBgi, w);    /* JRBiasg dBgi */
    vbo_so3_exp(w, dR_dbg);        /* :1028 */
    so3_mul(dRij, dR_dbg, prod);
    so3_inv(prod, inv);
    so3_mul(inv, RiT, u);
    so3_mul(u, navj + 3, rR);      /* :1029 */
    vbo_so3_log(rR, err + 6);      /* :1030 */
    for (int k = 0; k < 3; k++) {
        err[9 + k] = (navj[10 + k] + navj[16 + k]) - (navi[10 + k] + navi[16 + k]);    /* rBiasG :1033-1034 */
        err[12 + k] = (navj[13 + k] + navj[19 + k]) - (navi[13 + k] + navi[19 + k]);   /* rBiasA :1037-1038 */
    }
}

void vbo_edge_navstate_jac(const double *navi, const double *navj, const double *meas, const double *g, const double *err,
                           double *Ji, double *Jj) {
    const double *Pi = navi, *Vi = navi + 7, *dBgi = navi + 16;   /* :1062-1066 */
    const double *Pj = navj, *Vj = navj + 7;                      /* :1069-1072 */
    double Ri[9], Rj[9], RiT[9], RjT[9];
    quat_to_R(navi + 3, Ri);
    quat_to_R(navj + 3, Rj);
    m3_T(Ri, RiT);
    m3_T(Rj, RjT);
    const double dTij = meas[0], dT2 = dTij * dTij;   /* :1076-1077 */
    const double *rPhiij = err + 6;                   /* :1083 */
    double JrInv_rPhi[9];
    vbo_so3_jrinv(rPhiij, JrInv_rPhi);                /* :1084 */
    const double *J_rPhi_dbg = meas + 52;             /* :1085 */
    memset(Ji, 0, 225 * sizeof(double));
    memset(Jj, 0, 225 * sizeof(double));
#define BLK(J, r0, c0, M, sgn)                                                          \
    for (int _r = 0; _r < 3; _r++)                                                      \
        for (int _c = 0; _c < 3; _c++) (J)[((r0) + _r) * 15 + (c0) + _c] = (sgn) * (M)[3 * _r + _c];
    double a[3], ra[3], H[9], T1[9], T2[9];
    /* vertex 0: rows rP :1091-1096 */
    BLK(Ji, 0, 0, RiT, -1.0);
    BLK(Ji, 0, 3, RiT, -dTij);
    for (int k = 0; k < 3; k++) a[k] = Pj[k] - Pi[k] - Vi[k] * dTij - 0.5 * g[k] * dT2;
    m3_vec(RiT, a, ra);
    hat(ra, H);
    BLK(Ji, 0, 6, H, 1.0);
    BLK(Ji, 0, 9, meas + 16, -1.0);
    BLK(Ji, 0, 12, meas + 25, -1.0);
    /* rows rV :1099-1103 */
    BLK(Ji, 3, 3, RiT, -1.0);
    for (int k = 0; k < 3; k++) a[k] = Vj[k] - Vi[k] - g[k] * dTij;
    m3_vec(RiT, a, ra);
    hat(ra, H);
    BLK(Ji, 3, 6, H, 1.0);
    BLK(Ji, 3, 9, meas + 34, -1.0);
    BLK(Ji, 3, 12, meas + 43, -1.0);
    /* rows rPhi :1106-1112 */
    double qe[4], qei[4], ExprPhiijTrans[9], w[3], JrBiasGCorr[9];
    vbo_so3_exp(rPhiij, qe);
    so3_inv(qe, qei);
    quat_to_R(qei, ExprPhiijTrans);
    m3_vec(J_rPhi_dbg, dBgi, w);
    vbo_so3_jr(w, JrBiasGCorr);
    m3_mul(JrInv_rPhi, RjT, T1);
    m3_mul(T1, Ri, T2);
    BLK(Ji, 6, 6, T2, -1.0);
    m3_mul(JrInv_rPhi, ExprPhiijTrans, T1);
    m3_mul(T1, JrBiasGCorr, T2);
    m3_mul(T2, J_rPhi_dbg, T1);
    BLK(Ji, 6, 9, T1, -1.0);
    /* rows rBiasG, rBiasA :1115-1126 */
    for (int k = 0; k < 3; k++) { Ji[(9 + k) * 15 + 9 + k] = -1.0; Ji[(12 + k) * 15 + 12 + k] = -1.0; }
    /* vertex 1 :1131-1166 */
    BLK(Jj, 0, 0, RiT, 1.0);
    BLK(Jj, 3, 3, RiT, 1.0);
    BLK(Jj, 6, 6, JrInv_rPhi, 1.0);
    for (int k = 0; k < 3; k++) { Jj[(9 + k) * 15 + 9 + k] = 1.0; Jj[(12 + k) * 15 + 12 + k] = 1.0; }
#undef BLK
}

/* VertexNavState::oplusImpl, g2otypes.cpp:961-966 -> NavState::IncSmall, NavState.cpp:31-59 (update order P V Phi dBg dBa) */
void vbo_oplus_navstate(double *nav, const double *upd) {
    for (int k = 0; k < 3; k++) { nav[k] += upd[k]; nav[7 + k] += upd[3 + k]; }
    double dR[4];
    vbo_so3_exp(upd + 6, dR);
    so3_mul(nav + 3, dR, nav + 3);   /* _R = Get_R() * dR */
    for (int k = 0; k < 3; k++) { nav[16 + k] += upd[9 + k]; nav[19 + k] += upd[12 + k]; }
}

/* ------------------------------------------------------------------------------------------------
 * A15  IMUPreintegrator::update, src/IMU/IMUPreintegrator.cpp:63-112
 *   state: meas[VBA_IMU_MEAS_STRIDE] (dt,dP,dV,dR,JPg,JPa,JVg,JVa,JRg) + cov[81] in P,V,phi order
 *   gyr_cov / acc_cov: scalar diagonal of IMUData::_gyrMeasCov / _accMeasCov (imudata.cpp:28-31)
 * ---------------------------------------------------------------------------------------------- */
void vbo_preint_reset(double *meas, double *cov) {
    memset(meas, 0, VBA_IMU_MEAS_STRIDE * sizeof(double));
    meas[7] = meas[11] = meas[15] = 1.0;
    memset(cov, 0, 81 * sizeof(double));
}
void vbo_preint_update(double *meas, double *cov, const double *omega, const double *acc, double dt, double gyr_cov,
                       double acc_cov) {
    double *dP = meas + 1, *dV = meas + 4, *dR = meas + 7;
    double *JPg = meas + 16, *JPa = meas + 25, *JVg = meas + 34, *JVa = meas + 43, *JRg = meas + 52;
    const double dt2 = dt * dt;
    double wdt[3] = {omega[0] * dt, omega[1] * dt, omega[2] * dt};
    double q[4], dRk[9], dRkT[9], Jr[9];
    vbo_so3_exp(wdt, q);
    quat_to_R(q, dRk); /* Expmap, IMUPreintegrator.h:93-96 */
    m3_T(dRk, dRkT);
    vbo_so3_jr(wdt, Jr); /* IMUPreintegrator.h:102-119 is the same formula as so3.cpp:33-50 */
    double Sa[9], RS[9];
    hat(acc, Sa);
    m3_mul(dR, Sa, RS);
    /* A, Bg, Ca :75-90 (block order P,V,phi) */
    double A[81];
    memset(A, 0, sizeof A);
    for (int i = 0; i < 9; i++) A[10 * i] = 1.0;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            A[(6 + r) * 9 + 6 + c] = dRkT[3 * r + c];
            A[(3 + r) * 9 + 6 + c] = -RS[3 * r + c] * dt;
            A[(0 + r) * 9 + 6 + c] = -0.5 * RS[3 * r + c] * dt2;
            A[(0 + r) * 9 + 3 + c] = (r == c) ? dt : 0.0;
        }
    double Bg[27], Ca[27];
    memset(Bg, 0, sizeof Bg);
    memset(Ca, 0, sizeof Ca);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            Bg[(6 + r) * 3 + c] = Jr[3 * r + c] * dt;
            Ca[(3 + r) * 3 + c] = dR[3 * r + c] * dt;
            Ca[(0 + r) * 3 + c] = 0.5 * dR[3 * r + c] * dt2;
        }
    double AC[81], N[81];
    for (int i = 0; i < 9; i++)
        for (int j = 0; j < 9; j++) {
            double s = 0;
            for (int k = 0; k < 9; k++) s += A[9 * i + k] * cov[9 * k + j];
            AC[9 * i + j] = s;
        }
    for (int i = 0; i < 9; i++)
        for (int j = 0; j < 9; j++) {
            double s = 0;
            for (int k = 0; k < 9; k++) s += AC[9 * i + k] * A[9 * j + k];
            double sg = 0, sa = 0;
            for (int k = 0; k < 3; k++) {
                sg += Bg[3 * i + k] * gyr_cov * Bg[3 * j + k];
                sa += Ca[3 * i + k] * acc_cov * Ca[3 * j + k];
            }
            N[9 * i + j] = s + sg + sa;
        }
    memcpy(cov, N, sizeof N);
    /* bias Jacobians :98-102 (each line uses the not-yet-updated values of the ones below it) */
    double RSJ[9], T[9];
    m3_mul(RS, JRg, RSJ);
    for (int i = 0; i < 9; i++) JPa[i] += JVa[i] * dt - 0.5 * dR[i] * dt2;
    for (int i = 0; i < 9; i++) JPg[i] += JVg[i] * dt - 0.5 * RSJ[i] * dt2;
    for (int i = 0; i < 9; i++) JVa[i] += -dR[i] * dt;
    for (int i = 0; i < 9; i++) JVg[i] += -RSJ[i] * dt;
    m3_mul(dRkT, JRg, T);
    for (int i = 0; i < 9; i++) JRg[i] = T[i] - Jr[i] * dt;
    /* deltas :106-110 */
    double Ra[3];
    m3_vec(dR, acc, Ra);
    for (int k = 0; k < 3; k++) dP[k] += dV[k] * dt + 0.5 * Ra[k] * dt2;
    for (int k = 0; k < 3; k++) dV[k] += Ra[k] * dt;
    m3_mul(dR, dRk, T);
    double qn[4];
    R_to_quat(T, qn); /* normalizeRotationM, IMUPreintegrator.h:163-174 */
    if (qn[3] < 0) { qn[0] = -qn[0]; qn[1] = -qn[1]; qn[2] = -qn[2]; qn[3] = -qn[3]; }
    quat_normalize(qn);
    quat_to_R(qn, dR);
    meas[0] += dt;
}

/* symmetric positive definite 9x9 (general) inverse by Gauss-Jordan with partial pivoting:
 * stands in for Eigen's CovPRV.inverse() (src/Optimizer.cpp:280).  Also swaps V/phi (:274-279). */
int vbo_prv_information(const double *cov_pvphi, double *info_pphiv) {
    static const int perm[9] = {0, 1, 2, 6, 7, 8, 3, 4, 5};
    double M[9][18];
    for (int i = 0; i < 9; i++) {
        for (int j = 0; j < 9; j++) M[i][j] = cov_pvphi[9 * perm[i] + perm[j]];
        for (int j = 0; j < 9; j++) M[i][9 + j] = (i == j) ? 1.0 : 0.0;
    }
    for (int c = 0; c < 9; c++) {
        int p = c;
        for (int r = c + 1; r < 9; r++)
            if (fabs(M[r][c]) > fabs(M[p][c])) p = r;
        if (M[p][c] == 0.0) return -1;
        if (p != c)
            for (int j = 0; j < 18; j++) { double t = M[c][j]; M[c][j] = M[p][j]; M[p][j] = t; }
        const double inv = 1.0 / M[c][c];
        for (int j = 0; j < 18; j++) M[c][j] *= inv;
        for (int r = 0; r < 9; r++) {
            if (r == c) continue;
            const double f = M[r][c];
            if (f == 0.0) continue;
            for (int j = 0; j < 18; j++) M[r][j] -= f * M[c][j];
        }
    }
    for (int i = 0; i < 9; i++)
        for (int j = 0; j < 9; j++) info_pphiv[9 * i + j] = M[i][9 + j];
    return 0;
}

/* ================================================================================================
 * The optimiser (A8-A14): g2o SparseOptimizer + BlockSolver + GN/LM restated on dense storage
 * ============================================================================================== */
typedef struct {
    vba_problem *P;
    int variant, nkf, nfree, npt, nobs, nimu;
    int pdim, np, ldim, nl; /* pose dofs per KF, total pose dofs, landmark dim, total landmark dofs */
    double *pose, *vel, *bias, *pt; /* working state (copies) */
    double *pose_bk, *vel_bk, *bias_bk, *pt_bk;
    unsigned char *lvl;     /* [nobs] g2o edge level */
    int vis_robust;         /* Huber on vision edges (stage 1) */
    double *err;            /* [nobs][2] _error of every vision edge (stale for inactive ones) */
    double *imu_err;        /* [nimu][9] */
    double *bias_err;       /* [nimu][6] */
    double *Hpp, *b;        /* [np*np] full symmetric, b = [np + nl] */
    double *Hll;            /* [npt][ldim*ldim] */
    double *Wobs;           /* [nobs][6*ldim] H_pl block (pose rows x landmark cols) of the observing KF */
    double *Wref;           /* [npt][6] variant 2: H_pl block of the reference KF */
    double *Dinv;           /* [npt][ldim*ldim] */
    double *S, *bs, *x;     /* reduced system, x = [np + nl] */
    unsigned char *pt_act;  /* [npt] landmark in the active set */
    unsigned char *var_act; /* [np] pose scalar variable belongs to an active vertex */
    const unsigned char *fix; /* P->kf_fix or NULL: per-vertex setFixed() of listed-free keyframes */
    int imu_robust;         /* Huber on the PRV / bias edges (always, except bRobust = false in global BA) */
    int polls, stop_after;  /* test hook (vba_oracle_solve_ex): terminate() polls so far; >= 0: the flag reads 1 from that poll on */
    int solver_perm;        /* 1: order V/Bias blocks first for the skyline LDLT */
    int *perm;              /* [np] */
    double *Lwork;          /* [np*np] */
    int *first;             /* [np] */
    double *ywork;
} ctx;

/* first column of vertex `part` (0 PR, 1 V, 2 Bias) of keyframe kf in H_pp, or -1 if that vertex is fixed */
static int vcol(const ctx *c, int kf, int part) {
    static const int off[3] = {0, 6, 9};
    if (kf >= c->nfree) return -1;
    if (c->fix && ((c->fix[kf] >> part) & 1)) return -1;
    return kf * c->pdim + off[part];
}
static int kf_col(const ctx *c, int kf) { return vcol(c, kf, 0); }

static double chi2_2(const double *e, double w) { return e[0] * (w * e[0]) + e[1] * (w * e[1]); }

static double quadform(const double *e, const double *Om, int d) {
    double s = 0;
    for (int i = 0; i < d; i++) {
        double t = 0;
        for (int j = 0; j < d; j++) t += Om[d * i + j] * e[j];
        s += e[i] * t;
    }
    return s;
}

/* allVerticesFixed -> the edge is dropped from the active set, sparse_optimizer.cpp:236.  bit0: EdgeNavStatePRV
 * (PR_i, PR_j, V_i, V_j, Bias_i), bit1: EdgeNavStateBias (Bias_i, Bias_j) */
static int imu_edge_active(const ctx *c, int k) {
    const int i = c->P->imu_kf_i[k], j = c->P->imu_kf_j[k];
    const int prv = vcol(c, i, 0) >= 0 || vcol(c, j, 0) >= 0 || vcol(c, i, 1) >= 0 || vcol(c, j, 1) >= 0 || vcol(c, i, 2) >= 0;
    const int bias = vcol(c, i, 2) >= 0 || vcol(c, j, 2) >= 0;
    return prv | (bias << 1);
}
static void huber_or_not(int on, double s, double delta, double *rho) {
    if (on) vbo_huber(s, delta, rho);
    else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
}

/* vision residual of observation o of point p at the working state; returns camera-frame depth */
static double vis_eval(const ctx *c, int p, int o, double *e, double *J0, double *J1, double *J2) {
    const vba_problem *P = c->P;
    const int kf = P->obs_kf[o];
    double Pc[3];
    if (c->variant == VBA_VARIANT_PRV_IDP) {
        const int rf = P->pt_ref_kf[p];
        vbo_edge_idp(c->pt + 3 * p, c->pose + 7 * rf, c->pose + 7 * kf, P->T_cb, P->K, P->obs_uv + 2 * o, e, Pc, J0, J1, J2);
    } else if (c->variant == VBA_VARIANT_PRV_XYZ) {
        vbo_edge_prxyz(c->pt + 3 * p, c->pose + 7 * kf, P->T_cb, P->K, P->obs_uv + 2 * o, e, Pc, J0, J1);
    } else {
        vbo_edge_se3xyz(c->pt + 3 * p, c->pose + 7 * kf, P->K, P->obs_uv + 2 * o, e, Pc, J0, J1);
    }
    return Pc[2];
}

/* SparseOptimizer::initializeOptimization(level), sparse_optimizer.cpp:199-267 + buildIndexMapping :166-190 */
static void init_active(ctx *c) {
    const vba_problem *P = c->P;
    memset(c->pt_act, 0, c->npt);
    memset(c->var_act, 0, c->np);
    for (int p = 0; p < c->npt; p++)
        for (int o = P->pt_obs_begin[p]; o < P->pt_obs_begin[p + 1]; o++) {
            if (c->lvl[o]) continue;
            c->pt_act[p] = 1;
            int col = kf_col(c, P->obs_kf[o]);
            if (col >= 0) memset(c->var_act + col, 1, 6);
            if (c->variant == VBA_VARIANT_PRV_IDP) {
                col = kf_col(c, P->pt_ref_kf[p]);
                if (col >= 0) memset(c->var_act + col, 1, 6);
            }
        }
    for (int k = 0; k < c->nimu; k++) {
        const int act = imu_edge_active(c, k);
        const int i = P->imu_kf_i[k], j = P->imu_kf_j[k];
        static const int dims[3] = {6, 3, 6};
        for (int part = 0; part < 3; part++) {
            const int ci = vcol(c, i, part), cj = vcol(c, j, part);
            /* PRV touches PR_i, PR_j, V_i, V_j, Bias_i; the bias edge Bias_i, Bias_j */
            if (ci >= 0 && ((act & 1) || (part == 2 && (act & 2)))) memset(c->var_act + ci, 1, dims[part]);
            if (cj >= 0 && ((part < 2 && (act & 1)) || (part == 2 && (act & 2)))) memset(c->var_act + cj, 1, dims[part]);
        }
    }
}

/* SparseOptimizer::computeActiveErrors (sparse_optimizer.cpp:61-88) + activeRobustChi2 (:100-114).
 * Edge order N1: IMU edges first (PRV, Bias alternating), then vision edges point by point. */
static double compute_errors(ctx *c) {
    const vba_problem *P = c->P;
    double chi = 0, rho[3];
    for (int k = 0; k < c->nimu; k++) {
        const int act = imu_edge_active(c, k);
        if (!act) continue;
        const int i = P->imu_kf_i[k], j = P->imu_kf_j[k];
        const double *meas = P->imu_meas + VBA_IMU_MEAS_STRIDE * k;
        if (act & 1) {
            vbo_edge_prv_error(c->pose + 7 * i, c->pose + 7 * j, c->vel + 3 * i, c->vel + 3 * j, c->bias + 12 * i, meas, P->g_w,
                               c->imu_err + 9 * k);
            huber_or_not(c->imu_robust, quadform(c->imu_err + 9 * k, P->imu_info_prv + 81 * k, 9), P->huber_prv, rho);
            chi += rho[0];
        }
        if (act & 2) {
            vbo_edge_bias_error(c->bias + 12 * i, c->bias + 12 * j, c->bias_err + 6 * k);
            const double *e = c->bias_err + 6 * k;
            const double wg = P->inv_bg_rw2 / meas[0], wa = P->inv_ba_rw2 / meas[0];
            const double s = wg * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) + wa * (e[3] * e[3] + e[4] * e[4] + e[5] * e[5]);
            huber_or_not(c->imu_robust, s, P->huber_bias, rho);
            chi += rho[0];
        }
    }
    for (int p = 0; p < c->npt; p++)
        for (int o = P->pt_obs_begin[p]; o < P->pt_obs_begin[p + 1]; o++) {
            if (c->lvl[o]) continue;
            vis_eval(c, p, o, c->err + 2 * o, NULL, NULL, NULL);
            const double s = chi2_2(c->err + 2 * o, P->obs_w[o]);
            if (c->vis_robust) {
                vbo_huber(s, P->huber_vis, rho);
                chi += rho[0];
            } else
                chi += s;
        }
    return chi;
}

/* BaseMultiEdge::computeQuadraticForm, base_multi_edge.hpp:171-222 (dense storage, both triangles kept) */
static void accum_pose(ctx *c, int d, const double *Om, const double *omr, int nb, const int *col, const int *dim,
                       const double *const *J) {
    double AtO[15 * 9];
    for (int i = 0; i < nb; i++) {
        if (col[i] < 0) continue;
        const int di = dim[i];
        for (int a = 0; a < di; a++)
            for (int r = 0; r < d; r++) {
                double s = 0;
                for (int k = 0; k < d; k++) s += J[i][k * di + a] * Om[k * d + r];
                AtO[a * d + r] = s;
            }
        for (int a = 0; a < di; a++) {
            for (int bb = 0; bb < di; bb++) {
                double s = 0;
                for (int k = 0; k < d; k++) s += AtO[a * d + k] * J[i][k * di + bb];
                c->Hpp[(size_t)(col[i] + a) * c->np + col[i] + bb] += s;
            }
            double s = 0;
            for (int k = 0; k < d; k++) s += J[i][k * di + a] * omr[k];
            c->b[col[i] + a] += s;
        }
        for (int j = i + 1; j < nb; j++) {
            if (col[j] < 0) continue;
            const int dj = dim[j];
            for (int a = 0; a < di; a++)
                for (int bb = 0; bb < dj; bb++) {
                    double s = 0;
                    for (int k = 0; k < d; k++) s += AtO[a * d + k] * J[j][k * dj + bb];
                    c->Hpp[(size_t)(col[i] + a) * c->np + col[j] + bb] += s;
                    c->Hpp[(size_t)(col[j] + bb) * c->np + col[i] + a] += s;
                }
        }
    }
}

/* BlockSolver::buildSystem, block_solver.hpp:502-560: linearizeOplus + constructQuadraticForm per active edge */
static void build_system(ctx *c) {
    const vba_problem *P = c->P;
    const int L = c->ldim;
    memset(c->Hpp, 0, sizeof(double) * (size_t)c->np * c->np);
    memset(c->b, 0, sizeof(double) * (c->np + c->nl));
    memset(c->Hll, 0, sizeof(double) * (size_t)c->npt * L * L);
    memset(c->Wobs, 0, sizeof(double) * (size_t)c->nobs * 6 * L);
    if (c->Wref) memset(c->Wref, 0, sizeof(double) * (size_t)c->npt * 6);
    double rho[3];
    for (int k = 0; k < c->nimu; k++) {
        const int act = imu_edge_active(c, k);
        if (!act) continue;
        const int i = P->imu_kf_i[k], j = P->imu_kf_j[k];
        const double *meas = P->imu_meas + VBA_IMU_MEAS_STRIDE * k;
        if (act & 1) {   /* EdgeNavStatePRV */
            double J0[54], J1[54], J2[27], J3[27], J4[54], Om[81], omr[9];
            const double *e = c->imu_err + 9 * k;
            vbo_edge_prv_jac(c->pose + 7 * i, c->pose + 7 * j, c->vel + 3 * i, c->vel + 3 * j, c->bias + 12 * i, meas,
                             P->g_w, e, J0, J1, J2, J3, J4);
            const double *info = P->imu_info_prv + 81 * k;
            huber_or_not(c->imu_robust, quadform(e, info, 9), P->huber_prv, rho); /* base_multi_edge.hpp:36-48 */
            for (int a = 0; a < 81; a++) Om[a] = rho[1] * info[a];
            for (int a = 0; a < 9; a++) {
                double s = 0;
                for (int bb = 0; bb < 9; bb++) s += info[9 * a + bb] * e[bb];
                omr[a] = -s * rho[1];
            }
            const int col[5] = {vcol(c, i, 0), vcol(c, j, 0), vcol(c, i, 1), vcol(c, j, 1), vcol(c, i, 2)};
            const int dim[5] = {6, 6, 3, 3, 6};
            const double *const Js[5] = {J0, J1, J2, J3, J4};
            accum_pose(c, 9, Om, omr, 5, col, dim, Js);
        }
        if (act & 2) {   /* EdgeNavStateBias (BaseBinaryEdge, base_binary_edge.hpp:55-120): J = -I, +I */
            const double *e = c->bias_err + 6 * k;
            const double wg = P->inv_bg_rw2 / meas[0], wa = P->inv_ba_rw2 / meas[0];
            const double s = wg * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) + wa * (e[3] * e[3] + e[4] * e[4] + e[5] * e[5]);
            huber_or_not(c->imu_robust, s, P->huber_bias, rho);
            double Om[36], omr[6], Ji[36], Jj[36];
            memset(Om, 0, sizeof Om);
            memset(Ji, 0, sizeof Ji);
            memset(Jj, 0, sizeof Jj);
            for (int a = 0; a < 6; a++) {
                const double w = (a < 3 ? wg : wa);
                Om[7 * a] = rho[1] * w;
                omr[a] = -w * e[a] * rho[1];
                Ji[7 * a] = -1.0;
                Jj[7 * a] = 1.0;
            }
            const int col[2] = {vcol(c, i, 2), vcol(c, j, 2)};
            const int dim[2] = {6, 6};
            const double *const Js[2] = {Ji, Jj};
            accum_pose(c, 6, Om, omr, 2, col, dim, Js);
        }
    }
    for (int p = 0; p < c->npt; p++)
        for (int o = P->pt_obs_begin[p]; o < P->pt_obs_begin[p + 1]; o++) {
            if (c->lvl[o]) continue;
            double e[2], JA[6], JB[12], JC[12];
            vis_eval(c, p, o, e, JA, JB, JC);
            /* g2o linearises at the state whose errors were just computed: _error == e here */
            double w = P->obs_w[o];
            const double s = chi2_2(e, w);
            double rw = 1.0;
            if (c->vis_robust) { vbo_huber(s, P->huber_vis, rho); rw = rho[1]; }
            const double Wt = rw * w; /* robustInformation, base_edge.h:96-102 */
            const double omr[2] = {-w * e[0] * rw, -w * e[1] * rw};
            double *Hl = c->Hll + (size_t)p * L * L, *bl = c->b + c->np + p * L;
            for (int a = 0; a < L; a++) {
                for (int bb = 0; bb < L; bb++) Hl[a * L + bb] += JA[a] * Wt * JA[bb] + JA[L + a] * Wt * JA[L + bb];
                bl[a] += JA[a] * omr[0] + JA[L + a] * omr[1];
            }
            if (c->variant == VBA_VARIANT_PRV_IDP) {
                const int cr = kf_col(c, P->pt_ref_kf[p]), co = kf_col(c, P->obs_kf[o]);
                const int col[2] = {cr, co};
                const int dim[2] = {6, 6};
                const double Om[4] = {Wt, 0, 0, Wt};
                const double *const Js[2] = {JB, JC};
                accum_pose(c, 2, Om, omr, 2, col, dim, Js);
                if (cr >= 0)
                    for (int a = 0; a < 6; a++) c->Wref[6 * p + a] += JB[a] * Wt * JA[0] + JB[6 + a] * Wt * JA[1];
                if (co >= 0)
                    for (int a = 0; a < 6; a++) c->Wobs[6 * o + a] = JC[a] * Wt * JA[0] + JC[6 + a] * Wt * JA[1];
            } else {
                const int co = kf_col(c, P->obs_kf[o]);
                const int col[1] = {co};
                const int dim[1] = {6};
                const double Om[4] = {Wt, 0, 0, Wt};
                const double *const Js[1] = {JB};
                accum_pose(c, 2, Om, omr, 1, col, dim, Js);
                if (co >= 0)
                    for (int a = 0; a < 6; a++)
                        for (int bb = 0; bb < 3; bb++)
                            c->Wobs[18 * o + 3 * a + bb] = JB[a] * Wt * JA[bb] + JB[6 + a] * Wt * JA[3 + bb];
            }
        }
}

/* Eigen Matrix3d::inverse() (cofactor formula, compute_inverse<3>) / scalar reciprocal */
static void small_inverse(const double *D, int L, double *Di) {
    if (L == 1) { Di[0] = 1.0 / D[0]; return; }
    const double c00 = D[4] * D[8] - D[5] * D[7], c10 = D[5] * D[6] - D[3] * D[8], c20 = D[3] * D[7] - D[4] * D[6];
    const double det = D[0] * c00 + D[1] * c10 + D[2] * c20;
    const double id = 1.0 / det;
    Di[0] = c00 * id; Di[1] = (D[2] * D[7] - D[1] * D[8]) * id; Di[2] = (D[1] * D[5] - D[2] * D[4]) * id;
    Di[3] = c10 * id; Di[4] = (D[0] * D[8] - D[2] * D[6]) * id; Di[5] = (D[2] * D[3] - D[0] * D[5]) * id;
    Di[6] = c20 * id; Di[7] = (D[1] * D[6] - D[0] * D[7]) * id; Di[8] = (D[0] * D[4] - D[1] * D[3]) * id;
}

/* LDL^T without pivoting on the (optionally permuted) dense symmetric S, restricted to the row profile.
 * Restates LinearSolverEigen::solve (linear_solver_eigen.h:94-124, SimplicialLDLT): same factorisation
 * up to the fill-reducing order; fails only on an exactly zero (or non-finite) pivot. */
static int ldlt_solve(ctx *c, const double *S, const double *rhs, double *x) {
    const int n = c->np;
    double *A = c->Lwork, *y = c->ywork;
    const int *pm = c->perm;
    for (int i = 0; i < n; i++) {
        const double *src = S + (size_t)pm[i] * n;
        double *dst = A + (size_t)i * n;
        int f = i;
        for (int j = 0; j <= i; j++) {
            dst[j] = src[pm[j]];
            if (dst[j] != 0.0 && j < f) f = j;
        }
        c->first[i] = f;
        y[i] = rhs[pm[i]];
    }
    /* row-by-row (bordering) LDL^T inside the profile; D stored on the diagonal, unit L strictly below */
    for (int i = 0; i < n; i++) {
        double *Ai = A + (size_t)i * n;
        const int fi = c->first[i];
        for (int j = fi; j < i; j++) { /* t_j = L_ij D_j = A_ij - sum_k t_k L_jk */
            const double *Aj = A + (size_t)j * n;
            const int fj = c->first[j];
            double s = Ai[j];
            for (int k = (fi > fj ? fi : fj); k < j; k++) s -= Ai[k] * Aj[k];
            Ai[j] = s;
        }
        double d = Ai[i];
        for (int j = fi; j < i; j++) {
            const double t = Ai[j];
            const double l = t / A[(size_t)j * n + j];
            d -= l * t;
            Ai[j] = l;
        }
        if (d == 0.0 || !isfinite(d)) return 0;
        Ai[i] = d;
    }
    for (int i = 0; i < n; i++) { /* L z = y */
        const double *Ai = A + (size_t)i * n;
        double s = y[i];
        for (int j = c->first[i]; j < i; j++) s -= Ai[j] * y[j];
        y[i] = s;
    }
    for (int i = 0; i < n; i++) y[i] /= A[(size_t)i * n + i];
    for (int i = n - 1; i >= 0; i--) { /* L^T x = z */
        const double *Ai = A + (size_t)i * n;
        const double yi = y[i];
        for (int j = c->first[i]; j < i; j++) y[j] -= Ai[j] * yi;
    }
    for (int i = 0; i < n; i++) x[pm[i]] = y[i];
    return 1;
}

/* BlockSolver::solve with Schur complement, block_solver.hpp:354-486.  lambda: LM damping already added
 * by setLambda (:564-589) to every diagonal entry of H_pp and H_ll of the active vertices. */
static int solve_system(ctx *c, double lambda) {
    const vba_problem *P = c->P;
    const int L = c->ldim, np = c->np;
    memcpy(c->S, c->Hpp, sizeof(double) * (size_t)np * np);
    memcpy(c->bs, c->b, sizeof(double) * np);
    for (int i = 0; i < np; i++) {
        if (c->var_act[i]) c->S[(size_t)i * np + i] += lambda;
        else { c->S[(size_t)i * np + i] = 1.0; c->bs[i] = 0.0; } /* vertex outside the index mapping: no column */
    }
    for (int p = 0; p < c->npt; p++) {
        if (!c->pt_act[p]) continue;
        double D[9], *Di = c->Dinv + (size_t)p * L * L, db[3];
        memcpy(D, c->Hll + (size_t)p * L * L, sizeof(double) * L * L);
        for (int a = 0; a < L; a++) D[a * L + a] += lambda;
        small_inverse(D, L, Di);
        const double *bl = c->b + np + p * L;
        for (int a = 0; a < L; a++) {
            double s = 0;
            for (int k = 0; k < L; k++) s += Di[a * L + k] * bl[k];
            db[a] = s;
        }
        /* incidence list of this landmark column: (pose column, block) */
        int cols[1024];
        const double *Ws[1024];
        int ni = 0;
        if (c->variant == VBA_VARIANT_PRV_IDP) {
            const int cr = kf_col(c, P->pt_ref_kf[p]);
            if (cr >= 0) { cols[ni] = cr; Ws[ni] = c->Wref + 6 * p; ni++; }
        }
        for (int o = P->pt_obs_begin[p]; o < P->pt_obs_begin[p + 1] && ni < 1024; o++) {
            if (c->lvl[o]) continue;
            const int co = kf_col(c, P->obs_kf[o]);
            if (co >= 0) { cols[ni] = co; Ws[ni] = c->Wobs + (size_t)6 * L * o; ni++; }
        }
        for (int i1 = 0; i1 < ni; i1++) {
            double BD[18];
            for (int a = 0; a < 6; a++)
                for (int k = 0; k < L; k++) {
                    double s = 0;
                    for (int m = 0; m < L; m++) s += Ws[i1][a * L + m] * Di[m * L + k];
                    BD[a * L + k] = s;
                }
            for (int a = 0; a < 6; a++) {
                double s = 0;
                for (int k = 0; k < L; k++) s += Ws[i1][a * L + k] * db[k];
                c->bs[cols[i1] + a] -= s;
            }
            for (int i2 = 0; i2 < ni; i2++) {
                for (int a = 0; a < 6; a++)
                    for (int bb = 0; bb < 6; bb++) {
                        double s = 0;
                        for (int k = 0; k < L; k++) s += BD[a * L + k] * Ws[i2][bb * L + k];
                        c->S[(size_t)(cols[i1] + a) * np + cols[i2] + bb] -= s;
                    }
            }
        }
    }
    memset(c->x, 0, sizeof(double) * (np + c->nl));
    if (!ldlt_solve(c, c->S, c->bs, c->x)) return 0;
    /* landmark back-substitution :461-481 */
    for (int p = 0; p < c->npt; p++) {
        if (!c->pt_act[p]) continue;
        double cl[3];
        const double *bl = c->b + np + p * L, *Di = c->Dinv + (size_t)p * L * L;
        for (int a = 0; a < L; a++) cl[a] = bl[a];
        if (c->variant == VBA_VARIANT_PRV_IDP) {
            const int cr = kf_col(c, P->pt_ref_kf[p]);
            if (cr >= 0)
                for (int a = 0; a < 6; a++) cl[0] -= c->Wref[6 * p + a] * c->x[cr + a];
        }
        for (int o = P->pt_obs_begin[p]; o < P->pt_obs_begin[p + 1]; o++) {
            if (c->lvl[o]) continue;
            const int co = kf_col(c, P->obs_kf[o]);
            if (co < 0) continue;
            const double *W = c->Wobs + (size_t)6 * L * o;
            for (int k = 0; k < L; k++)
                for (int a = 0; a < 6; a++) cl[k] -= W[a * L + k] * c->x[co + a];
        }
        for (int a = 0; a < L; a++) {
            double s = 0;
            for (int k = 0; k < L; k++) s += Di[a * L + k] * cl[k];
            c->x[np + p * L + a] = s;
        }
    }
    return 1;
}

/* SparseOptimizer::update (sparse_optimizer.cpp:422-435) -> vertex oplus (A7) */
static void apply_update(ctx *c) {
    for (int a = 0; a < c->nfree; a++) {
        const int col = a * c->pdim;
        const double *dx = c->x + col;
        double *T = c->pose + 7 * a;
        if (c->variant == VBA_VARIANT_SE3_XYZ) {
            if (!c->var_act[col]) continue;
            double E[7];
            vbo_se3_exp(dx, E);      /* VertexSE3Expmap::oplusImpl, types_six_dof_expmap.h:73-76 */
            se3_mul(E, T, T);
        } else {
            if (c->var_act[col]) {   /* NavState::IncSmallPR, NavState.cpp:63-70 */
                T[0] += dx[0]; T[1] += dx[1]; T[2] += dx[2];
                double dq[4];
                vbo_so3_exp(dx + 3, dq);
                so3_mul(T + 3, dq, T + 3);
            }
            if (c->var_act[col + 6])  /* IncSmallV :74-77 */
                for (int k = 0; k < 3; k++) c->vel[3 * a + k] += dx[6 + k];
            if (c->var_act[col + 9])  /* IncSmallBias :100-109 */
                for (int k = 0; k < 6; k++) c->bias[12 * a + 6 + k] += dx[9 + k];
        }
    }
    for (int p = 0; p < c->npt; p++) {
        if (!c->pt_act[p]) continue;
        const double *dl = c->x + c->np + p * c->ldim;
        if (c->variant == VBA_VARIANT_PRV_IDP) { /* VertexIDP::oplusImpl, g2otypes.h:50-55 */
            c->pt[3 * p] += dl[0];
            if (c->pt[3 * p] < 1e-6) c->pt[3 * p] = 1e-6;
        } else {                                   /* VertexSBAPointXYZ::oplusImpl, types_sba.h:52-56 */
            for (int k = 0; k < 3; k++) c->pt[3 * p + k] += dl[k];
        }
    }
}

static void push_state(ctx *c) {
    memcpy(c->pose_bk, c->pose, sizeof(double) * 7 * c->nkf);
    memcpy(c->vel_bk, c->vel, sizeof(double) * 3 * c->nkf);
    memcpy(c->bias_bk, c->bias, sizeof(double) * 12 * c->nkf);
    memcpy(c->pt_bk, c->pt, sizeof(double) * 3 * c->npt);
}
static void pop_state(ctx *c) {
    memcpy(c->pose, c->pose_bk, sizeof(double) * 7 * c->nkf);
    memcpy(c->vel, c->vel_bk, sizeof(double) * 3 * c->nkf);
    memcpy(c->bias, c->bias_bk, sizeof(double) * 12 * c->nkf);
    memcpy(c->pt, c->pt_bk, sizeof(double) * 3 * c->npt);
}

static void trace(vba_result *out, double v) {
    if (out->n_trace < VBA_TRACE_MAX) out->chi2_trace[out->n_trace++] = v;
}

/* One read of g2o's forceStopFlag: SparseOptimizer::terminate() (sparse_optimizer.cpp:376, levenberg.cpp:149) and the
 * bDoMore check between the stages (src/Optimizer.cpp:462-466).  The polls are counted from the first terminate() of the first
 * optimize(); with stop_after >= 0 the flag reads 1 from poll number stop_after on -- the deterministic stand-in for
 * LocalMapping::InterruptBA firing in the middle of a solve (the GPU backend has the same counter: vba_debug_set_stop_after). */
static int stop_now(ctx *c, const volatile int *stop) {
    const int n = c->polls++;
    return (stop && *stop) || (c->stop_after >= 0 && n >= c->stop_after);
}

/* SparseOptimizer::optimize (sparse_optimizer.cpp:354-419) driving OptimizationAlgorithmGaussNewton::solve
 * (optimization_algorithm_gauss_newton.cpp:50-105).  Returns cjIterations; *failed set on solver Fail. */
static int optimize_gn(ctx *c, int iterations, const volatile int *stop, vba_result *out, int *failed) {
    int cj = 0;
    for (int i = 0; i < iterations && !stop_now(c, stop); i++) {
        const double pre = compute_errors(c);
        if (i == 0) trace(out, pre);
        build_system(c);
        const int ok = solve_system(c, 0.0);
        /* on Fail (zero / non-finite pivot, linear_solver_eigen.h:105-111) g2o applies whatever x held before (stale); here the
         * step is dropped (DESIGN.md deviation) -- the chi2 then cannot move, so the Terminate test below would hide the failure:
         * it is recorded first (status VBA_SOLVER_FAILED), and this optimize() ends either way (Terminate or Fail: both leave the
         * loop of sparse_optimizer.cpp:376) */
        if (ok) apply_update(c);
        else *failed = 1;
        const double post = compute_errors(c);
        trace(out, post);
        ++cj;
        if (fabs(pre - post) < 1e-3) break; /* Terminate */
        if (!ok) break;
    }
    return cj;
}

/* OptimizationAlgorithmLevenberg::solve, optimization_algorithm_levenberg.cpp:61-164 */
static int optimize_lm(ctx *c, int iterations, const volatile int *stop, vba_result *out, int *failed) {
    int cj = 0, nBad = 0;
    double lambda = 0, ni = 2;
    (void)failed;
    for (int it = 0; it < iterations && !stop_now(c, stop); it++) {
        double cur = compute_errors(c);
        double tempChi = cur;
        const double iniChi = cur;
        if (it == 0) trace(out, cur);
        build_system(c);
        if (it == 0) { /* computeLambdaInit :166-180, tau = 1e-5 */
            double mx = 0;
            for (int i = 0; i < c->np; i++)
                if (c->var_act[i]) mx = fmax(fabs(c->Hpp[(size_t)i * c->np + i]), mx);
            for (int p = 0; p < c->npt; p++)
                if (c->pt_act[p])
                    for (int a = 0; a < c->ldim; a++) mx = fmax(fabs(c->Hll[(size_t)p * c->ldim * c->ldim + a * c->ldim + a]), mx);
            lambda = 1e-5 * mx;
            ni = 2;
            nBad = 0;
        }
        double rho = 0;
        int qmax = 0;
        do {
            push_state(c);
            const int ok2 = solve_system(c, lambda);
            if (ok2) apply_update(c);
            tempChi = compute_errors(c);
            if (!ok2) tempChi = DBL_MAX;
            rho = cur - tempChi;
            double scale = 0; /* computeScale :182-189 over poses and landmarks */
            for (int j = 0; j < c->np + c->nl; j++) scale += c->x[j] * (lambda * c->x[j] + c->b[j]);
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && isfinite(tempChi)) {
                double alpha = 1. - pow((2 * rho - 1), 3);
                alpha = fmin(alpha, 2. / 3.);
                const double sf = fmax(1. / 3., alpha);
                lambda *= sf;
                ni = 2;
                cur = tempChi;
            } else {
                lambda *= ni;
                ni *= 2;
                pop_state(c);
            }
            qmax++;
        } while (rho < 0 && qmax < 10 && !stop_now(c, stop));
        trace(out, cur);
        ++cj;
        if (qmax == 10 || rho == 0) break;
        if ((iniChi - cur) * 1e3 < iniChi) nBad++;
        else nBad = 0;
        if (nBad >= 3) break;
    }
    out->lambda_final = lambda;
    return cj;
}

static void *xcalloc(size_t n, size_t sz) {
    void *p = calloc(n ? n : 1, sz);
    if (!p) { fprintf(stderr, "vba_oracle: out of memory\n"); abort(); }
    return p;
}

/* solver_mode: 0 = LDL^T in natural (g2o vertex-id) order, 1 = V/Bias-first order (fewer flops; timed baseline) */
int vba_oracle_solve_ex(vba_problem *P, vba_result *out, const volatile int *stop, int solver_mode, int stop_after);
int vba_oracle_solve(vba_problem *P, vba_result *out, const volatile int *stop, int solver_mode) {
    return vba_oracle_solve_ex(P, out, stop, solver_mode, -1);
}
/* stop_after: test hook, see stop_now (-1: only the caller's flag) */
int vba_oracle_solve_ex(vba_problem *P, vba_result *out, const volatile int *stop, int solver_mode, int stop_after) {
    out->chi2_vis = out->chi2_prv = out->chi2_bias = 0;
    out->its_done[0] = out->its_done[1] = 0;
    out->n_outliers = 0;
    out->n_trace = 0;
    out->lambda_final = 0;
    out->status = VBA_OK;
    if (stop && *stop) { out->status = VBA_ABORTED_BEFORE; return 0; } /* src/Optimizer.cpp:453-455 */
    if (P->n_kf_free <= 0 || P->n_kf_free > P->n_kf) return -1;
    ctx C, *c = &C;
    memset(c, 0, sizeof C);
    c->P = P;
    c->variant = P->variant;
    c->polls = 0; c->stop_after = stop_after;
    c->nkf = P->n_kf; c->nfree = P->n_kf_free; c->npt = P->n_pt; c->nobs = P->n_obs;
    c->fix = P->kf_fix; c->imu_robust = !(P->protocol == VBA_PROTO_SINGLE && !P->robust);
    c->nimu = (P->variant == VBA_VARIANT_SE3_XYZ) ? 0 : P->n_imu;
    c->pdim = (P->variant == VBA_VARIANT_SE3_XYZ) ? 6 : 15;
    c->np = c->pdim * c->nfree;
    c->ldim = (P->variant == VBA_VARIANT_PRV_IDP) ? 1 : 3;
    c->nl = c->ldim * c->npt;
    c->pose = xcalloc(7 * c->nkf, 8); c->vel = xcalloc(3 * c->nkf, 8); c->bias = xcalloc(12 * c->nkf, 8);
    c->pt = xcalloc(3 * c->npt, 8);
    c->pose_bk = xcalloc(7 * c->nkf, 8); c->vel_bk = xcalloc(3 * c->nkf, 8); c->bias_bk = xcalloc(12 * c->nkf, 8);
    c->pt_bk = xcalloc(3 * c->npt, 8);
    memcpy(c->pose, P->kf_pose, 56 * c->nkf);
    if (P->kf_vel) memcpy(c->vel, P->kf_vel, 24 * c->nkf);
    if (P->kf_bias) memcpy(c->bias, P->kf_bias, 96 * c->nkf);
    memcpy(c->pt, P->pt, 24 * c->npt);
    c->lvl = xcalloc(c->nobs, 1);
    c->err = xcalloc(2 * c->nobs, 8);
    c->imu_err = xcalloc(9 * c->nimu, 8);
    c->bias_err = xcalloc(6 * c->nimu, 8);
    c->Hpp = xcalloc((size_t)c->np * c->np, 8);
    c->b = xcalloc(c->np + c->nl, 8);
    c->Hll = xcalloc((size_t)c->npt * c->ldim * c->ldim, 8);
    c->Wobs = xcalloc((size_t)c->nobs * 6 * c->ldim, 8);
    c->Wref = (P->variant == VBA_VARIANT_PRV_IDP) ? xcalloc((size_t)c->npt * 6, 8) : NULL;
    c->Dinv = xcalloc((size_t)c->npt * c->ldim * c->ldim, 8);
    c->S = xcalloc((size_t)c->np * c->np, 8);
    c->bs = xcalloc(c->np, 8);
    c->x = xcalloc(c->np + c->nl, 8);
    c->pt_act = xcalloc(c->npt, 1);
    c->var_act = xcalloc(c->np, 1);
    c->perm = xcalloc(c->np, sizeof(int));
    c->first = xcalloc(c->np, sizeof(int));
    c->Lwork = xcalloc((size_t)c->np * c->np, 8);
    c->ywork = xcalloc(c->np, 8);
    if (solver_mode == 1 && c->pdim == 15) {
        int k = 0;
        for (int a = 0; a < c->nfree; a++)
            for (int j = 6; j < 15; j++) c->perm[k++] = 15 * a + j;
        for (int a = 0; a < c->nfree; a++)
            for (int j = 0; j < 6; j++) c->perm[k++] = 15 * a + j;
    } else
        for (int i = 0; i < c->np; i++) c->perm[i] = i;

    int failed = 0;
    /* stage 1: all edges level 0, Huber everywhere; optimize(its_stage1)  (src/Optimizer.cpp:458-459) */
    const int single = P->protocol == VBA_PROTO_SINGLE; /* BundleAdjustment: one optimize(nIterations), :3517 / :835 */
    c->vis_robust = single ? (P->robust != 0) : 1;
    init_active(c);
    out->its_done[0] = (P->algo == VBA_ALGO_LM) ? optimize_lm(c, P->its_stage1, stop, out, &failed)
                                                 : optimize_gn(c, P->its_stage1, stop, out, &failed);
    int do_more = !single && !stop_now(c, stop); /* :462-466 */
    if (!do_more && !single) out->status = VBA_ABORTED_AFTER_STAGE1;
    if (do_more) {
        /* outlier pass :475-490 (reads e->chi2() from the stored _error, recomputes the depth) */
        for (int p = 0; p < c->npt; p++)
            for (int o = P->pt_obs_begin[p]; o < P->pt_obs_begin[p + 1]; o++) {
                double e[2];
                const double z = vis_eval(c, p, o, e, NULL, NULL, NULL);
                const double chi = chi2_2(c->err + 2 * o, P->obs_w[o]);
                int bad = (chi > P->chi2_th) || !(z > P->depth_min);
                if (c->variant == VBA_VARIANT_PRV_IDP && c->pt[3 * p] < P->rho_min) bad = 1;
                if (bad) c->lvl[o] = 1;
            }
        c->vis_robust = 0; /* setRobustKernel(0) on every vision edge */
        init_active(c);    /* initializeOptimization(0) :492 */
        out->its_done[1] = (P->algo == VBA_ALGO_LM) ? optimize_lm(c, P->its_stage2, stop, out, &failed)
                                                     : optimize_gn(c, P->its_stage2, stop, out, &failed);
    }
    if (failed) out->status = VBA_SOLVER_FAILED;
    /* erase list + chi2 :496-517 */
    for (int p = 0; p < c->npt; p++)
        for (int o = P->pt_obs_begin[p]; o < P->pt_obs_begin[p + 1]; o++) {
            double e[2];
            const double z = vis_eval(c, p, o, e, NULL, NULL, NULL);
            const double chi = chi2_2(c->err + 2 * o, P->obs_w[o]);
            int bad = (chi > P->chi2_th) || !(z > P->depth_min);
            if (c->variant == VBA_VARIANT_PRV_IDP && (c->pt[3 * p] < P->rho_min || c->lvl[o] != 0)) bad = 1;
            if (single) bad = 0; /* global BA classifies nothing */
            if (out->obs_outlier) out->obs_outlier[o] = (uint8_t)bad;
            if (out->obs_chi2) out->obs_chi2[o] = chi;
            out->n_outliers += bad;
            if (!c->lvl[o]) out->chi2_vis += chi2_2(e, P->obs_w[o]); /* recomputed at the final estimates (N3) */
        }
    for (int k = 0; k < c->nimu; k++) {
        const int act = imu_edge_active(c, k);
        if (!act) continue;
        const int i = P->imu_kf_i[k], j = P->imu_kf_j[k];
        const double *meas = P->imu_meas + VBA_IMU_MEAS_STRIDE * k;
        double e9[9], e6[6];
        vbo_edge_prv_error(c->pose + 7 * i, c->pose + 7 * j, c->vel + 3 * i, c->vel + 3 * j, c->bias + 12 * i, meas, P->g_w, e9);
        if (act & 1) out->chi2_prv += quadform(e9, P->imu_info_prv + 81 * k, 9);
        vbo_edge_bias_error(c->bias + 12 * i, c->bias + 12 * j, e6);
        const double wg = P->inv_bg_rw2 / meas[0], wa = P->inv_ba_rw2 / meas[0];
        if (act & 2) out->chi2_bias += wg * (e6[0] * e6[0] + e6[1] * e6[1] + e6[2] * e6[2]) + wa * (e6[3] * e6[3] + e6[4] * e6[4] + e6[5] * e6[5]);
    }
    /* write back free entries */
    memcpy(P->kf_pose, c->pose, 56 * c->nfree);
    if (P->kf_vel && c->pdim == 15) memcpy(P->kf_vel, c->vel, 24 * c->nfree);
    if (P->kf_bias && c->pdim == 15) memcpy(P->kf_bias, c->bias, 96 * c->nfree);
    memcpy(P->pt, c->pt, 24 * c->npt);
    free(c->pose); free(c->vel); free(c->bias); free(c->pt);
    free(c->pose_bk); free(c->vel_bk); free(c->bias_bk); free(c->pt_bk);
    free(c->lvl); free(c->err); free(c->imu_err); free(c->bias_err);
    free(c->Hpp); free(c->b); free(c->Hll); free(c->Wobs); free(c->Wref); free(c->Dinv);
    free(c->S); free(c->bs); free(c->x); free(c->pt_act); free(c->var_act);
    free(c->perm); free(c->first); free(c->Lwork); free(c->ywork);
    return 0;
}

/* ---- test hooks: one linearisation of the whole problem, dense, for cross-checks against numpy ---- */
/* Fills H (full (np+nl)^2 dense, row-major), b, and the Schur solution x for the problem's CURRENT state,
 * all edges level 0, Huber as in stage 1.  Returns np+nl (or <0). */
int vba_oracle_linearize_ex(vba_problem *P, double lambda, int robust_vis, const uint8_t *lvl, double *Hfull, double *bfull,
                            double *xschur, double *chi2);
int vba_oracle_linearize(vba_problem *P, double lambda, double *Hfull, double *bfull, double *xschur, double *chi2) {
    return vba_oracle_linearize_ex(P, lambda, 1, NULL, Hfull, bfull, xschur, chi2);
}
/* the same with the stage's settings spelled out: Huber on the vision edges or not, and the g2o level of every vision edge
 * (NULL: all at level 0) -- lets tests/test_oracle_protocol.py drive the two-stage protocol from Python */
int vba_oracle_linearize_ex(vba_problem *P, double lambda, int robust_vis, const uint8_t *lvl, double *Hfull, double *bfull,
                            double *xschur, double *chi2) {
    vba_result dummy;
    memset(&dummy, 0, sizeof dummy);
    ctx C, *c = &C;
    memset(c, 0, sizeof C);
    c->P = P; c->variant = P->variant;
    c->nkf = P->n_kf; c->nfree = P->n_kf_free; c->npt = P->n_pt; c->nobs = P->n_obs;
    c->fix = P->kf_fix; c->imu_robust = !(P->protocol == VBA_PROTO_SINGLE && !P->robust);
    c->nimu = (P->variant == VBA_VARIANT_SE3_XYZ) ? 0 : P->n_imu;
    c->pdim = (P->variant == VBA_VARIANT_SE3_XYZ) ? 6 : 15;
    c->np = c->pdim * c->nfree;
    c->ldim = (P->variant == VBA_VARIANT_PRV_IDP) ? 1 : 3;
    c->nl = c->ldim * c->npt;
    const int n = c->np + c->nl;
    c->pose = P->kf_pose; c->vel = P->kf_vel; c->bias = P->kf_bias; c->pt = P->pt;
    double *zv = NULL, *zb = NULL;
    if (!c->vel) c->vel = zv = xcalloc(3 * c->nkf, 8);
    if (!c->bias) c->bias = zb = xcalloc(12 * c->nkf, 8);
    c->lvl = xcalloc(c->nobs, 1);
    c->err = xcalloc(2 * c->nobs, 8);
    c->imu_err = xcalloc(9 * c->nimu, 8);
    c->bias_err = xcalloc(6 * c->nimu, 8);
    c->Hpp = xcalloc((size_t)c->np * c->np, 8);
    c->b = xcalloc(n, 8);
    c->Hll = xcalloc((size_t)c->npt * c->ldim * c->ldim, 8);
    c->Wobs = xcalloc((size_t)c->nobs * 6 * c->ldim, 8);
    c->Wref = (P->variant == VBA_VARIANT_PRV_IDP) ? xcalloc((size_t)c->npt * 6, 8) : NULL;
    c->Dinv = xcalloc((size_t)c->npt * c->ldim * c->ldim, 8);
    c->S = xcalloc((size_t)c->np * c->np, 8);
    c->bs = xcalloc(c->np, 8);
    c->x = xcalloc(n, 8);
    c->pt_act = xcalloc(c->npt, 1);
    c->var_act = xcalloc(c->np, 1);
    c->perm = xcalloc(c->np, sizeof(int));
    c->first = xcalloc(c->np, sizeof(int));
    c->Lwork = xcalloc((size_t)c->np * c->np, 8);
    c->ywork = xcalloc(c->np, 8);
    for (int i = 0; i < c->np; i++) c->perm[i] = i;
    c->vis_robust = robust_vis;
    if (lvl) memcpy(c->lvl, lvl, c->nobs);
    init_active(c);
    *chi2 = compute_errors(c);
    build_system(c);
    const int ok = solve_system(c, lambda);
    if (Hfull) {
        const int L = c->ldim;
        memset(Hfull, 0, sizeof(double) * (size_t)n * n);
        for (int i = 0; i < c->np; i++)
            for (int j = 0; j < c->np; j++) Hfull[(size_t)i * n + j] = c->Hpp[(size_t)i * c->np + j];
        for (int p = 0; p < c->npt; p++) {
            for (int a = 0; a < L; a++)
                for (int bb = 0; bb < L; bb++) Hfull[(size_t)(c->np + p * L + a) * n + c->np + p * L + bb] = c->Hll[(size_t)p * L * L + a * L + bb];
            if (c->variant == VBA_VARIANT_PRV_IDP) {
                const int cr = kf_col(c, P->pt_ref_kf[p]);
                if (cr >= 0)
                    for (int a = 0; a < 6; a++) {
                        Hfull[(size_t)(cr + a) * n + c->np + p] += c->Wref[6 * p + a];
                        Hfull[(size_t)(c->np + p) * n + cr + a] += c->Wref[6 * p + a];
                    }
            }
            for (int o = P->pt_obs_begin[p]; o < P->pt_obs_begin[p + 1]; o++) {
                const int co = kf_col(c, P->obs_kf[o]);
                if (co < 0) continue;
                for (int a = 0; a < 6; a++)
                    for (int k = 0; k < L; k++) {
                        Hfull[(size_t)(co + a) * n + c->np + p * L + k] += c->Wobs[(size_t)6 * L * o + a * L + k];
                        Hfull[(size_t)(c->np + p * L + k) * n + co + a] += c->Wobs[(size_t)6 * L * o + a * L + k];
                    }
            }
        }
    }
    if (bfull) memcpy(bfull, c->b, sizeof(double) * n);
    if (xschur) memcpy(xschur, c->x, sizeof(double) * n);
    free(zv); free(zb);
    free(c->lvl); free(c->err); free(c->imu_err); free(c->bias_err);
    free(c->Hpp); free(c->b); free(c->Hll); free(c->Wobs); free(c->Wref); free(c->Dinv);
    free(c->S); free(c->bs); free(c->x); free(c->pt_act); free(c->var_act);
    free(c->perm); free(c->first); free(c->Lwork); free(c->ywork);
    return ok ? n : -1;
}

/* Residual-only evaluation at the problem's CURRENT state (no system is built: cheap at any size).
 * out[0] = robust chi2 of all edges (Huber on vision iff robust_vis), out[1..3] = plain chi2 of vision / PRV / bias
 * edges; per-edge vision chi2 and depth in obs_chi2 / obs_depth (may be NULL). */
int vba_oracle_eval(vba_problem *P, int robust_vis, double *out, double *obs_chi2, double *obs_depth) {
    ctx C, *c = &C;
    memset(c, 0, sizeof C);
    c->P = P; c->variant = P->variant;
    c->nkf = P->n_kf; c->nfree = P->n_kf_free; c->npt = P->n_pt; c->nobs = P->n_obs;
    c->fix = P->kf_fix; c->imu_robust = !(P->protocol == VBA_PROTO_SINGLE && !P->robust);
    c->nimu = (P->variant == VBA_VARIANT_SE3_XYZ) ? 0 : P->n_imu;
    c->pose = P->kf_pose; c->vel = P->kf_vel; c->bias = P->kf_bias; c->pt = P->pt;
    double rho[3];
    out[0] = out[1] = out[2] = out[3] = 0;
    for (int k = 0; k < c->nimu; k++) {
        const int act = imu_edge_active(c, k);
        if (!act) continue;
        const int i = P->imu_kf_i[k], j = P->imu_kf_j[k];
        const double *meas = P->imu_meas + VBA_IMU_MEAS_STRIDE * k;
        double e9[9], e6[6];
        vbo_edge_prv_error(c->pose + 7 * i, c->pose + 7 * j, c->vel + 3 * i, c->vel + 3 * j, c->bias + 12 * i, meas, P->g_w, e9);
        const double s = quadform(e9, P->imu_info_prv + 81 * k, 9);
        huber_or_not(c->imu_robust, s, P->huber_prv, rho);
        if (act & 1) { out[0] += rho[0]; out[2] += s; }
        vbo_edge_bias_error(c->bias + 12 * i, c->bias + 12 * j, e6);
        const double wg = P->inv_bg_rw2 / meas[0], wa = P->inv_ba_rw2 / meas[0];
        const double sb = wg * (e6[0] * e6[0] + e6[1] * e6[1] + e6[2] * e6[2]) + wa * (e6[3] * e6[3] + e6[4] * e6[4] + e6[5] * e6[5]);
        huber_or_not(c->imu_robust, sb, P->huber_bias, rho);
        if (act & 2) { out[0] += rho[0]; out[3] += sb; }
    }
    for (int p = 0; p < c->npt; p++)
        for (int o = P->pt_obs_begin[p]; o < P->pt_obs_begin[p + 1]; o++) {
            double e[2];
            const double z = vis_eval(c, p, o, e, NULL, NULL, NULL);
            const double s = chi2_2(e, P->obs_w[o]);
            if (robust_vis) { vbo_huber(s, P->huber_vis, rho); out[0] += rho[0]; }
            else out[0] += s;
            out[1] += s;
            if (obs_chi2) obs_chi2[o] = s;
            if (obs_depth) obs_depth[o] = z;
        }
    return 0;
}

/* retraction hooks for the finite-difference Jacobian tests (same oplus the optimiser uses) */
void vbo_oplus_pr(double *pose7, const double *d6) { /* NavState::IncSmallPR */
    pose7[0] += d6[0]; pose7[1] += d6[1]; pose7[2] += d6[2];
    double dq[4];
    vbo_so3_exp(d6 + 3, dq);
    so3_mul(pose7 + 3, dq, pose7 + 3);
}
void vbo_oplus_se3(double *T7, const double *d6) { /* VertexSE3Expmap::oplusImpl */
    double E[7];
    vbo_se3_exp(d6, E);
    se3_mul(E, T7, T7);
}
void vbo_quat_to_R(const double *q, double *R) { quat_to_R(q, R); }
void vbo_R_to_quat(const double *R, double *q) { R_to_quat(R, q); }

/* =================================================================================================
 * IMU-aided per-frame pose optimisation: CPU restatement of
 * Optimizer::PoseOptimization(Frame*, KeyFrame*|Frame*, IMUPreintegrator, gw, bComputeMarg), src/Optimizer.cpp:1671-2317.
 * Hessian order: [frame PVR (9) | frame Bias (6) | last-frame PVR (9) | last-frame Bias (6)].
 * ================================================================================================= */
typedef struct {
    const vba_frame_problem *F;
    int n;                   /* 15 or 30 */
    double cur[22], last[22], cur_bk[22], last_bk[22];
    unsigned char *lvl, *lvl_last; /* g2o edge level: 1 = outside the active set */
    double *err, *err_last;  /* stored _error of the mono edges */
    int vis_robust;
    double e_pvr[9], e_bias[6], e_prior[15];
    double info_pvr[81];
    double H[900], b[30], x[30], Hlast[900];
} fctx;

/* VertexNavStatePVR::oplusImpl -> NavState::IncSmallPVR (NavState.cpp:81-96); VertexNavStateBias -> IncSmallBias (:100-109) */
static void nav_oplus(double *nav, const double *dpvr, const double *dbias) {
    for (int k = 0; k < 3; k++) { nav[k] += dpvr[k]; nav[7 + k] += dpvr[3 + k]; }
    double dq[4];
    vbo_so3_exp(dpvr + 6, dq);
    so3_mul(nav + 3, dq, nav + 3);
    for (int k = 0; k < 6; k++) nav[16 + k] += dbias[k];
}

/* EdgeNavStatePVRPointXYZOnlyPose: error (g2otypes.h, computeError) + Jacobian (g2otypes.cpp:792-837), J is 2x9 [P V R] */
static double frame_mono(const vba_frame_problem *F, const double *nav, const double *Pw, const double *uv, double *e, double *J) {
    double Rwb[9], Rcb[9], d[3], t1[3], Pc[3];
    quat_to_R(nav + 3, Rwb);
    quat_to_R(F->T_cb + 3, Rcb);
    for (int k = 0; k < 3; k++) d[k] = Pw[k] - nav[k];
    double RwbT[9];
    m3_T(Rwb, RwbT);
    m3_vec(RwbT, d, t1);
    m3_vec(Rcb, t1, Pc); /* Paux */
    const double Paux[3] = {Pc[0], Pc[1], Pc[2]};
    for (int k = 0; k < 3; k++) Pc[k] += F->T_cb[k];
    const double fx = F->K[0], fy = F->K[1], cx = F->K[2], cy = F->K[3];
    const double iz = 1.0 / Pc[2];
    e[0] = uv[0] - (fx * Pc[0] * iz + cx);
    e[1] = uv[1] - (fy * Pc[1] * iz + cy);
    if (J) {
        const double Jpi[6] = {fx * iz, 0, -Pc[0] * iz * fx * iz, 0, fy * iz, -Pc[1] * iz * fy * iz};
        double M[9], HA[9], HR[9];
        m3_mul(Rcb, RwbT, M);       /* JdPwb = -Jpi * (-Rcb Rwb^T) = Jpi M */
        hat(Paux, HA);
        m3_mul(HA, Rcb, HR);        /* JdRwb = -Jpi * hat(Paux) Rcb */
        memset(J, 0, 18 * sizeof(double));
        for (int r = 0; r < 2; r++)
            for (int c = 0; c < 3; c++) {
                double s1 = 0, s2 = 0;
                for (int k = 0; k < 3; k++) { s1 += Jpi[3 * r + k] * M[3 * k + c]; s2 += Jpi[3 * r + k] * HR[3 * k + c]; }
                J[9 * r + c] = s1;
                J[9 * r + 6 + c] = -s2;
            }
    }
    return Pc[2];
}

/* EdgeNavStatePVR::computeError (g2otypes.cpp:529-585): the PRV residual with rows ordered rP, rV, rPhi */
static void frame_pvr_error(const vba_frame_problem *F, const double *ni, const double *nj, double *e) {
    double t[9];
    vbo_edge_prv_error(ni, nj, ni + 7, nj + 7, ni + 10, F->imu_meas, F->g_w, t); /* rP, rPhi, rV */
    for (int k = 0; k < 3; k++) { e[k] = t[k]; e[3 + k] = t[6 + k]; e[6 + k] = t[3 + k]; }
}
/* EdgeNavStatePVR::linearizeOplus (:587-701): Ji, Jj 9x9 (columns P V R), Jb 9x6 */
static void frame_pvr_jac(const vba_frame_problem *F, const double *ni, const double *nj, const double *e, double *Ji, double *Jj, double *Jb) {
    double t[9], JPRi[54], JPRj[54], JVi[27], JVj[27], JBi[54];
    for (int k = 0; k < 3; k++) { t[k] = e[k]; t[6 + k] = e[3 + k]; t[3 + k] = e[6 + k]; }
    vbo_edge_prv_jac(ni, nj, ni + 7, nj + 7, ni + 10, F->imu_meas, F->g_w, t, JPRi, JPRj, JVi, JVj, JBi);
    static const int rowmap[9] = {0, 1, 2, 6, 7, 8, 3, 4, 5}; /* PVR row r <- PRV row */
    for (int r = 0; r < 9; r++) {
        const int s = rowmap[r];
        for (int c = 0; c < 3; c++) {
            Ji[9 * r + c] = JPRi[6 * s + c]; Ji[9 * r + 3 + c] = JVi[3 * s + c]; Ji[9 * r + 6 + c] = JPRi[6 * s + 3 + c];
            Jj[9 * r + c] = JPRj[6 * s + c]; Jj[9 * r + 3 + c] = JVj[3 * s + c]; Jj[9 * r + 6 + c] = JPRj[6 * s + 3 + c];
        }
        for (int c = 0; c < 6; c++) Jb[6 * r + c] = JBi[6 * s + c];
    }
}
/* EdgeNavStatePriorPVRBias::computeError (:839-871) */
static void frame_prior_error(const vba_frame_problem *F, const double *nl, double *e) {
    const double *pr = F->prior_nav;
    for (int k = 0; k < 3; k++) { e[k] = pr[k] - nl[k]; e[3 + k] = pr[7 + k] - nl[7 + k]; }
    double qi[4], q[4];
    so3_inv(pr + 3, qi);
    so3_mul(qi, nl + 3, q);
    vbo_so3_log(q, e + 6);
    for (int k = 0; k < 3; k++) {
        e[9 + k] = (pr[10 + k] + pr[16 + k]) - (nl[10 + k] + nl[16 + k]);
        e[12 + k] = (pr[13 + k] + pr[19 + k]) - (nl[13 + k] + nl[19 + k]);
    }
}

/* computeActiveErrors + activeRobustChi2 */
static double frame_errors(fctx *c) {
    const vba_frame_problem *F = c->F;
    double chi = 0, rho[3];
    if (F->last_is_frame == VBA_FRAME_VISION) { /* EdgeSE3ProjectXYZOnlyPose only (src/Optimizer.cpp:3640-3672) */
        const double dm = (double)(float)sqrt(5.991);
        for (int i = 0; i < F->n_obs; i++) {
            if (c->lvl[i]) continue;
            vbo_edge_se3xyz(F->obs_pw + 3 * i, c->cur, F->K, F->obs_uv + 2 * i, c->err + 2 * i, NULL, NULL, NULL);
            const double s = chi2_2(c->err + 2 * i, F->obs_w[i]);
            if (c->vis_robust) { vbo_huber(s, dm, rho); chi += rho[0]; } else chi += s;
        }
        return chi;
    }
    if (F->last_is_frame) {
        frame_prior_error(F, c->last, c->e_prior);
        vbo_huber(quadform(c->e_prior, F->prior_info, 15), (double)(float)sqrt(30.5779), rho);
        chi += rho[0];
    }
    frame_pvr_error(F, c->last, c->cur, c->e_pvr);
    vbo_huber(quadform(c->e_pvr, c->info_pvr, 9), (double)(float)sqrt(21.666), rho);
    chi += rho[0];
    vbo_edge_bias_error(c->last + 10, c->cur + 10, c->e_bias);
    {
        const double wg = F->inv_bg_rw2 / F->imu_meas[0], wa = F->inv_ba_rw2 / F->imu_meas[0];
        const double *e = c->e_bias;
        vbo_huber(wg * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) + wa * (e[3] * e[3] + e[4] * e[4] + e[5] * e[5]), (double)(float)sqrt(16.812), rho);
        chi += rho[0];
    }
    const double dm = (double)(float)sqrt(5.991);
    for (int i = 0; i < F->n_obs; i++) {
        if (c->lvl[i]) continue;
        frame_mono(F, c->cur, F->obs_pw + 3 * i, F->obs_uv + 2 * i, c->err + 2 * i, NULL);
        const double s = chi2_2(c->err + 2 * i, F->obs_w[i]);
        if (c->vis_robust) { vbo_huber(s, dm, rho); chi += rho[0]; } else chi += s;
    }
    if (F->last_is_frame)
        for (int i = 0; i < F->n_obs_last; i++) {
            if (c->lvl_last[i]) continue;
            frame_mono(F, c->last, F->last_pw + 3 * i, F->last_uv + 2 * i, c->err_last + 2 * i, NULL);
            const double s = chi2_2(c->err_last + 2 * i, F->last_w[i]);
            if (c->vis_robust) { vbo_huber(s, dm, rho); chi += rho[0]; } else chi += s;
        }
    return chi;
}

static void frame_accum(fctx *c, int d, const double *Om, const double *e, double rw, int nb, const int *col, const int *dim, const double *const *J) {
    const int n = c->n;
    double Oe[15];
    for (int a = 0; a < d; a++) {
        double s = 0;
        for (int k = 0; k < d; k++) s += Om[a * d + k] * e[k];
        Oe[a] = s * rw;
    }
    for (int i = 0; i < nb; i++) {
        if (col[i] < 0) continue;
        for (int a = 0; a < dim[i]; a++) {
            double s = 0;
            for (int k = 0; k < d; k++) s += J[i][k * dim[i] + a] * Oe[k];
            c->b[col[i] + a] -= s;
        }
        for (int j = 0; j < nb; j++) {
            if (col[j] < 0) continue;
            for (int a = 0; a < dim[i]; a++)
                for (int bb = 0; bb < dim[j]; bb++) {
                    double s = 0;
                    for (int k = 0; k < d; k++)
                        for (int l = 0; l < d; l++) s += J[i][k * dim[i] + a] * (rw * Om[k * d + l]) * J[j][l * dim[j] + bb];
                    c->H[(col[i] + a) * n + col[j] + bb] += s;
                }
        }
    }
}

static void frame_build(fctx *c) {
    const vba_frame_problem *F = c->F;
    const int n = c->n, lf = F->last_is_frame;
    memset(c->H, 0, sizeof c->H);
    memset(c->b, 0, sizeof c->b);
    double rho[3];
    if (lf == VBA_FRAME_VISION) {
        const double dm = (double)(float)sqrt(5.991);
        for (int i = 0; i < F->n_obs; i++) {
            if (c->lvl[i]) continue;
            double e[2], Jp[6], J[12];
            vbo_edge_se3xyz(F->obs_pw + 3 * i, c->cur, F->K, F->obs_uv + 2 * i, e, NULL, Jp, J);
            double rw = 1.0;
            if (c->vis_robust) { vbo_huber(chi2_2(e, F->obs_w[i]), dm, rho); rw = rho[1]; }
            const double Wt = rw * F->obs_w[i];
            for (int a = 0; a < 6; a++) {
                c->b[a] -= J[a] * Wt * e[0] + J[6 + a] * Wt * e[1];
                for (int bb = 0; bb < 6; bb++) c->H[a * n + bb] += J[a] * Wt * J[bb] + J[6 + a] * Wt * J[6 + bb];
            }
        }
        return;
    }
    if (lf) {
        double J0[135], J1[90], JrI[9];
        memset(J0, 0, sizeof J0);
        memset(J1, 0, sizeof J1);
        vbo_so3_jrinv(c->e_prior + 6, JrI);
        for (int k = 0; k < 3; k++) { J0[9 * k + k] = -1.0; J0[9 * (3 + k) + 3 + k] = -1.0; }
        for (int r = 0; r < 3; r++)
            for (int cc = 0; cc < 3; cc++) J0[9 * (6 + r) + 6 + cc] = JrI[3 * r + cc];
        for (int k = 0; k < 6; k++) J1[6 * (9 + k) + k] = -1.0;
        vbo_huber(quadform(c->e_prior, F->prior_info, 15), (double)(float)sqrt(30.5779), rho);
        const int col[2] = {15, 24}, dim[2] = {9, 6};
        const double *const Js[2] = {J0, J1};
        frame_accum(c, 15, F->prior_info, c->e_prior, rho[1], 2, col, dim, Js);
    }
    {
        double Ji[81], Jj[81], Jb[54];
        frame_pvr_jac(F, c->last, c->cur, c->e_pvr, Ji, Jj, Jb);
        vbo_huber(quadform(c->e_pvr, c->info_pvr, 9), (double)(float)sqrt(21.666), rho);
        const int col[3] = {lf ? 15 : -1, 0, lf ? 24 : -1}, dim[3] = {9, 9, 6};
        const double *const Js[3] = {Ji, Jj, Jb};
        frame_accum(c, 9, c->info_pvr, c->e_pvr, rho[1], 3, col, dim, Js);
    }
    {
        double Om[36], Ji[36], Jj[36];
        memset(Om, 0, sizeof Om); memset(Ji, 0, sizeof Ji); memset(Jj, 0, sizeof Jj);
        const double wg = F->inv_bg_rw2 / F->imu_meas[0], wa = F->inv_ba_rw2 / F->imu_meas[0];
        const double *e = c->e_bias;
        for (int a = 0; a < 6; a++) { Om[7 * a] = a < 3 ? wg : wa; Ji[7 * a] = -1.0; Jj[7 * a] = 1.0; }
        vbo_huber(wg * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) + wa * (e[3] * e[3] + e[4] * e[4] + e[5] * e[5]), (double)(float)sqrt(16.812), rho);
        const int col[2] = {lf ? 24 : -1, 9}, dim[2] = {6, 6};
        const double *const Js[2] = {Ji, Jj};
        frame_accum(c, 6, Om, e, rho[1], 2, col, dim, Js);
    }
    const double dm = (double)(float)sqrt(5.991);
    for (int pass = 0; pass < (lf ? 2 : 1); pass++) {
        const int N = pass ? F->n_obs_last : F->n_obs;
        const unsigned char *lvl = pass ? c->lvl_last : c->lvl;
        const double *pw = pass ? F->last_pw : F->obs_pw, *uv = pass ? F->last_uv : F->obs_uv, *w = pass ? F->last_w : F->obs_w;
        const double *nav = pass ? c->last : c->cur;
        const int c0 = pass ? 15 : 0;
        for (int i = 0; i < N; i++) {
            if (lvl[i]) continue;
            double e[2], J[18];
            frame_mono(F, nav, pw + 3 * i, uv + 2 * i, e, J);
            double rw = 1.0;
            if (c->vis_robust) { vbo_huber(chi2_2(e, w[i]), dm, rho); rw = rho[1]; }
            const double Wt = rw * w[i];
            for (int a = 0; a < 9; a++) {
                c->b[c0 + a] -= J[a] * Wt * e[0] + J[9 + a] * Wt * e[1];
                for (int bb = 0; bb < 9; bb++) c->H[(c0 + a) * n + c0 + bb] += J[a] * Wt * J[bb] + J[9 + a] * Wt * J[9 + bb];
            }
        }
    }
}

/* LinearSolverCholmod::solve: L L^T, "not positive definite" -> false (linear_solver_cholmod.h:86-125) */
static int chol_solve_dense(int n, const double *A, const double *rhs, double *x) {
    double L[900];
    for (int j = 0; j < n; j++) {
        double d = A[j * n + j];
        for (int k = 0; k < j; k++) d -= L[j * n + k] * L[j * n + k];
        if (!(d > 0.0) || !isfinite(d)) return 0;
        d = sqrt(d);
        L[j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[i * n + j];
            for (int k = 0; k < j; k++) s -= L[i * n + k] * L[j * n + k];
            L[i * n + j] = s / d;
        }
    }
    double y[30];
    for (int i = 0; i < n; i++) {
        double s = rhs[i];
        for (int k = 0; k < i; k++) s -= L[i * n + k] * y[k];
        y[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = y[i];
        for (int k = i + 1; k < n; k++) s -= L[k * n + i] * x[k];
        x[i] = s / L[i * n + i];
    }
    return 1;
}

static int frame_lm(fctx *c, int iterations, double *chi_out) {
    const int n = c->n;
    int cj = 0, nBad = 0;
    double lambda = 0, ni = 2, cur = 0;
    for (int it = 0; it < iterations; it++) {
        cur = frame_errors(c);
        const double iniChi = cur;
        double tempChi = cur;
        frame_build(c);
        memcpy(c->Hlast, c->H, sizeof c->H);
        if (it == 0) {
            double mx = 0;
            for (int i = 0; i < n; i++) mx = fmax(fabs(c->H[i * n + i]), mx);
            lambda = 1e-5 * mx;
            ni = 2;
            nBad = 0;
        }
        double rho = 0;
        int qmax = 0;
        do {
            memcpy(c->cur_bk, c->cur, sizeof c->cur);
            memcpy(c->last_bk, c->last, sizeof c->last);
            double A[900];
            memcpy(A, c->H, sizeof A);
            for (int i = 0; i < n; i++) A[i * n + i] += lambda;
            const int ok2 = chol_solve_dense(n, A, c->b, c->x);
            if (ok2) {
                if (n == 6) { /* VertexSE3Expmap::oplusImpl: SE3Quat::exp(update) * estimate */
                    double E[7];
                    vbo_se3_exp(c->x, E);
                    se3_mul(E, c->cur, c->cur);
                } else {
                    nav_oplus(c->cur, c->x, c->x + 9);
                    if (n == 30) nav_oplus(c->last, c->x + 15, c->x + 24);
                }
            }
            tempChi = frame_errors(c);
            if (!ok2) tempChi = DBL_MAX;
            rho = cur - tempChi;
            double scale = 0;
            for (int j = 0; j < n; j++) scale += c->x[j] * (lambda * c->x[j] + c->b[j]);
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
                memcpy(c->cur, c->cur_bk, sizeof c->cur);
                memcpy(c->last, c->last_bk, sizeof c->last);
            }
            qmax++;
        } while (rho < 0 && qmax < 10);
        ++cj;
        if (qmax == 10 || rho == 0) break;
        if ((iniChi - cur) * 1e3 < iniChi) nBad++;
        else nBad = 0;
        if (nBad >= 3) break;
    }
    *chi_out = cur;
    return cj;
}

static void dense_inverse(int n, const double *A, double *Ai) { /* Gauss-Jordan with partial pivoting */
    double M[30][60];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) { M[i][j] = A[i * n + j]; M[i][n + j] = (i == j) ? 1.0 : 0.0; }
    for (int cc = 0; cc < n; cc++) {
        int p = cc;
        for (int r = cc + 1; r < n; r++)
            if (fabs(M[r][cc]) > fabs(M[p][cc])) p = r;
        if (p != cc)
            for (int j = 0; j < 2 * n; j++) { const double t = M[cc][j]; M[cc][j] = M[p][j]; M[p][j] = t; }
        const double inv = 1.0 / M[cc][cc];
        for (int j = 0; j < 2 * n; j++) M[cc][j] *= inv;
        for (int r = 0; r < n; r++) {
            if (r == cc) continue;
            const double f = M[r][cc];
            if (f == 0.0) continue;
            for (int j = 0; j < 2 * n; j++) M[r][j] -= f * M[cc][j];
        }
    }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) Ai[i * n + j] = M[i][n + j];
}

int vba_oracle_pose_optimize(vba_frame_problem *F, vba_frame_result *out) {
    memset(out->its_done, 0, sizeof out->its_done);
    memset(out->chi2_round, 0, sizeof out->chi2_round);
    memset(out->marg_cov_inv, 0, sizeof out->marg_cov_inv);
    out->status = VBA_OK;
    out->n_inliers = 0;
    for (int i = 0; i < F->n_obs; i++) out->outlier[i] = 0; /* pFrame->mvbOutlier[i] = false, :2147 */
    if (F->n_obs < 3) return 0;                              /* nInitialCorrespondences < 3, :2178 */
    fctx C, *c = &C;
    memset(c, 0, sizeof C);
    c->F = F;
    const int vision = F->last_is_frame == VBA_FRAME_VISION, lif = F->last_is_frame == VBA_FRAME_FRAME;
    c->n = vision ? 6 : (lif ? 30 : 15);
    c->lvl = calloc(F->n_obs + 1, 1);
    c->err = calloc(2 * F->n_obs + 2, 8);
    c->lvl_last = calloc(F->n_obs_last + 1, 1);
    c->err_last = calloc(2 * F->n_obs_last + 2, 8);
    if (!vision) dense_inverse(9, F->imu_cov_pvphi, c->info_pvr);       /* Matrix9d InvCovPVR = imupreint.getCovPVPhi().inverse() */
    if (lif && out->outlier_last)
        for (int i = 0; i < F->n_obs_last; i++) out->outlier_last[i] = 0;
    c->vis_robust = 1;
    int nBad = 0;
    for (int it = 0; it < 4; it++) {
        memcpy(c->cur, F->nav, sizeof c->cur);             /* setEstimate(pFrame->GetNavState()) before every round */
        memcpy(c->last, F->nav_last, sizeof c->last);
        out->its_done[it] = frame_lm(c, 10, &out->chi2_round[it]);
        const double dm2 = 5.991;
        for (int pass = 0; pass < (lif ? 2 : 1); pass++) {
            const int N = pass ? F->n_obs_last : F->n_obs;
            unsigned char *lvl = pass ? c->lvl_last : c->lvl;
            double *err = pass ? c->err_last : c->err;
            int bad = 0;
            for (int i = 0; i < N; i++) {
                if (lvl[i] && vision)
                    vbo_edge_se3xyz(F->obs_pw + 3 * i, c->cur, F->K, F->obs_uv + 2 * i, err + 2 * i, NULL, NULL, NULL);
                else if (lvl[i]) /* outliers are outside the active set: their error is recomputed at the final estimate */
                    frame_mono(F, pass ? c->last : c->cur, (pass ? F->last_pw : F->obs_pw) + 3 * i, (pass ? F->last_uv : F->obs_uv) + 2 * i, err + 2 * i, NULL);
                const float chi2 = (float)chi2_2(err + 2 * i, (pass ? F->last_w : F->obs_w)[i]); /* const float chi2 = e->chi2() */
                if (chi2 > (float)dm2) { lvl[i] = 1; bad++; } else lvl[i] = 0;
            }
            if (!pass) nBad = bad;
        }
        if (it == 2) c->vis_robust = 0;                    /* e->setRobustKernel(0) */
        if (F->n_obs + (lif ? F->n_obs_last + 1 : 0) + (vision ? 0 : 2) < 10) break; /* optimizer.edges().size() < 10 */
    }
    if (vision) memcpy(F->nav, c->cur, 7 * sizeof(double)); /* pFrame->SetPose(Converter::toCvMat(SE3quat_recov)) */
    else {
        memcpy(F->nav, c->cur, 10 * sizeof(double));            /* P, R, V of the PVR vertex */
        memcpy(F->nav + 16, c->cur + 16, 6 * sizeof(double));   /* dbg, dba of the bias vertex */
    }
    for (int i = 0; i < F->n_obs; i++) out->outlier[i] = c->lvl[i];
    if (lif && out->outlier_last)
        for (int i = 0; i < F->n_obs_last; i++) out->outlier_last[i] = c->lvl_last[i];
    out->n_inliers = F->n_obs - nBad;
    if (F->compute_marg && !vision) {
        /* computeMarginals on the Hessian of the last linearisation (lambda already restored) */
        double Hi[900];
        dense_inverse(c->n, c->Hlast, Hi);
        if (!lif) {
            /* margCovInv = blockdiag(spinv(0,0)^-1, spinv(1,1)^-1), :2251-2253 */
            double A[81], Ai[81], B[36], Bi[36];
            for (int i = 0; i < 9; i++)
                for (int j = 0; j < 9; j++) A[9 * i + j] = Hi[i * c->n + j];
            for (int i = 0; i < 6; i++)
                for (int j = 0; j < 6; j++) B[6 * i + j] = Hi[(9 + i) * c->n + 9 + j];
            dense_inverse(9, A, Ai);
            dense_inverse(6, B, Bi);
            for (int i = 0; i < 9; i++)
                for (int j = 0; j < 9; j++) out->marg_cov_inv[15 * i + j] = Ai[9 * i + j];
            for (int i = 0; i < 6; i++)
                for (int j = 0; j < 6; j++) out->marg_cov_inv[15 * (9 + i) + 9 + j] = Bi[6 * i + j];
        } else {
            /* the joint 15x15 marginal of the frame, inverted (:2011-2018; the reference asks g2o for the diagonal
             * blocks only and then reads the off-diagonal ones too -- the intent, the full joint marginal, is restated) */
            double A[225];
            for (int i = 0; i < 15; i++)
                for (int j = 0; j < 15; j++) A[15 * i + j] = Hi[i * c->n + j];
            dense_inverse(15, A, out->marg_cov_inv);
        }
    }
    free(c->lvl); free(c->err); free(c->lvl_last); free(c->err_last);
    return 0;
}

/* test hook: robust chi2, H and b of the frame problem at F->nav / F->nav_last (all edges active, kernels on) */
int vba_oracle_frame_linearize(vba_frame_problem *F, double *H, double *b, double *chi) {
    fctx C, *c = &C;
    memset(c, 0, sizeof C);
    c->F = F;
    c->n = F->last_is_frame == VBA_FRAME_VISION ? 6 : (F->last_is_frame ? 30 : 15);
    c->lvl = calloc(F->n_obs + 1, 1);
    c->err = calloc(2 * F->n_obs + 2, 8);
    c->lvl_last = calloc(F->n_obs_last + 1, 1);
    c->err_last = calloc(2 * F->n_obs_last + 2, 8);
    if (F->last_is_frame != VBA_FRAME_VISION) dense_inverse(9, F->imu_cov_pvphi, c->info_pvr);
    c->vis_robust = 1;
    memcpy(c->cur, F->nav, sizeof c->cur);
    memcpy(c->last, F->nav_last, sizeof c->last);
    *chi = frame_errors(c);
    if (H) {
        frame_build(c);
        memcpy(H, c->H, sizeof(double) * c->n * c->n);
        memcpy(b, c->b, sizeof(double) * c->n);
    }
    free(c->lvl); free(c->err); free(c->lvl_last); free(c->err_last);
    return c->n;
}
void vbo_nav_oplus(double *nav, const double *dpvr, const double *dbias) { nav_oplus(nav, dpvr, dbias); }
