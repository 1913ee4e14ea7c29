// vba_device.h -- device-side data layout and small Lie-group helpers for the gfx950 local-BA backend.
//
// Everything is IEEE double: the reference optimises in double and its IMU
// information matrices reach 1/(2e-5)^2 = 2.5e9 (src/IMU/imudata.cpp:25), so FP32 Jacobians are not an option.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VBA_NB 32          // block size of the dense reduced-system factorisation
#define VBA_EREC 18        // doubles per edge record, XYZ variants (144 B): Bi (2x6), g = -Bi^T r
#define VBA_EREC1 8        // doubles per edge record, inverse-depth variant (64 B): P_c (3), sqrt(rho' w) (1), r (2); the
                           // readers rebuild Bi = [A | B_rot] from it and the observer's rotation (rebuild_edge)
#define VBA_PREC 32        // doubles per point record  (256 B)
#define VBA_SLOT 8         // doubles per slot record   (64 B = one line), inverse-depth landmarks
#define VBA_SLOT3 18       // doubles per slot record, XYZ landmarks (144 B): W = Bi^T A (6x3), independent of the damping
#define VBA_IMUH 960       // doubles per IMU edge pair: 30x30 local Hessian + 30 rhs (+ pad)
#define VBA_TRACE 64

// Linearisation products kept in HBM between k_lin2 and its consumers (variant 2, EdgePRIDP).  A "slot" is one
// (landmark, keyframe) incidence = one H_pl block: observation e -> slot e, reference keyframe of landmark p
// -> slot n_obs + p.  Jacobians are pre-scaled by sqrt(rho' * invSigma2) and never stored unreduced.
//   slot record  [0..5] U = W * sqrt(Dinv)  (W = H_pl block, 6x1)   [6] beta = sqrt(Dinv) * b_l   [7] sqrt(Dinv) (ref slot)
//                -> Schur term of a keyframe pair is -U_a U_b^T, reduced rhs term -U_a beta, x_l = sD (beta - sum U.x_p)
//   edge record  [0..11] Bi (2x6, d/d observing KF PR, g2otypes.cpp:139-145)  [12..23] Br (2x6, d/d reference KF PR,
//                :128-134)  [24..29] g = -Bi^T r
//   point record [0..20] G0 = sum Br^T Br (upper 6x6 packed)  [21..26] g0 = -sum Br^T r  [27] D
struct WinDesc {
    int variant, algo;
    int n_kf, n_free, n_pt, n_obs, n_imu;
    int pdim, np, nS, nb;
    int its[2];
    int kf0, pt0, obs0, imu0;
    int pair0, n_pairs;
    int item0;      // offset into the item array
    int pimu0;      // offset into the pair-imu list
    int vec0;       // offset into rhs/x vectors (nS slots per window)
    int part0;      // offset into the chi2 partial array
    int n_part_lin; // workgroups of this window in the linearise launch (= chi2 partials; XYZ: also the max-diagonal partials)
    int n_part_pt;  // 64-landmark blocks of the window (= the computeScale partials of k_update_xyz)
    int lin_runs;   // 1: the window has the work split of the edge-parallel linearisation (lin_blk); 0: thread-per-landmark fallback
    int tl_step0;   // offset of this window's step_begin / pan_begin rows (nb + 1 entries each)
    int tl_pair0;   // offset into the tile-pair list
    int tl_pan0;    // offset into the panel-tile list
    int lb0;        // first record of the window in the k_lin2 run table
    int win;        // index of this window in the uploaded batch: the CSR-style tables (pt_obs_begin, item_begin,
                    // pimu_begin) carry one extra entry per window, so their rows start at offset + win
    int tl_kb0;     // offset of this window's column-entry table of the left-looking factorisation (pan entries + nb + 1)
    int tl_k0;      // offset into its k lists
    int order;      // elimination order: 0 = V/Bias blocks first, 1 = keyframe by keyframe
    int nc;         // chain columns: block columns [0, nc) are factored by k_chol_chain, the per-column kernels start at nc
    int ct0;        // offset (records of four ints) of the window's chain-column table (Structure::chain_tab)
    int cu0, n_cu;  // few-window regime: the window's tiles that collect updates from chain columns (k_chol_chain_upd)
    int vp_pr0, vp_prs, vp_vb0, vp_vbs;  // position of dof r of free keyframe a: r < 6 ? pr0 + prs a + r : vb0 + vbs a + r - 6
    int vp_h, vp_vb1;   // order 2 (two-sided): the V/Bias block of keyframe a >= vp_h sits at vp_vb1 - 9 a (other orders: vp_h = INT_MAX)
    int pad0[3], padn[3];  // rows of S that belong to no variable (identity): up to three ranges (order 2 pads each chain and the tail)
    int nc_split;       // > 0: the chain columns [0, nc_split) and [nc_split, nc) are independent (k_chol_chain_rows walks them side by side)
    long long S0;   // offset (doubles) into S
    long long mask0; // offset (64-bit words) of the window's landmark masks (n_pt x mwords)
    int mwords;      // 64-bit words per landmark mask = ceil(n_kf / 64)
    int adj0;        // offset into the keyframe adjacency list (PCG)
    double K[4];
    double Rcb[9], tcb[3], g[3];
    double inv_bg, inv_ba;
    double hub_vis, hub_prv, hub_bias;
    double chi2_th, depth_min, rho_min;
    int protocol;    // VBA_PROTO_*: 1 = one optimize(its[0]) and no outlier pass (global BA)
    int robust;      // protocol 1: Huber on every edge, or on none
};

struct WinCtrl {
    int stage;        // 0,1
    int it;           // outer iteration inside the stage
    int active;       // 1 while the stage's optimize() loop is still running for this window
    int status;
    int its_done[2];
    int robust_vis;   // Huber on vision edges (stage 1)
    int chol_fail;    // set by the factorisation of the current iteration
    int aborted;      // stop flag seen
    int n_trace;
    int n_outliers;
    // LM state (levenberg.cpp)
    int lm_trial;     // trials done in the current outer iteration (qmax)
    int lm_need_trial;// 1: another trial must run in this outer iteration
    int lm_restore;   // 1: the last trial was rejected, k_restore must pop the state
    int nbad;
    int lin_its;      // PCG iterations of all solves so far
    int polls;        // terminate() polls of this window so far (test hook vba_debug_set_stop_after; oracle twin: stop_now)
    double lambda, ni;
    double chi_prev;  // GN: preChi2 of the last started iteration.  LM: currentChi
    double chi_ini;   // LM: iniChi
    double chi2_vis, chi2_prv, chi2_bias;
    double trace[VBA_TRACE];
};

// ------------------------------------------------------------------------------------------------
// small math (all __device__ __forceinline__, row-major 3x3)
// ------------------------------------------------------------------------------------------------
#define DEVI __device__ __forceinline__

DEVI void q2R(const double* q, double* R) {  // q = x,y,z,w  (Eigen::Quaterniond::toRotationMatrix)
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
DEVI void R2q(const double* m, double* q) {  // Eigen quaternion-from-matrix
    double t = m[0] + m[4] + m[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m[7] - m[5]) * t; q[1] = (m[2] - m[6]) * t; q[2] = (m[3] - m[1]) * t;
    } else {
        // Eigen: i = 0; if (m(1,1) > m(0,0)) i = 1; if (m(2,2) > m(i,i)) i = 2; j = (i+1)%3; k = (j+1)%3 -- spelled out per case with
        // constant indices (indexing m / q with the run-time i, j, k put the callers' matrices into scratch memory); same arithmetic
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > ((i == 1) ? m[4] : m[0])) i = 2;
        if (i == 0) {           // j = 1, k = 2
            t = sqrt(m[0] - m[4] - m[8] + 1.0);
            q[0] = 0.5 * t;
            t = 0.5 / t;
            q[3] = (m[7] - m[5]) * t; q[1] = (m[3] + m[1]) * t; q[2] = (m[6] + m[2]) * t;
        } else if (i == 1) {    // j = 2, k = 0
            t = sqrt(m[4] - m[8] - m[0] + 1.0);
            q[1] = 0.5 * t;
            t = 0.5 / t;
            q[3] = (m[2] - m[6]) * t; q[2] = (m[7] + m[5]) * t; q[0] = (m[1] + m[3]) * t;
        } else {                // j = 0, k = 1
            t = sqrt(m[8] - m[0] - m[4] + 1.0);
            q[2] = 0.5 * t;
            t = 0.5 / t;
            q[3] = (m[3] - m[1]) * t; q[0] = (m[2] + m[6]) * t; q[1] = (m[5] + m[7]) * t;
        }
    }
}
DEVI void qmul(const double* a, const double* b, double* o) {
    const double w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    const double x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    const double y = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    const double z = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}
DEVI void qnorm(double* q) {
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
DEVI void qrot(const double* q, const double* v, double* o) {  // Eigen _transformVector
    double ux = q[1] * v[2] - q[2] * v[1], uy = q[2] * v[0] - q[0] * v[2], uz = q[0] * v[1] - q[1] * v[0];
    ux += ux; uy += uy; uz += uz;
    const double cx = q[1] * uz - q[2] * uy, cy = q[2] * ux - q[0] * uz, cz = q[0] * uy - q[1] * ux;
    const double r0 = v[0] + q[3] * ux + cx, r1 = v[1] + q[3] * uy + cy, r2 = v[2] + q[3] * uz + cz;
    o[0] = r0; o[1] = r1; o[2] = r2;
}
DEVI void mm3(const double* A, const double* B, double* C) {  // C = A B (C must not alias)
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
DEVI void mtm3(const double* A, const double* B, double* C) {  // C = A^T B
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) C[3 * i + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}
DEVI void mv3(const double* A, const double* v, double* o) {
    const double a = A[0] * v[0] + A[1] * v[1] + A[2] * v[2];
    const double b = A[3] * v[0] + A[4] * v[1] + A[5] * v[2];
    const double c = A[6] * v[0] + A[7] * v[1] + A[8] * v[2];
    o[0] = a; o[1] = b; o[2] = c;
}
DEVI void mtv3(const double* A, const double* v, double* o) {  // A^T v
    const double a = A[0] * v[0] + A[3] * v[1] + A[6] * v[2];
    const double b = A[1] * v[0] + A[4] * v[1] + A[7] * v[2];
    const double c = A[2] * v[0] + A[5] * v[1] + A[8] * v[2];
    o[0] = a; o[1] = b; o[2] = c;
}
DEVI void hat3(const double* v, double* M) {
    M[0] = 0; M[1] = -v[2]; M[2] = v[1];
    M[3] = v[2]; M[4] = 0; M[5] = -v[0];
    M[6] = -v[1]; M[7] = v[0]; M[8] = 0;
}
DEVI double nrm3(const double* v) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

// Sophus::SO3::exp (src/IMU/so3.cpp:237-261): quaternion, normalised
DEVI void so3exp(const double* w, double* q) {
    const double th = nrm3(w), h = 0.5 * th;
    double im;
    if (th < 1e-10) {
        const double t2 = th * th;
        im = 0.5 - 0.0208333 * t2 + 0.000260417 * t2 * t2;
    } else
        im = sin(h) / th;
    q[0] = im * w[0]; q[1] = im * w[1]; q[2] = im * w[2]; q[3] = cos(h);
    qnorm(q);
}
// Sophus::SO3::log (so3.cpp:190-228)
DEVI void so3log(const double* q, double* w) {
    const double n = nrm3(q), ww = q[3];
    const double f = (n < 1e-10) ? (2. / ww - 2. * (n * n) / (ww * ww * ww)) : (2 * atan(n / ww) / n);
    w[0] = f * q[0]; w[1] = f * q[1]; w[2] = f * q[2];
}
// SO3::JacobianR / JacobianRInv (so3.cpp:33-72)
DEVI void so3jr(const double* w, double* J) {
    const double th = nrm3(w);
    J[0] = 1; J[1] = 0; J[2] = 0; J[3] = 0; J[4] = 1; J[5] = 0; J[6] = 0; J[7] = 0; J[8] = 1;
    if (th < 0.00001) return;
    const double k[3] = {w[0] / th, w[1] / th, w[2] / th};
    double K[9], K2[9];
    hat3(k, K);
    mm3(K, K, K2);
    const double a = (1 - cos(th)) / th, b = 1 - sin(th) / th;
#pragma unroll
    for (int i = 0; i < 9; i++) J[i] = J[i] - a * K[i] + b * K2[i];
}
DEVI void so3jrinv(const double* w, double* J) {
    const double th = nrm3(w);
    J[0] = 1; J[1] = 0; J[2] = 0; J[3] = 0; J[4] = 1; J[5] = 0; J[6] = 0; J[7] = 0; J[8] = 1;
    if (th < 0.00001) return;
    const double k[3] = {w[0] / th, w[1] / th, w[2] / th};
    double K[9], K2[9], W[9];
    hat3(k, K);
    hat3(w, W);
    mm3(K, K, K2);
    const double c = 1.0 - (1.0 + cos(th)) * th / (2.0 * sin(th));
#pragma unroll
    for (int i = 0; i < 9; i++) J[i] = J[i] + 0.5 * W[i] + c * K2[i];
}
// SO3 product with Sophus' normalisation pattern (so3.cpp:93-96,127-133)
DEVI void so3mul(const double* a, const double* b, double* o) {
    double t[4] = {a[0], a[1], a[2], a[3]};
    qnorm(t);
    double r[4];
    qmul(t, b, r);
    qnorm(r);
    o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = r[3];
}
DEVI void so3inv(const double* a, double* o) {
    o[0] = -a[0]; o[1] = -a[1]; o[2] = -a[2]; o[3] = a[3];
    qnorm(o);
}
// Position of local dof r of free keyframe a inside the reduced system.  VI variants order all V/Bias
// blocks first (9 per keyframe, chain order) and all PR blocks last: the V/Bias part of the factor then
// stays block-banded and whole 32x32 tiles of L are structurally zero (skipped by the tile lists).
DEVI int vpos(const WinDesc& d, int a, int r) {
    return r < 6 ? d.vp_pr0 + d.vp_prs * a + r : (a < d.vp_h ? d.vp_vb0 + d.vp_vbs * a : d.vp_vb1 - 9 * a) + (r - 6);
}
DEVI double rl64(double v, int lane) {  // wave-uniform broadcast of one lane's double
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
typedef double d4_t __attribute__((ext_vector_type(4)));

// Sums inside a wave without the LDS crossbar: data-parallel-primitive moves (quad permutes, then the two row mirrors) add up the 16
// lanes of a row in four steps of plain VALU work; `__shfl_xor` costs two ds_bpermute per double and step.  Fixed order.
template <int CTRL>
DEVI double dpp_f64(double v) {
    const int l = __double2loint(v), h = __double2hiint(v);   // (every lane has a source under these controls: `old` is never used)
    const int lo = __builtin_amdgcn_update_dpp(l, l, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(h, h, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
DEVI double row16_sum(double v) {   // every lane of a 16-lane row ends up with the row's sum
    v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);   // row_half_mirror
    v += dpp_f64<0x140>(v);   // row_mirror
    return v;
}
DEVI double quad_sum(double v) {    // every lane of a quad ends up with the quad's sum
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    return v;
}
DEVI double wave64_sum(double v) {  // every lane ends up with the wave's sum: rows by DPP, the four row sums through scalar reads
    v = row16_sum(v);
    return ((rl64(v, 0) + rl64(v, 16)) + rl64(v, 32)) + rl64(v, 48);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a fence over every address space: it drains the wave's
// global loads and stores (s_waitcnt vmcnt(0)) before the barrier, so nothing fetched for LATER can be in flight across it and
// every store issued before it is paid for in full.  Where the barrier only hands LDS data from wave to wave, this form keeps the
// vector-memory queue running.  (Global data written before and read after it by ANOTHER wave still needs __syncthreads().)
DEVI void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// The same for code that ONE wave runs on its own (LDS written by some lanes, read by others of the same wave): the wave's LDS
// operations complete in order, so draining them is all it takes -- no s_barrier, hence usable where other waves of the workgroup
// have already left.
DEVI void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Huber (robust_kernel_impl.cpp:78-91): returns rho(e), sets *w = rho'(e)
DEVI double huber(double e, double delta, double* w) {
    const double dsqr = delta * delta;
    if (e <= dsqr) { *w = 1.0; return e; }
    const double s = sqrt(e);
    *w = delta / s;
    return 2 * s * delta - dsqr;
}

// deterministic block-wide sum of one double per thread (fixed tree order); result valid in thread 0
template <int NT>
DEVI double block_sum(double v, double* sm) {
    const int t = threadIdx.x;
    sm[t] = v;
    __syncthreads();
#pragma unroll
    for (int s = NT / 2; s > 0; s >>= 1) {
        if (t < s) sm[t] += sm[t + s];
        __syncthreads();
    }
    const double r = sm[0];
    __syncthreads();
    return r;
}

// the same for a 256-thread workgroup with 4 doubles of LDS: fixed butterfly inside each wave, then the four
// wave sums in order
DEVI double block_sum256(double v, double* sm4) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int t = threadIdx.x;
    __syncthreads();
    if ((t & 63) == 0) sm4[t >> 6] = v;
    __syncthreads();
    return ((sm4[0] + sm4[1]) + sm4[2]) + sm4[3];
}
