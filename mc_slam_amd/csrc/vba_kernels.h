// vba_kernels.h -- hand-written HIP kernels (gfx950, wave64) of the local-BA hot path.
//
// One launch = one phase of one outer iteration for EVERY window of the batch (lock-step); a window whose
// optimize() loop has terminated makes its workgroups exit at the first instruction.  All reductions use a
// fixed order (no floating-point atomics): results are bit-reproducible run to run.
#pragma once
#include "vba_device.h"

struct Batch {
    const WinDesc* desc;
    WinCtrl* ctrl;
    int n_win;
    // keyframe state: working copy, uploaded initial copy, LM backup, and the R|t cache the edges read
    double *pose, *vel, *bias, *kfR;
    const double *pose0, *vel0, *bias0;
    double *pose_bk, *vel_bk, *bias_bk;
    // landmarks
    double *pt;
    const double* pt0;
    double* pt_bk;
    const int *pt_ref, *pt_obs_begin;
    // observations (CSR by point)
    const int *obs_kf, *obs_pt;
    const double *obs_uv, *obs_w;
    unsigned char* lvl;
    double *chi2_e, *depth_e, *chi2_f;  // chi2_f: chi2 recomputed at the final estimates (LM: chi2_e may be stale)
    double *erec, *prec, *slot;
    const int* slot_perm;         // [n_obs] record position of every observation edge (keyframe-major for inverse-depth windows)
    const int* pt_perm;           // [n_pt] record position of every landmark (grouped by reference keyframe)
    const unsigned char* kf_fix;  // [n_kf] per-vertex setFixed() of listed-free keyframes: bit0 PR, bit1 V, bit2 Bias
    // IMU factors
    const int *imu_i, *imu_j;
    const double *imu_meas, *imu_info;
    double *imuH, *imu_chi;  // imu_chi: [4] per edge: robust prv, robust bias, raw prv, raw bias
    double* imu_jrec;        // [IMU_JREC] per edge: Jacobian, weighted information, errors of the last linearisation (lin_imu_res -> lin_imu_hess)
    // reduced system
    double *S, *vec, *bpose;
    double *Lf, *yv;  // factor tiles and forward-substituted rhs (written out of place: S tiles are read by
                      // other workgroups of the same launch)
    int l_packed;     // Lf holds the factor tile by tile in MFMA operand order (left-looking kernels, ll_pk) instead of row-major
    int* var_act;
    // structure (g2o buildStructure analogue, built on the host at upload)
    const int *pair_a, *pair_b, *item_begin, *items, *pimu_begin, *pimu;
    const int *adj_begin, *adj;         // PCG: per free keyframe the other free keyframes it shares a landmark or an IMU edge with
    double* kf_dir;                     // XYZ landmarks: per keyframe the damping-independent part of its diagonal block and b_p (32 doubles)
    double *pcg_v, *pcg_m, *pcg_s;      // PCG: x r z p q (5 nS per window); preconditioner blocks (450 per keyframe); CG state (8 per window)
    int pcg_tri;                        // PCG: block-tridiagonal (keyframe chain) preconditioner instead of block-Jacobi
    const int *slot_o, *rec_lm;         // [n_obs] observation of the record in every slot; [n_pt] landmark of every landmark record
    const int* slot_lm;                 // [n_obs] landmark of the record in every slot (XYZ gathers fetch the landmark's Sigma through it)
    const int* item_mid;                // per pair: its first item that involves the landmark's reference keyframe
    const unsigned long long* lmask;    // [n_pt x mwords] observing keyframes of every landmark (host-built while validating)
    const int *kf_seg, *ref_seg;        // [n_kf + 1] per window: record range of every keyframe -- slot / edge records by observing
                                        // keyframe, landmark records by reference keyframe (the items of the diagonal pair (a,a))
    const int *off_pair, *pair_mask;  // off-diagonal pair indices; per pair: which sub-blocks of S are ever read
    const int* lin_blk;  // k_lin2: per workgroup its run of landmarks and edges (p0, p1, e0, e1), n_part_lin records per window
    // inverse depth: the reference-keyframe terms (G0, g0) leave k_lin2 summed over runs of consecutive landmarks with one reference
    // keyframe -- prun0[workgroup]: id of its first run record; per keyframe (rows at kf0 + win) the run records it is the
    // reference of: pref_list[pt0 + pref_begin[k] .. pref_begin[k + 1])
    const int *prun0, *pref_begin, *pref_list;
    // tile structure of the factor (symbolic factorisation on 32x32 tiles, built at upload)
    const int *tl_step_begin, *tl_pairs, *tl_pan_begin, *tl_pan;
    const int *tl_kl_begin, *tl_kl;  // left-looking factorisation: per column entry (J,J),(I,J).. the steps k < J that update it
    const int* tl_ct;                // per chain column: row mask (two words), rides, 0 (Structure::chain_tab)
    const int* tl_cu;                // few-window regime: tiles behind the chain columns that collect updates from them (I << 16 | J, first, end of the chain columns in its k list, 0)
    double *dvec, *winv;             // D of the factor (nS per window); L_JJ^-T D_J^-1 of the current step (32x32 per window)
    int w_total, w_stride;           // behind those: W_J of every chain column, w_stride tiles per window (batch-wide window index d.win)
    double* part;
    const volatile int* stop_word;   // DEVICE copy of the caller's stop flag (word 1023 of the group's mirror words), refreshed by k_poll_stop
                                     // before every control launch: thousands of windows reading the pinned host word cost 0.5 ms per launch
    const volatile int* stop_host_word;   // the pinned host word itself
    int* alive_cnt;  // pinned host words: [stage * 32 + it] = 1 if a window is still iterating after control call `it`
    int* alive_dev;  // device mirror of the group's words ([0,64): Gauss-Newton slots, [64,1024): LM slot groups), zeroed at run start:
                     // only the FIRST window that flips a mirror word writes the host word (thousands of windows posting the same
                     // 4 bytes over PCIe cost 0.4 ms per control launch)
    unsigned char* out_outlier;
    double* out_chi2;
    double* dbg;  // 4 KiB scratch for diagnostic builds (in-kernel stamps); never read by the product path
    int dbg_stop_after;  // test hook (vba_debug_set_stop_after): >= 0 -- the stop flag reads 1 from that poll of a window on; -1: off
};

#define LIN_FULL 0
#define LIN_ERR 1

// One read of the caller's stop flag by one window: g2o's terminate() before an iteration (sparse_optimizer.cpp:376), inside the
// Levenberg-Marquardt trial loop (levenberg.cpp:149), and the bDoMore check between the stages (src/Optimizer.cpp:462-466).  The
// polls of a window are counted from the first terminate() of its first optimize() so that a test can raise the flag at a chosen
// poll (dbg_stop_after) -- the deterministic stand-in for LocalMapping::InterruptBA firing in the middle of a solve; the oracle
// counts the same polls (stop_now in oracle/vba_oracle.c).  Thread 0 of the window's control workgroup only.
DEVI int poll_stop(const Batch& B, WinCtrl& c) {
    const int n = c.polls++;
    return ((B.stop_word && *B.stop_word) || (B.dbg_stop_after >= 0 && n >= B.dbg_stop_after)) ? 1 : 0;
}

// window takes part in the current solve: GN -> while its optimize() loop runs; LM -> only while a trial is due
DEVI bool win_on(const WinDesc& d, const WinCtrl& c) { return c.active && (d.algo == 0 || c.lm_need_trial); }

// ------------------------------------------------------------------------------------------------
// K_reset: working state <- uploaded state, R|t cache, control block
// ------------------------------------------------------------------------------------------------
DEVI void kf_cache(const Batch& B, const WinDesc& d, int a) {
    const double* T = B.pose + 7 * (size_t)(d.kf0 + a);
    double* C = B.kfR + 12 * (size_t)(d.kf0 + a);
    double R[9];
    q2R(T + 3, R);
#pragma unroll
    for (int i = 0; i < 9; i++) C[i] = R[i];
    C[9] = T[0]; C[10] = T[1]; C[11] = T[2];
}

// free vertices of keyframe kf: bit0 PR, bit1 V, bit2 Bias (0 for the keyframes listed as fixed)
DEVI int kf_free(const Batch& B, const WinDesc& d, int kf) {
    if (kf >= d.n_free) return 0;
    return 7 & ~(int)B.kf_fix[d.kf0 + kf];
}
// allVerticesFixed edges are dropped (sparse_optimizer.cpp:236): bit0 EdgeNavStatePRV (PR_i PR_j V_i V_j Bias_i),
// bit1 EdgeNavStateBias (Bias_i Bias_j)
DEVI int imu_act(const Batch& B, const WinDesc& d, int i, int j) {
    const int fi = kf_free(B, d, i), fj = kf_free(B, d, j);
    return ((fi | (fj & 3)) ? 1 : 0) | (((fi | fj) & 4) ? 2 : 0);
}
DEVI bool imu_robust(const WinDesc& d) { return !(d.protocol == 1 && !d.robust); }

// upload: identity on the padded diagonal of S (rows np..nS, fewer than VBA_NB of them); the solve never touches the pads
__global__ void __launch_bounds__(64) k_init_pads(Batch B) {
    const WinDesc& d = B.desc[blockIdx.x];
#pragma unroll
    for (int q = 0; q < 3; q++) {
        const int i = d.pad0[q] + threadIdx.x;
        if ((int)threadIdx.x < d.padn[q]) B.S[d.S0 + (size_t)i * d.nS + i] = 1.0;
    }
}

// one PCIe read per control launch instead of one per window
__global__ void k_poll_stop(Batch B) { *const_cast<int*>(B.stop_word) = *B.stop_host_word; }

__global__ void __launch_bounds__(256) k_reset(Batch B) {
    const int w = blockIdx.y;
    const WinDesc& d = B.desc[w];
    // flat, coalesced copies: consecutive threads take consecutive doubles of each array (the first version gave every thread a
    // whole keyframe / landmark: stride-7 / -3 / -12 scalar loops, 6 ms per 4096 windows)
    const int stride = gridDim.x * 256, t0 = blockIdx.x * 256 + threadIdx.x;
    const size_t k0 = d.kf0, p0 = d.pt0, o0 = d.obs0;
    for (int i = t0; i < 7 * d.n_kf; i += stride) B.pose[7 * k0 + i] = B.pose0[7 * k0 + i];
    for (int i = t0; i < 3 * d.n_kf; i += stride) B.vel[3 * k0 + i] = B.vel0[3 * k0 + i];
    for (int i = t0; i < 12 * d.n_kf; i += stride) B.bias[12 * k0 + i] = B.bias0[12 * k0 + i];
    for (int i = t0; i < 3 * d.n_pt; i += stride) B.pt[3 * p0 + i] = B.pt0[3 * p0 + i];
    for (int i = t0; i < d.n_obs; i += stride) { B.lvl[o0 + i] = 0; B.chi2_e[o0 + i] = 0.0; }
    for (int a = t0; a < d.n_kf; a += stride) {   // R|t cache from the UPLOADED pose (the working copy is being written by other threads)
        const double* T = B.pose0 + 7 * (k0 + a);
        double* C = B.kfR + 12 * (k0 + a);
        double R[9];
        q2R(T + 3, R);
#pragma unroll
        for (int i = 0; i < 9; i++) C[i] = R[i];
        C[9] = T[0]; C[10] = T[1]; C[11] = T[2];
    }
    if (t0 == 0) {
        WinCtrl& c = B.ctrl[w];
        c.stage = 0; c.it = 0; c.active = 0; c.status = 0;
        c.its_done[0] = c.its_done[1] = 0;
        c.robust_vis = (d.protocol == 1) ? (d.robust != 0) : 1;
        c.chol_fail = 0; c.aborted = 0;
        c.n_trace = 0; c.n_outliers = 0;
        c.lm_trial = 0; c.lm_need_trial = 0; c.nbad = 0; c.lin_its = 0; c.polls = 0;
        c.lambda = 0; c.ni = 2; c.chi_prev = 0; c.chi_ini = 0;
        c.chi2_vis = c.chi2_prv = c.chi2_bias = 0;
    }
}

// ------------------------------------------------------------------------------------------------
// K_stage: initializeOptimization(level) -- active set of the stage (sparse_optimizer.cpp:199-267)
//   phase 0: clear var_act, arm the control block.  phase 1: mark variables touched by an active edge.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_stage_clear(Batch B, int stage) {
    const int w = blockIdx.y;
    const WinDesc& d = B.desc[w];
    WinCtrl& c = B.ctrl[w];
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t < d.nS) B.var_act[d.vec0 + t] = 0;
    if (t == 0) {
        if (stage == 0) {
            const int stop = B.stop_word ? *B.stop_word : 0;   // the check on entry: not one of the counted polls
            if (stop) { c.aborted = 1; c.status = 2; c.active = 0; }  // src/Optimizer.cpp:453-455
            else c.active = 1;
        } else if (d.protocol == 1) {
            c.active = 0;  // BundleAdjustment: a single optimize(nIterations), no second stage (:3517, :835)
        } else {
            const int stop = (c.status == 2) ? 1 : poll_stop(B, c);
            if (c.aborted || stop) {  // :462-470 -- skip stage 2
                if (!c.aborted) { c.aborted = 1; c.status = 1; }
                c.active = 0;
            } else {
                c.active = 1;
                c.lambda = 0;  // the second optimize() starts its own lambda (computeLambdaInit at its iteration 0)
            }
            c.robust_vis = 0;  // e->setRobustKernel(0) on every vision edge, :489
        }
        if (d.its[stage] <= 0) c.active = 0;  // optimize(0) runs nothing
        c.stage = stage; c.it = 0; c.chol_fail = 0;
        c.lm_trial = 0; c.lm_need_trial = 0; c.nbad = 0; c.ni = 2;
    }
}

// One wave per (window, keyframe): the keyframe's PR block is in the index mapping if one of its edges is active -- as the
// observer (its slot records, in slot order) or, for inverse-depth landmarks, as the reference keyframe of a landmark with an
// active edge.  The wave stops at the first hit, which is almost always in its first 64 records.  (Round 1: a thread per edge,
// five dependent loads each, 1.9 ms per 4096 windows.)  Blocks behind the keyframes: the IMU edges, a thread each.
__global__ void __launch_bounds__(64) k_stage_mark(Batch B, int nblk_kf) {
    const int w = blockIdx.y;
    const WinDesc& d = B.desc[w];
    int* va = B.var_act + d.vec0;
    const int lane = threadIdx.x;
    if ((int)blockIdx.x < nblk_kf) {
        const int a = blockIdx.x;
        if (a >= d.n_free || !(kf_free(B, d, a) & 1)) return;
        const int* kseg = B.kf_seg + d.kf0 + d.win;
        bool hit = false;
        for (int c = kseg[a]; c < kseg[a + 1] && !hit; c += 64) {
            const int slot = c + lane;
            const bool on = slot < kseg[a + 1] && !B.lvl[d.obs0 + B.slot_o[d.obs0 + slot]];
            hit = __ballot(on) != 0ull;
        }
        if (!hit && d.variant == 2) {
            const int* rseg = B.ref_seg + d.kf0 + d.win;
            const int* ob = B.pt_obs_begin + d.pt0 + d.win;
            for (int c = rseg[a]; c < rseg[a + 1] && !hit; c += 64) {
                const int rec = c + lane;
                bool on = false;
                if (rec < rseg[a + 1]) {
                    const int p = B.rec_lm[d.pt0 + rec];
                    for (int o = ob[p]; o < ob[p + 1] && !on; o++) on = !B.lvl[d.obs0 + o];
                }
                hit = __ballot(on) != 0ull;
            }
        }
        if (hit && lane < 6) va[vpos(d, a, lane)] = 1;
        return;
    }
    const int t = (blockIdx.x - nblk_kf) * 64 + lane;
    if (t < d.n_imu) {
        const int i = B.imu_i[d.imu0 + t], j = B.imu_j[d.imu0 + t];
        const int fi = kf_free(B, d, i), fj = kf_free(B, d, j), act = imu_act(B, d, i, j);
        // the PRV edge touches PR_i, PR_j, V_i, V_j, Bias_i; the bias edge Bias_i, Bias_j
        for (int k = 0; k < 15; k++) {
            const int part = (k < 6) ? 0 : ((k < 9) ? 1 : 2);
            if (((fi >> part) & 1) && ((act & 1) || (part == 2 && (act & 2)))) va[vpos(d, i, k)] = 1;
            if (((fj >> part) & 1) && ((part < 2 && (act & 1)) || (part == 2 && (act & 2)))) va[vpos(d, j, k)] = 1;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Linearisation: residuals + analytic Jacobians + Huber + per-landmark products.  Replaces
// computeActiveErrors (sparse_optimizer.cpp:61-88) + the vision/IMU part of buildSystem
// (block_solver.hpp:502-560); Jacobians never reach HBM unreduced: only their products do.
// ------------------------------------------------------------------------------------------------
// EdgeNavStatePRV + EdgeNavStateBias of one keyframe pair (the fused 15-D IMU factor): error, chi2, and in
// LIN_FULL mode the 30x30 local Hessian J^T (rho' Omega) J and rhs in local order [PR_i V_i B_i | PR_j V_j B_j].
// Two steps.  lin_imu_res: the Lie-group part -- residuals, Huber weights, the 9x30 Jacobian -- is scalar work, ONE LANE per
// keyframe pair (64 pairs per wave side by side; round 1 ran it on lane 0 of a wave per pair, 63 lanes idle for ~1500
// instructions); its products go to a 104-double record per pair (imu_jrec): the six 3x3 matrices the Jacobian is made of (R_i^T,
// hat(R_i^T vP), JrInv R_j^T R_i, hat(R_i^T vV), JrInv Exp(r_phi)^T Jr JRg, JrInv: 54), the errors, weights and column masks (47),
// the Huber weight and the active flags (3).  The records of a window are stored ELEMENT-major (element q of pair k at
// q * n_imu + k): the lanes of k_lin_imu_res -- one pair each -- store to consecutive addresses.  (Round 2 first wrote the whole
// 9x30 Jacobian and the weighted 9x9 information, 400 doubles pair-major: 400 store instructions of 49 scattered 8-byte pieces each,
// 2 GB of partial-line writes per pass and 0.6 of the kernel's 0.89 ms.)  lin_imu_hess: one wave per pair rebuilds J (preintegration
// Jacobians from imu_meas) and Omega = w * info, then H = J^T (Omega J) and the rhs.
#define IMU_JREC 104
DEVI void lin_imu_res(const Batch& B, const WinDesc& d, int k, int mode) {
    const size_t gk = d.imu0 + k;
    const int i = B.imu_i[gk], j = B.imu_j[gk];
    const int act = imu_act(B, d, i, j);
    if (!act) return;  // every vertex fixed: not in the active set
    const double* meas = B.imu_meas + 61 * gk;
    double* jr = B.imu_jrec + IMU_JREC * (size_t)d.imu0 + k;   // element q of this pair: jr[q * n_imu]
    const int js = d.n_imu;
    {
        const double* Ti = B.pose + 7 * (size_t)(d.kf0 + i);
        const double* Tj = B.pose + 7 * (size_t)(d.kf0 + j);
        const double* Vi = B.vel + 3 * (size_t)(d.kf0 + i);
        const double* Vj = B.vel + 3 * (size_t)(d.kf0 + j);
        const double* bi = B.bias + 12 * (size_t)(d.kf0 + i);
        const double* bj = B.bias + 12 * (size_t)(d.kf0 + j);
        const double dT = meas[0], dT2 = dT * dT;
        const double *dP = meas + 1, *dV = meas + 4, *dRm = meas + 7;
        const double *JPg = meas + 16, *JPa = meas + 25, *JVg = meas + 34, *JVa = meas + 43, *JRg = meas + 52;
        const double *dbg = bi + 6, *dba = bi + 9;
        double Ri[9], Rj[9];
        q2R(Ti + 3, Ri);
        q2R(Tj + 3, Rj);
        // residuals (g2otypes.cpp:207-215)
        double vP[3], vV[3], rvP[3], rvV[3], c1[3], c2[3], e[9];
        for (int m = 0; m < 3; m++) {
            vP[m] = Tj[m] - Ti[m] - Vi[m] * dT - 0.5 * d.g[m] * dT2;
            vV[m] = Vj[m] - Vi[m] - d.g[m] * dT;
        }
        double qiT[4];
        so3inv(Ti + 3, qiT);
        qrot(qiT, vP, rvP);
        qrot(qiT, vV, rvV);
        mv3(JPg, dbg, c1); mv3(JPa, dba, c2);
        for (int m = 0; m < 3; m++) e[m] = rvP[m] - (dP[m] + c1[m] + c2[m]);
        mv3(JVg, dbg, c1); mv3(JVa, dba, c2);
        for (int m = 0; m < 3; m++) e[6 + m] = rvV[m] - (dV[m] + c1[m] + c2[m]);
        double wv[3], qd[4], qR[4], qA[4], qAi[4], qB[4], qC[4];
        mv3(JRg, dbg, wv);
        so3exp(wv, qd);
        R2q(dRm, qR);
        qnorm(qR);
        so3mul(qR, qd, qA);
        so3inv(qA, qAi);
        so3mul(qAi, qiT, qB);
        so3mul(qB, Tj + 3, qC);
        so3log(qC, e + 3);
        const double* info = B.imu_info + 81 * gk;
        double s = 0;
        for (int a = 0; a < 9; a++) {
            double tt = 0;
            for (int b = 0; b < 9; b++) tt += info[9 * a + b] * e[b];
            s += e[a] * tt;
        }
        double rw = 1.0, rwb = 1.0;
        const bool rk = imu_robust(d);
        const double rob = rk ? huber(s, d.hub_prv, &rw) : s;
        double eb[6];
        for (int m = 0; m < 3; m++) {
            eb[m] = (bj[m] + bj[6 + m]) - (bi[m] + bi[6 + m]);
            eb[3 + m] = (bj[3 + m] + bj[9 + m]) - (bi[3 + m] + bi[9 + m]);
        }
        const double wg = d.inv_bg / dT, wa = d.inv_ba / dT;
        const double sb = wg * (eb[0] * eb[0] + eb[1] * eb[1] + eb[2] * eb[2]) + wa * (eb[3] * eb[3] + eb[4] * eb[4] + eb[5] * eb[5]);
        const double robb = rk ? huber(sb, d.hub_bias, &rwb) : sb;
        double* ch = B.imu_chi + 4 * gk;
        ch[0] = (act & 1) ? rob : 0.0; ch[1] = (act & 2) ? robb : 0.0; ch[2] = (act & 1) ? s : 0.0; ch[3] = (act & 2) ? sb : 0.0;
        if (mode == LIN_FULL) {
            // record elements 54..100: 9 err + 6 bias err + 2 weights (bias wg, wa scaled) + 30 column masks; 101: Huber weight of the
            // PRV edge (0: edge inactive), so that Omega = w * info is formed by the reader
            for (int a = 0; a < 9; a++) jr[(54 + a) * js] = e[a];
            for (int a = 0; a < 6; a++) jr[(54 + 9 + a) * js] = eb[a];
            jr[(54 + 15) * js] = (act & 2) ? rwb * wg : 0.0; jr[(54 + 16) * js] = (act & 2) ? rwb * wa : 0.0;
            {   // column mask of the 30 local variables: 0 where the vertex is fixed through kf_fix
                const int fi = (i < d.n_free) ? kf_free(B, d, i) : 7, fj = (j < d.n_free) ? kf_free(B, d, j) : 7;
                for (int q = 0; q < 15; q++) {
                    const int part = (q < 6) ? 0 : ((q < 9) ? 1 : 2);
                    jr[(54 + 17 + q) * js] = ((fi >> part) & 1) ? 1.0 : 0.0;
                    jr[(54 + 32 + q) * js] = ((fj >> part) & 1) ? 1.0 : 0.0;
                }
            }
            jr[101 * js] = rw;
            jr[102 * js] = (act & 1) ? 1.0 : 0.0;
            // Jacobians (g2otypes.cpp:296-359); local columns: PR_i 0..5, V_i 6..8, B_i 9..14, PR_j 15..20, V_j 21..23
            double RiT[9], RjT[9], JrI[9], H1[9], H2[9], T1[9], T2[9];
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) { RiT[3 * a + b] = Ri[3 * b + a]; RjT[3 * a + b] = Rj[3 * b + a]; }
            so3jrinv(e + 3, JrI);
            double mP[3], mV[3];
            mv3(RiT, vP, mP);
            mv3(RiT, vV, mV);
            hat3(mP, H1);
            hat3(mV, H2);
            mm3(JrI, RjT, T1);
            mm3(T1, Ri, T2);  // JrInv Rj^T Ri
            double qe[4], qei[4], ExpT[9], JrB[9], T3[9];
            so3exp(e + 3, qe);
            so3inv(qe, qei);
            q2R(qei, ExpT);
            so3jr(wv, JrB);
            mm3(JrI, ExpT, T1);
            mm3(T1, JrB, T3);
            mm3(T3, JRg, T1);  // JrInv Exp(rphi)^T Jr(JRg dbg) JRg
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) {
                    const int q = 3 * a + b;
                    // the Jacobian's building blocks (assembled by lin_imu_fill_J): record elements 0..53
                    jr[(0 + q) * js] = RiT[q];
                    jr[(9 + q) * js] = H1[q];
                    jr[(18 + q) * js] = T2[q];
                    jr[(27 + q) * js] = H2[q];
                    jr[(36 + q) * js] = T1[q];
                    jr[(45 + q) * js] = JrI[q];
                }
        }
    }
}

// H = J^T (Omega J) + the bias edge's -I / +I terms, rhs = -J^T Omega e, of one keyframe pair.  ONE WAVE by contract (sm: 672 doubles):
// the phases are ordered with wave_lds_sync(), not with a workgroup barrier -- k_lin2_imu calls it from wave 0 of a 256-thread
// workgroup whose other waves have left (a barrier behind a divergent exit is undefined in the HIP model)
DEVI void lin_imu_hess(const Batch& B, const WinDesc& d, int k, double* sm) {
    const size_t gk = d.imu0 + k;
    const int i = B.imu_i[gk], j = B.imu_j[gk];
    if (!imu_act(B, d, i, j)) return;
    double* J = sm;            // 9 x 30
    double* Om = sm + 270;     // 9 x 9 weighted information
    double* T = sm + 351;      // 9 x 30 = Om J ; before that: the 54 building blocks of J
    double* er = sm + 621;     // 9 err + 6 bias err + 2 weights + 30 column masks
    const int t = threadIdx.x;
    const double* jr = B.imu_jrec + IMU_JREC * (size_t)d.imu0 + k;
    const int js = d.n_imu;
    const double* meas = B.imu_meas + 61 * gk;
    const double* info = B.imu_info + 81 * gk;
    const double rw = jr[101 * js];
    const bool on = jr[102 * js] != 0.0;
    for (int q = t; q < 270; q += 64) J[q] = 0.0;
    if (t < 54) T[t] = jr[t * js];
    if (t < 47) er[t] = jr[(54 + t) * js];
    for (int q = t; q < 81; q += 64) Om[q] = on ? rw * info[q] : 0.0;
    wave_lds_sync();
    if (t < 9) {   // Jacobians (g2otypes.cpp:296-359); local columns: PR_i 0..5, V_i 6..8, B_i 9..14, PR_j 15..20, V_j 21..23
        const int a = t / 3, b = t % 3, q = t;
        const double dT = meas[0];
        const double *JPg = meas + 16, *JPa = meas + 25, *JVg = meas + 34, *JVa = meas + 43;
        const double *RiT = T, *H1 = T + 9, *T2 = T + 18, *H2 = T + 27, *T1 = T + 36, *JrI = T + 45;
        J[(0 + a) * 30 + 0 + b] = -RiT[q];            // d rP / d P_i
        J[(0 + a) * 30 + 3 + b] = H1[q];              // d rP / d phi_i
        J[(3 + a) * 30 + 3 + b] = -T2[q];             // d rphi / d phi_i
        J[(6 + a) * 30 + 3 + b] = H2[q];              // d rV / d phi_i
        J[(0 + a) * 30 + 6 + b] = -RiT[q] * dT;       // d rP / d V_i
        J[(6 + a) * 30 + 6 + b] = -RiT[q];            // d rV / d V_i
        J[(0 + a) * 30 + 9 + b] = -JPg[q];            // d rP / d dbg_i
        J[(0 + a) * 30 + 12 + b] = -JPa[q];           // d rP / d dba_i
        J[(3 + a) * 30 + 9 + b] = -T1[q];             // d rphi / d dbg_i
        J[(6 + a) * 30 + 9 + b] = -JVg[q];
        J[(6 + a) * 30 + 12 + b] = -JVa[q];
        J[(0 + a) * 30 + 15 + b] = RiT[q];            // d rP / d P_j
        J[(3 + a) * 30 + 18 + b] = JrI[q];            // d rphi / d phi_j
        J[(6 + a) * 30 + 21 + b] = RiT[q];            // d rV / d V_j
    }
    wave_lds_sync();
    for (int q = t; q < 270; q += 64) {  // T = Om J
        const int a = q / 30, col = q % 30;
        double s = 0;
#pragma unroll
        for (int b = 0; b < 9; b++) s += Om[9 * a + b] * J[b * 30 + col];
        T[q] = s;
    }
    wave_lds_sync();
    double* H = B.imuH + VBA_IMUH * gk;
    for (int q = t; q < 900; q += 64) {  // H = J^T T  (+ bias edge: -I/+I Jacobians on B_i (9..14), B_j (24..29))
        const int r = q / 30, col = q % 30;
        double s = 0;
#pragma unroll
        for (int a = 0; a < 9; a++) s += J[a * 30 + r] * T[a * 30 + col];
        const int rb = (r >= 9 && r < 15) ? r - 9 : ((r >= 24) ? r - 24 : -1);
        const int cb = (col >= 9 && col < 15) ? col - 9 : ((col >= 24) ? col - 24 : -1);
        if (rb >= 0 && rb == cb) {
            const double wq = (rb < 3) ? er[15] : er[16];
            const double sr = (r < 15) ? -1.0 : 1.0, scn = (col < 15) ? -1.0 : 1.0;
            s += sr * scn * wq;
        }
        H[q] = s * er[17 + r] * er[17 + col];
    }
    if (t < 30) {  // rhs = -J^T Om e  (+ bias edge)
        double s = 0;
        for (int a = 0; a < 9; a++) s -= T[a * 30 + t] * er[a];
        const int rb = (t >= 9 && t < 15) ? t - 9 : ((t >= 24) ? t - 24 : -1);
        if (rb >= 0) {
            const double wq = (rb < 3) ? er[15] : er[16];
            const double sr = (t < 15) ? -1.0 : 1.0;
            s -= sr * wq * er[9 + rb];
        }
        H[900 + t] = s * er[17 + t];
    }
}

// ------------------------------------------------------------------------------------------------
// K_lin2 (inverse-depth variant), edge-parallel.  A 256-thread workgroup owns a
// run of consecutive landmarks with <= 256 edges (ranges built at upload): one lane per EDGE evaluates the
// residual and Jacobians (coalesced observation loads), the per-landmark sums (D, b_l, W0, g0, G0) are taken by
// one lane per LANDMARK from LDS, and the 256-B edge records leave through an LDS transpose as contiguous
// 16-B chunks instead of 64 scattered records per store instruction.
// ------------------------------------------------------------------------------------------------
#define LIN2_ES 15   // LDS row stride (doubles) of one edge: phase B/C [0..5] A [6..7] a [8..9] r ; phase D/E [0..11] Bi [12..13] r
#define LIN2_PS 19   // LDS row stride of one landmark: dd, y(3), Xw(3), N0(9), ref_free, sD, beta
#define LIN2_LDS ((256 * LIN2_ES + 64 * LIN2_PS + 4) * 8)   // 40 480 B: four workgroups per CU

// The IMU factors: a launch of their own behind the vision linearisation (same stream) -- inlined into k_lin2 their ~60 live
// doubles of Lie algebra set the register budget of the 130x more numerous edge workgroups.  k_lin_imu_res: a lane per keyframe
// pair; k_lin_imu_hess (LIN_FULL only): a wave per pair.  `gate_lm`: the XYZ / Levenberg-Marquardt gating of k_lin_xyz.
DEVI bool lin_imu_gate(const WinDesc& d, const WinCtrl& c, int mode) {
    if (!c.active) return false;
    if (d.variant == 2) return true;
    if (mode == 2 /* LIN_ERR_TRIAL */ && !win_on(d, c)) return false;
    if (mode == LIN_FULL && d.algo == 1 && c.lm_need_trial) return false;
    return true;
}
__global__ void __launch_bounds__(64) k_lin_imu_res(Batch B, int mode) {
    const int w = blockIdx.y;
    const WinDesc& d = B.desc[w];
    if (!lin_imu_gate(d, B.ctrl[w], mode)) return;
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= d.n_imu) return;
    lin_imu_res(B, d, k, (mode == LIN_FULL) ? LIN_FULL : LIN_ERR);
}
// few windows: both steps in one launch, a wave per pair (lane 0 runs the Lie-group part), one launch gap less per pass
__global__ void __launch_bounds__(64) k_lin_imu_pair(Batch B, int mode) {
    __shared__ double sm[672];
    const int w = blockIdx.y;
    const WinDesc& d = B.desc[w];
    if (!lin_imu_gate(d, B.ctrl[w], mode)) return;
    const int k = blockIdx.x;
    if (k >= d.n_imu) return;
    if (threadIdx.x == 0) lin_imu_res(B, d, k, (mode == LIN_FULL) ? LIN_FULL : LIN_ERR);
    if (mode != LIN_FULL) return;
    __threadfence_block();
    __syncthreads();
    lin_imu_hess(B, d, k, sm);
}
__global__ void __launch_bounds__(64) k_lin_imu_hess(Batch B) {
    __shared__ double sm[672];
    const int w = blockIdx.y;
    const WinDesc& d = B.desc[w];
    if (!lin_imu_gate(d, B.ctrl[w], LIN_FULL)) return;
    const int k = blockIdx.x;
    if (k >= d.n_imu) return;
    lin_imu_hess(B, d, k, sm);
}

// The geometry of one inverse-depth edge (g2otypes.cpp:25-60), shared by the linearisation and by the two passes that classify
// edges (k_classify, k_final_edges): those evaluate chi2 and depth again at the estimates instead of reading a per-edge copy the
// linearisation used to leave behind on EVERY pass (16 of its 218 bytes per edge; the estimates do not move between a window's last
// linearisation and its classification, and the expressions are these same ones, so the values are the same).
DEVI void idp_point_world(const WinDesc& d, const double* C0, double rho, double xb, double yb, double& dd, double (&c0)[3], double (&b0)[3],
                          double (&Xw)[3]) {   // C0: R|t of the reference keyframe
    if (rho < 1e-6) rho = 1e-6;  // g2otypes.cpp:42-47
    dd = 1.0 / rho;
    const double P0[3] = {xb * dd, yb * dd, dd};
    double tb[3];
    mtv3(d.Rcb, P0, c0);
    mtv3(d.Rcb, d.tcb, tb);
    b0[0] = c0[0] - tb[0]; b0[1] = c0[1] - tb[1]; b0[2] = c0[2] - tb[2];
    mv3(C0, b0, Xw);
    Xw[0] += C0[9]; Xw[1] += C0[10]; Xw[2] += C0[11];
}
DEVI void idp_edge_pc(const WinDesc& d, const double* Xw, const double* Ri, const double* ti, double (&ta)[3], double (&Pc)[3]) {
    const double v[3] = {Xw[0] - ti[0], Xw[1] - ti[1], Xw[2] - ti[2]};
    mtv3(Ri, v, ta);
    mv3(d.Rcb, ta, Pc);
    Pc[0] += d.tcb[0]; Pc[1] += d.tcb[1]; Pc[2] += d.tcb[2];
}
DEVI double idp_edge_chi2(const WinDesc& d, const double (&Pc)[3], double u, double v, double wgt, double& iz, double& ex, double& ey) {
    iz = 1.0 / Pc[2];
    ex = u - (Pc[0] * iz * d.K[0] + d.K[2]);
    ey = v - (Pc[1] * iz * d.K[1] + d.K[3]);
    return ex * (wgt * ex) + ey * (wgt * ey);
}
// chi2 and depth of edge go at the current estimates (the classification passes)
DEVI void idp_edge_eval(const Batch& B, const WinDesc& d, size_t go, double& s, double& depth) {
    const size_t gp = d.pt0 + B.obs_pt[go];
    const double* C0 = B.kfR + 12 * (size_t)(d.kf0 + B.pt_ref[gp]);
    const double* Ci = B.kfR + 12 * (size_t)(d.kf0 + B.obs_kf[go]);
    double dd, c0[3], b0[3], Xw[3], ta[3], Pc[3], iz, ex, ey;
    idp_point_world(d, C0, B.pt[3 * gp], B.pt[3 * gp + 1], B.pt[3 * gp + 2], dd, c0, b0, Xw);
    idp_edge_pc(d, Xw, Ci, Ci + 9, ta, Pc);
    depth = Pc[2];
    s = idp_edge_chi2(d, Pc, B.obs_uv[2 * go], B.obs_uv[2 * go + 1], B.obs_w[go], iz, ex, ey);
}

DEVI void lin2_body(const Batch& B, int nblk_lin, int mode, double* lsm) {
    double* ER = lsm;                       // 256 x LIN2_ES
    double* PT = lsm + 256 * LIN2_ES;       // 64 x LIN2_PS
    double* red = PT + 64 * LIN2_PS;        // 4
    const int w = blockIdx.y;
    const WinDesc& d = B.desc[w];
    const WinCtrl& c = B.ctrl[w];
    if (!c.active) return;
    const int t = threadIdx.x;
    const int lb = blockIdx.x;
    if (lb >= d.n_part_lin) return;
    const int4 run = reinterpret_cast<const int4*>(B.lin_blk)[d.lb0 + lb];  // one load instead of the chain table -> CSR
    const int p0 = run.x, p1 = run.y, e0 = run.z, e1 = run.w;
    const int* ob = B.pt_obs_begin + d.pt0 + d.win;
    const int ne = e1 - e0, npb = p1 - p0;
    const double fx = d.K[0], fy = d.K[1], cx = d.K[2], cy = d.K[3];
    // the edge lanes fetch their own inputs BEFORE phase A: the chain obs_kf -> keyframe R|t then overlaps with phase A's
    // pt_ref -> reference R|t instead of following it behind the barrier
    int pl = 0, e_kf = 0, e_pe = 0;
    bool e_out = true;
    double Ri[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, e_t[3] = {0, 0, 0}, e_u = 0, e_v = 0, e_w = 0;
    if (t < ne) {
        const size_t go = d.obs0 + e0 + t;
        pl = B.obs_pt[go] - p0;
        e_kf = B.obs_kf[go];
        e_pe = B.slot_perm[go];
        e_out = B.lvl[go] != 0;
        e_u = B.obs_uv[2 * go]; e_v = B.obs_uv[2 * go + 1];
        e_w = B.obs_w[go];
        const double* Ci = B.kfR + 12 * (size_t)(d.kf0 + e_kf);
#pragma unroll
        for (int i = 0; i < 9; i++) Ri[i] = Ci[i];
        e_t[0] = Ci[9]; e_t[1] = Ci[10]; e_t[2] = Ci[11];
    }
    // likewise the phase-C lanes (four per landmark): CSR bounds and record position of their landmark
    int c_ob = 0, c_oe = 0, c_pp = 0;
    if (t < 4 * npb) {
        const int p = p0 + (t >> 2);
        c_ob = ob[p] - e0;
        c_oe = ob[p + 1] - e0;
        c_pp = B.pt_perm[d.pt0 + p];
    }
    // A. one lane per landmark: quantities shared by all its edges
    int a_run = 0, a_nruns = 0;   // (landmark lanes, all in wave 0: run of consecutive landmarks with one reference keyframe; runs of the workgroup)
    bool a_first = false;
    if (t < npb) {
        const size_t gp = d.pt0 + p0 + t;
        const int rf = B.pt_ref[gp];
        const double* C0 = B.kfR + 12 * (size_t)(d.kf0 + rf);
        double R0[9], c0[3], b0[3], y[3], Xw[3], Hb[9], N0[9], dd;
#pragma unroll
        for (int i = 0; i < 9; i++) R0[i] = C0[i];
        idp_point_world(d, C0, B.pt[3 * gp], B.pt[3 * gp + 1], B.pt[3 * gp + 2], dd, c0, b0, Xw);
        mv3(R0, c0, y);
        hat3(b0, Hb);
        mm3(R0, Hb, N0);
        double* q = PT + t * LIN2_PS;
        q[0] = dd;
#pragma unroll
        for (int i = 0; i < 3; i++) { q[1 + i] = y[i]; q[4 + i] = Xw[i]; }
#pragma unroll
        for (int i = 0; i < 9; i++) q[7 + i] = N0[i];
        q[16] = (kf_free(B, d, rf) & 1) ? 1.0 : 0.0;
        if (mode == LIN_FULL) {
            const int rf_prev = __shfl_up(rf, 1, 64);
            a_first = (t == 0) || (rf != rf_prev);
            const unsigned long long fm = __ballot(a_first);
            a_run = __popcll(fm & ((2ull << t) - 1ull)) - 1;
            a_nruns = __popcll(fm);
        }
    }
    __syncthreads();
    // B. one lane per edge.  With A = sqrt(rho' w) J_pi R_cb R_i^T (2x3) the two pose Jacobians of the edge are
    //    Bi = [A | -sqrt(.) (J_pi R_cb) x ta] (observer) and Br = A [-I | N0] (reference keyframe, N0 per landmark).
    double chi = 0.0;
    double ub[6] = {0, 0, 0, 0, 0, 0};      // Bi^T a: numerator of the edge's slot record
    double rpc[3] = {0, 0, 1}, rsc = 0.0;   // edge record: P_c and the Jacobian scale (0: no Jacobian for the readers)
    double a[2] = {0, 0}, r2[2] = {0, 0};
    if (t < ne) {
        const size_t go = d.obs0 + e0 + t;
        const double* q = PT + pl * LIN2_PS;
        const int kf = e_kf;
        double ta[3], Pc[3];
        idp_edge_pc(d, q + 4, Ri, e_t, ta, Pc);
        rpc[0] = Pc[0]; rpc[1] = Pc[1]; rpc[2] = Pc[2];
        double A[6] = {0, 0, 0, 0, 0, 0};
        if (!e_out) {
            double iz, ex, ey;
            const double wgt = e_w;
            const double s = idp_edge_chi2(d, Pc, e_u, e_v, wgt, iz, ex, ey);
            double rw = 1.0;
            if (c.robust_vis) chi = huber(s, d.hub_vis, &rw);
            else chi = s;
            if (mode == LIN_FULL) {
                const double sc = sqrt(rw * wgt);
                const double Jp[6] = {fx * iz, 0.0, -Pc[0] * iz * fx * iz, 0.0, fy * iz, -Pc[1] * iz * fy * iz};
                double Jc[6];
#pragma unroll
                for (int rr = 0; rr < 2; rr++)
#pragma unroll
                    for (int k = 0; k < 3; k++)
                        Jc[3 * rr + k] = Jp[3 * rr] * d.Rcb[k] + Jp[3 * rr + 1] * d.Rcb[3 + k] + Jp[3 * rr + 2] * d.Rcb[6 + k];
                const bool of = kf_free(B, d, kf) & 1;
                rsc = of ? sc : 0.0;
                double Bi[12];
#pragma unroll
                for (int rr = 0; rr < 2; rr++) {
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        A[3 * rr + k] = sc * (Jc[3 * rr] * Ri[3 * k] + Jc[3 * rr + 1] * Ri[3 * k + 1] + Jc[3 * rr + 2] * Ri[3 * k + 2]);
                        Bi[6 * rr + k] = of ? A[3 * rr + k] : 0.0;
                    }
                    a[rr] = q[0] * (A[3 * rr] * q[1] + A[3 * rr + 1] * q[2] + A[3 * rr + 2] * q[3]);
                    const double j0 = Jc[3 * rr], j1 = Jc[3 * rr + 1], j2 = Jc[3 * rr + 2];
                    Bi[6 * rr + 3] = of ? -sc * (j1 * ta[2] - j2 * ta[1]) : 0.0;
                    Bi[6 * rr + 4] = of ? -sc * (j2 * ta[0] - j0 * ta[2]) : 0.0;
                    Bi[6 * rr + 5] = of ? -sc * (j0 * ta[1] - j1 * ta[0]) : 0.0;
                }
#pragma unroll
                for (int i = 0; i < 6; i++) ub[i] = Bi[i] * a[0] + Bi[6 + i] * a[1];
                r2[0] = sc * ex; r2[1] = sc * ey;
            }
        }
        if (mode == LIN_FULL) {
            double* er = ER + t * LIN2_ES;
#pragma unroll
            for (int i = 0; i < 6; i++) er[i] = A[i];
            er[6] = a[0]; er[7] = a[1]; er[8] = r2[0]; er[9] = r2[1];
        }
    }
    if (mode != LIN_FULL) {
        const double tot = block_sum256(chi, red);
        if (t == 0) B.part[d.part0 + lb] = tot;
        return;
    }
    __syncthreads();
    // C. FOUR lanes per landmark: each sums every fourth edge (fixed order), the quad adds up (fixed order again), then
    //    all four hold D, b_l and, for the reference keyframe, sum Br^T Br = Q^T (sum A^T A) Q, sum Br^T a = Q^T sum A^T a,
    //    sum Br^T r = Q^T sum A^T r with Q = [-I | N0].  The landmark's reference slot record leaves from lane 0 of the quad; the
    //    reference keyframe's direct terms (G0: 21, g0: 6) go to LDS -- over the edge rows, which nobody reads any more after the
    //    barrier -- and leave summed over RUNS of consecutive landmarks with one reference keyframe (phase E): one 256-byte record
    //    per run instead of one per landmark (a quarter of the bytes of a pass when the caller's landmarks come track by track).
    double* GS = ER;                                            // 64 x 28
    int* RS = reinterpret_cast<int*>(ER + 64 * 28);             // run starts [nruns + 1], then the number of runs
    double gv[7];
    const int c_jl = t >> 2, c_sub = t & 3;
    if (t < 4 * npb) {
        const int jl = c_jl, sub = c_sub;
        double D = 0, bl = 0, M[6] = {0, 0, 0, 0, 0, 0}, wa[3] = {0, 0, 0}, wr[3] = {0, 0, 0};
        for (int o = c_ob + sub; o < c_oe; o += 4) {
            const double* er = ER + o * LIN2_ES;
            const double a0 = er[6], a1 = er[7], q0 = er[8], q1 = er[9];
            D += a0 * a0 + a1 * a1;
            bl -= a0 * q0 + a1 * q1;
            int gi = 0;
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const double b0 = er[i], b1 = er[3 + i];
                wa[i] += b0 * a0 + b1 * a1;
                wr[i] += b0 * q0 + b1 * q1;
#pragma unroll
                for (int j = i; j < 3; j++) M[gi++] += b0 * er[j] + b1 * er[3 + j];
            }
        }
#define QUADSUM(v) { v = quad_sum(v); }
        QUADSUM(D) QUADSUM(bl)
#pragma unroll
        for (int i = 0; i < 6; i++) QUADSUM(M[i])
#pragma unroll
        for (int i = 0; i < 3; i++) { QUADSUM(wa[i]) QUADSUM(wr[i]) }
#undef QUADSUM
        const double sD = (D > 0.0) ? sqrt(1.0 / D) : 0.0;
        const double beta = sD * bl;
        double* q = PT + jl * LIN2_PS;
        const double rfm = q[16];  // 1: reference keyframe is free, 0: fixed (no Hessian block)
        double N0[9];
#pragma unroll
        for (int i = 0; i < 9; i++) N0[i] = q[7 + i];
        const int pp = c_pp;  // landmark records grouped by reference keyframe
        if (sub == 0) {
            q[17] = sD;
            q[18] = beta;
            double* sr = B.slot + VBA_SLOT * (size_t)(d.obs0 + d.pt0 + d.n_obs + pp);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                sr[k] = -wa[k] * rfm * sD;
                sr[3 + k] = (N0[k] * wa[0] + N0[3 + k] * wa[1] + N0[6 + k] * wa[2]) * rfm * sD;
            }
            sr[6] = beta;
            sr[7] = sD;
        }
        // (N0 = R0 hat(b0) had a 128-byte record of its own, 10 % of the bytes of a pass: its readers -- the reference items of the
        //  off-diagonal Schur pairs -- rebuild it from the landmark and the reference keyframe's rotation)
        {   // G0 (21) and g0 (6): every lane of the quad forms them, lane `sub` keeps values 7 sub .. 7 sub + 6 across the barrier
            double Ms[9], MN[9], G[28];
            Ms[0] = M[0]; Ms[1] = M[1]; Ms[2] = M[2]; Ms[3] = M[1]; Ms[4] = M[3]; Ms[5] = M[4]; Ms[6] = M[2]; Ms[7] = M[4]; Ms[8] = M[5];
            mm3(Ms, N0, MN);
            int gi = 0;
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int j = i; j < 6; j++) {
                    double v;
                    if (i < 3 && j < 3) v = Ms[3 * i + j];
                    else if (i < 3) v = -MN[3 * i + (j - 3)];
                    else v = N0[i - 3] * MN[j - 3] + N0[3 + i - 3] * MN[3 + j - 3] + N0[6 + i - 3] * MN[6 + j - 3];
                    G[gi++] = v * rfm;
                }
#pragma unroll
            for (int k = 0; k < 3; k++) {
                G[21 + k] = wr[k] * rfm;
                G[24 + k] = -(N0[k] * wr[0] + N0[3 + k] * wr[1] + N0[6 + k] * wr[2]) * rfm;
            }
            G[27] = 0.0;
#pragma unroll
            for (int i = 0; i < 7; i++) gv[i] = (sub == 0) ? G[i] : (sub == 1) ? G[7 + i] : (sub == 2) ? G[14 + i] : G[21 + i];
        }
    }
    __syncthreads();   // every quad has read its edge rows: they become GS / RS
    if (t < 4 * npb) {
        double* g = GS + c_jl * 28 + 7 * c_sub;   // [0..20] G0, [21..26] g0
#pragma unroll
        for (int i = 0; i < 7; i++) g[i] = gv[i];
    }
    if (t < npb) {
        if (a_first) RS[a_run] = t;
        if (t == 0) { RS[a_nruns] = npb; RS[65] = a_nruns; }
    }
    __syncthreads();
    // D. one lane per edge: its slot record U = Bi^T a / sqrt(D) and its 64-B edge record (P_c, scale, weighted residual),
    //    from which the Schur kernels rebuild Bi, g = -Bi^T r and Br = [-A | A N0].  Records live keyframe-major (slot_perm).
    if (t < ne) {
        const double sD = PT[pl * LIN2_PS + 17], beta = PT[pl * LIN2_PS + 18];
        const int pe = e_pe;
        double* sl = B.slot + VBA_SLOT * (size_t)(d.obs0 + d.pt0 + pe);
#pragma unroll
        for (int i = 0; i < 6; i++) sl[i] = ub[i] * sD;
        sl[6] = beta;
        sl[7] = sD;
        double2* dst = reinterpret_cast<double2*>(B.erec + VBA_EREC1 * (size_t)(d.obs0 + pe));
        dst[0] = make_double2(rpc[0], rpc[1]);
        dst[1] = make_double2(rpc[2], rsc);
        dst[2] = make_double2(r2[0], r2[1]);
    }
    // E. the reference-keyframe terms, one record per run: value v of run r is the sum over the run's landmarks, in order
    {
        const int nruns = RS[65];
        double* pr0 = B.prec + VBA_PREC * (size_t)(d.pt0 + B.prun0[d.lb0 + lb]);
        for (int idx = t; idx < nruns * 27; idx += 256) {
            const int r = idx / 27, v = idx - 27 * r;
            double sum = 0.0;
            for (int jl = RS[r]; jl < RS[r + 1]; jl++) sum += GS[jl * 28 + v];
            pr0[VBA_PREC * r + v] = sum;
        }
    }
    const double tot = block_sum256(chi, red);
    if (t == 0) B.part[d.part0 + lb] = tot;
}
__global__ void __launch_bounds__(256, 4) k_lin2(Batch B, int nblk_lin, int mode) {
    extern __shared__ double lsm[];
    lin2_body(B, nblk_lin, mode, lsm);
}
// Few windows: the IMU factors of the window in the SAME launch (the workgroups behind the edge workgroups, one keyframe pair
// each, wave 0) -- one launch and its latency less per linearisation pass; no occupancy to protect here, so no register cap
__global__ void __launch_bounds__(256) k_lin2_imu(Batch B, int nblk_lin, int mode) {
    extern __shared__ double lsm[];
    if ((int)blockIdx.x < nblk_lin) {
        lin2_body(B, nblk_lin, mode, lsm);
        return;
    }
    if (threadIdx.x >= 64) return;
    const int w = blockIdx.y;
    const WinDesc& d = B.desc[w];
    if (!lin_imu_gate(d, B.ctrl[w], mode)) return;
    const int k = blockIdx.x - nblk_lin;
    if (k >= d.n_imu) return;
    if (threadIdx.x == 0) lin_imu_res(B, d, k, (mode == LIN_FULL) ? LIN_FULL : LIN_ERR);
    if (mode != LIN_FULL) return;
    __threadfence_block();              // lane 0's record (global memory) before the other lanes of this wave read it
    __builtin_amdgcn_wave_barrier();
    lin_imu_hess(B, d, k, lsm);
}

// ------------------------------------------------------------------------------------------------
// K_ctrl: one workgroup per window.  Sums the robust chi2 in a fixed order and runs the outer-loop
// logic of OptimizationAlgorithmGaussNewton::solve (gauss_newton.cpp:50-105) as driven by
// SparseOptimizer::optimize (sparse_optimizer.cpp:354-419).  final != 0: evaluation after the last iteration.
// ------------------------------------------------------------------------------------------------
DEVI double window_chi2(const Batch& B, const WinDesc& d, double* sm) {
    const int t = threadIdx.x;
    double s = 0;
    for (int k = t; k < d.n_imu; k += 64) {
        const int i = B.imu_i[d.imu0 + k], j = B.imu_j[d.imu0 + k];
        if (i < d.n_free || j < d.n_free) s += B.imu_chi[4 * (size_t)(d.imu0 + k)] + B.imu_chi[4 * (size_t)(d.imu0 + k) + 1];
    }
    for (int k = t; k < d.n_part_lin; k += 64) s += B.part[d.part0 + k];
    return block_sum<64>(s, sm);
}

// host_slot >= 0 (a single window, no profile): the launch reports through the pinned word alive_cnt[host_slot] itself -- 1: done, the
// window has stopped; 2: done, it goes on -- and the host reads that word instead of waiting for an event behind this kernel (an
// event record between two kernels of a stream costs the next one ~4 us).
__global__ void __launch_bounds__(64) k_ctrl_gn(Batch B, int final_eval, int host_slot) {
    __shared__ double sm[64];
    const int w = blockIdx.x;
    const WinDesc& d = B.desc[w];
    WinCtrl& c = B.ctrl[w];
    if (!c.active) {
        if (host_slot >= 0 && threadIdx.x == 0) __hip_atomic_store(B.alive_cnt + host_slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    const double cur = window_chi2(B, d, sm);
    if (threadIdx.x != 0) return;
    const int st = c.stage, it = c.it;
    int active = 1;
    if (it >= 1) {
        c.its_done[st] = it;  // iterations 0..it-1 have run (++cjIterations)
        // solver Fail (zero / non-finite pivot, linear_solver_eigen.h:105-111): the step was dropped (k_update), this optimize() ends
        // (Terminate or Fail both leave the loop of sparse_optimizer.cpp:376), the window reports VBA_SOLVER_FAILED
        if (c.chol_fail) { c.status = -2; active = 0; }
        if (fabs(c.chi_prev - cur) < 1e-3) active = 0;  // Terminate, gauss_newton.cpp:97
    }
    if (final_eval || it >= d.its[st]) active = 0;
    if (active && poll_stop(B, c)) {  // terminate() before each iteration, sparse_optimizer.cpp:376
        c.aborted = 1;
        if (st == 0) c.status = 1;
        active = 0;
    }
    // the trace holds preChi2 of iteration 0 and afterChi2 of every iteration run: an optimize() stopped before its first
    // iteration has evaluated nothing
    if ((it >= 1 || active) && c.n_trace < VBA_TRACE) c.trace[c.n_trace++] = cur;
    c.chi_prev = cur;
    c.active = active;
    c.chol_fail = 0;
    c.it = it + 1;
    if (host_slot >= 0) {
        __hip_atomic_store(B.alive_cnt + host_slot, active ? 2 : 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    if (active && B.alive_cnt && it < 32 && atomicExch(B.alive_dev + st * 32 + it, 1) == 0)  // lets the host skip dead iterations
        __hip_atomic_store(B.alive_cnt + st * 32 + it, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // a posted store, not a PCIe atomic
}

// ------------------------------------------------------------------------------------------------
// K_schur: one workgroup per keyframe block pair (a <= b).  Gathers, in a fixed order, every landmark
// observed from both keyframes: S_ab = H_ab - sum_l W_al Dinv_l W_bl^T (block_solver.hpp:373-430), adds the
// IMU blocks, writes the pdim x pdim block (and its mirror) exactly once.  Diagonal pairs also produce the
// reduced right-hand side (:436-439), the unreduced b_p and the H_pp diagonal (for LM's lambda init).
// ------------------------------------------------------------------------------------------------
DEVI double wave_sum(double v) {   // (the 64-lane sums of the diagonal pass stay on the crossbar: with the DPP form k_schur_all needs two registers more and spills)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// window / pair of a workgroup.  With >= 8 windows the mapping is XCD-aware: the dispatcher deals consecutive
// workgroups round-robin over the 8 XCDs, so all workgroups of one window share (id % 8) and the window's slot
// records stay in one XCD's L2.  (Speed only; results do not depend on placement.)
DEVI bool schur_map(const Batch& B, int per_win, int& w, int& idx) {
    const int lid = blockIdx.x;
    if (B.n_win >= 8) {
        w = (lid & 7) + 8 * ((lid >> 3) / per_win);
        idx = (lid >> 3) % per_win;
    } else {
        w = lid / per_win;
        idx = lid % per_win;
    }
    return w < B.n_win;
}

// adds the IMU blocks, applies the active-set / damping rules and writes one pdim x pdim block of S (lower
// triangle only: the factorisation never reads above the diagonal)
DEVI void schur_write_block(const Batch& B, const WinDesc& d, const WinCtrl& c, int w, int pr, int a, int b,
                            const double* blk, int t = threadIdx.x, int nt = 64, int bstride = -1) {
    const int P = d.pdim, n = d.nS;
    if (bstride < 0) bstride = P;
    const bool diag = (a == b);
    const double lambda = (d.algo == 1) ? c.lambda : 0.0;
    const int qb = B.pimu_begin[d.pair0 + d.win + pr], qe = B.pimu_begin[d.pair0 + d.win + pr + 1];
    const int* va = B.var_act + d.vec0;
    double* S = B.S + d.S0;
    // sub-blocks that no tile of the factor ever reads (structurally zero in L under the V/Bias-first order) are
    // not written at all: bit0 PRxPR, bit1 PRxVB, bit2 VBxPR, bit3 VBxVB (mask built with the tile lists at upload)
    const int mask = B.pair_mask[d.pair0 + pr];
    if (mask == 0) return;   // nothing of this block is ever read (or it keeps the zero of the upload)
    if (mask == 1 && !diag && qb == qe) {  // only the 6x6 PR x PR sub-block lands in a tile the factor reads, no IMU term
        for (int q = t; q < 36; q += nt) {
            const int r = q / 6, col = q % 6;
            const int gr = vpos(d, a, r), gc = vpos(d, b, col);
            const double s = blk[r * bstride + col];
            if (gr >= gc) S[(size_t)gr * n + gc] = s;
            else S[(size_t)gc * n + gr] = s;
        }
        return;
    }
    for (int q = t; q < P * P; q += nt) {
        const int r = q / P, col = q % P;
        if (!((mask >> ((r >= 6 ? 2 : 0) + (col >= 6 ? 1 : 0))) & 1)) continue;
        double s = (r < 6 && col < 6) ? blk[r * bstride + col] : 0.0;
        for (int m = qb; m < qe; m++) {
            const int k = B.pimu[2 * (size_t)(d.pimu0 + m)], role = B.pimu[2 * (size_t)(d.pimu0 + m) + 1];
            const double* H = B.imuH + VBA_IMUH * (size_t)(d.imu0 + k);
            const int ro = (role & 1) ? 15 : 0, co = (role & 2) ? 15 : 0;  // bit0: a is j ; bit1: b is j
            s += H[(ro + r) * 30 + co + col];
        }
        const int gr = vpos(d, a, r), gc = vpos(d, b, col);
        if (diag) {
            if (!va[gr] || !va[gc]) s = (r == col) ? 1.0 : 0.0;  // vertex outside the index mapping
            else if (r == col) s += lambda;                       // setLambda, block_solver.hpp:564-589
        }
        if (gr >= gc) S[(size_t)gr * n + gc] = s;
        else if (!diag) S[(size_t)gc * n + gr] = s;
    }
}

// off-diagonal keyframe pairs (a < b).  A pair has ~90 items on average, far too few for a whole wave: four
// pairs share one wave, 16 lanes each (the 36-value reduction then costs 4 butterfly steps per FOUR pairs
// instead of 6 per pair, and every lane sees ~6 items instead of ~1.4).
// LD = landmark dimension (1: inverse depth, slot record 8 doubles; 3: XYZ, slot record 24 doubles)
// LP = lanes per pair: 64 (one pair per wave: small batches, latency) or 16 (four pairs per wave: throughput)
// Bi = [A | B_rot] of an EdgePRIDP from its 64-B record (P_c, s = sqrt(rho' w), r) and the observer's rotation R_a:
// J_c = J_pi(P_c) R_cb, A = s J_c R_a^T, B_rot = -s J_c x t_a with t_a = R_cb^T (P_c - t_cb)   (k_lin2 phase B, g2otypes.cpp:139-145)
DEVI void rebuild_edge(const WinDesc& d, const double* Ra, const double* rec, double* b0, double* b1, double& r0, double& r1) {
    const double Pc[3] = {rec[0], rec[1], rec[2]}, sc = rec[3];
    r0 = rec[4]; r1 = rec[5];
    const double fx = d.K[0], fy = d.K[1];
    const double iz = 1.0 / Pc[2];
    const double Jp[6] = {fx * iz, 0.0, -Pc[0] * iz * fx * iz, 0.0, fy * iz, -Pc[1] * iz * fy * iz};
    double Jc[6];
#pragma unroll
    for (int rr = 0; rr < 2; rr++)
#pragma unroll
        for (int k = 0; k < 3; k++)
            Jc[3 * rr + k] = Jp[3 * rr] * d.Rcb[k] + Jp[3 * rr + 1] * d.Rcb[3 + k] + Jp[3 * rr + 2] * d.Rcb[6 + k];
    const double pm[3] = {Pc[0] - d.tcb[0], Pc[1] - d.tcb[1], Pc[2] - d.tcb[2]};
    double ta[3];
    mtv3(d.Rcb, pm, ta);
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
        double* bb = rr ? b1 : b0;
#pragma unroll
        for (int k = 0; k < 3; k++)
            bb[k] = sc * (Jc[3 * rr] * Ra[3 * k] + Jc[3 * rr + 1] * Ra[3 * k + 1] + Jc[3 * rr + 2] * Ra[3 * k + 2]);
        const double j0 = Jc[3 * rr], j1 = Jc[3 * rr + 1], j2 = Jc[3 * rr + 2];
        bb[3] = -sc * (j1 * ta[2] - j2 * ta[1]);
        bb[4] = -sc * (j2 * ta[0] - j0 * ta[2]);
        bb[5] = -sc * (j0 * ta[1] - j1 * ta[0]);
    }
}

#ifndef VBA_SCHUR_UNR
#define VBA_SCHUR_UNR 1
#endif
template <int LD, int LP>
DEVI void schur_off_body(const Batch& B, int max_quads, double* blk4, int w_in = -1, int quad_in = 0) {
    constexpr int SS = (LD == 1) ? VBA_SLOT : VBA_SLOT3;
    int w = w_in, quad = quad_in;
    if (w_in < 0 && !schur_map(B, max_quads, w, quad)) return;
    const WinDesc& d = B.desc[w];
    const WinCtrl& c = B.ctrl[w];
    if (!win_on(d, c)) return;
    constexpr int NG = 64 / LP;  // pairs per wave
    const int t = threadIdx.x, g = t / LP, l16 = t % LP;
    // off-diagonal pairs only: enumerate them through the host-built list of their pair indices
    const int n_off = d.n_pairs - d.n_free;
    const int oi = quad * NG + g;
    const bool have = oi < n_off;
    int pr = 0, a = 0, b = 0, ib = 0, im = 0, ie = 0;
    if (have) {
        pr = B.off_pair[d.pair0 + oi];
        a = B.pair_a[d.pair0 + pr];
        b = B.pair_b[d.pair0 + pr];
        ib = B.item_begin[d.pair0 + d.win + pr];
        ie = B.item_begin[d.pair0 + d.win + pr + 1];
        im = (LD == 1) ? B.item_mid[d.pair0 + d.win + pr] : ie;
    }
    double acc[36];
#pragma unroll
    for (int i = 0; i < 36; i++) acc[i] = 0;
    const double* slots = B.slot + SS * (size_t)(d.obs0 + d.pt0);
    const int2* items = reinterpret_cast<const int2*>(B.items) + d.item0;
#if VBA_SCHUR_UNR
    if constexpr (LD == 1) {
        // two items per lane and trip, their four records requested together: the gather is bound by the latency of its dependent
        // fetches (occupancy experiment: 3 / 2 / 1 waves per SIMD = 107 / 128 / 179 ms), so a wave keeps twice as many in flight.
        // Every lane still takes its items in the same order: the same sums.
        int2 n0 = make_int2(0, 0), n1 = make_int2(0, 0);
        if (ib + l16 < ie) n0 = items[ib + l16];
        if (ib + l16 + LP < ie) n1 = items[ib + l16 + LP];
        for (int it = ib + l16; it < ie; it += 2 * LP) {
            const int2 i0 = n0;
            const bool two = it + LP < ie;
            const int2 i1 = two ? n1 : n0;
            if (it + 2 * LP < ie) n0 = items[it + 2 * LP];
            if (it + 3 * LP < ie) n1 = items[it + 3 * LP];
            const double *qa0 = slots + SS * (size_t)i0.x, *qb0 = slots + SS * (size_t)i0.y;
            const double *qa1 = slots + SS * (size_t)i1.x, *qb1 = slots + SS * (size_t)i1.y;
            double UA0[6], UB0[6], UA1[6], UB1[6];
#pragma unroll
            for (int i = 0; i < 6; i++) { UA0[i] = qa0[i]; UB0[i] = qb0[i]; UA1[i] = qa1[i]; UB1[i] = qb1[i]; }
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int j = 0; j < 6; j++) acc[6 * i + j] -= UA0[i] * UB0[j];
            if (two) {
#pragma unroll
                for (int i = 0; i < 6; i++)
#pragma unroll
                    for (int j = 0; j < 6; j++) acc[6 * i + j] -= UA1[i] * UB1[j];
            }
        }
    } else
#endif
    {
    int2 nxt = make_int2(0, 0);
    int nxt_lm = 0;
    if (ib + l16 < ie) {
        nxt = items[ib + l16];
        if (LD == 3) nxt_lm = B.slot_lm[d.obs0 + nxt.x];
    }
    // every item: the Schur term of one landmark seen from both keyframes -- -U_a U_b^T (inverse depth: U = W sqrt(D^-1), 6x1)
    // or -W_a Sigma W_b^T (XYZ: W 6x3 from the slot records, Sigma = (H_ll + lambda I)^-1 from the landmark's record)
    for (int it = ib + l16; it < ie; it += LP) {
        const int2 itm = nxt;  // indices were fetched one trip ahead: one dependent round trip per item, not two
        const int lm = nxt_lm;
        if (it + LP < ie) {
            nxt = items[it + LP];
            if (LD == 3) nxt_lm = B.slot_lm[d.obs0 + nxt.x];
        }
        const double* qa = slots + SS * (size_t)itm.x;
        const double* qb = slots + SS * (size_t)itm.y;
        double UA[6 * LD], UB[6 * LD];
#pragma unroll
        for (int i = 0; i < 6 * LD; i++) { UA[i] = qa[i]; UB[i] = qb[i]; }
        if (LD == 3) {
            const double* sg = B.prec + VBA_PREC * (size_t)(d.pt0 + lm) + 10;
            const double s00 = sg[0], s01 = sg[1], s02 = sg[2], s11 = sg[3], s12 = sg[4], s22 = sg[5];
#pragma unroll
            for (int i = 0; i < 6; i++) {   // UA <- W_a Sigma
                const double w0 = UA[3 * i], w1 = UA[3 * i + 1], w2 = UA[3 * i + 2];
                UA[3 * i] = w0 * s00 + w1 * s01 + w2 * s02;
                UA[3 * i + 1] = w0 * s01 + w1 * s11 + w2 * s12;
                UA[3 * i + 2] = w0 * s02 + w1 * s12 + w2 * s22;
            }
        }
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = 0; j < 6; j++)
#pragma unroll
                for (int l = 0; l < LD; l++) acc[6 * i + j] -= UA[LD * i + l] * UB[LD * j + l];
    }
    }
    if (LD == 1) {
        // the items [im, ie) involve the landmark's reference keyframe and also carry a direct H_pp term: the edge of the
        // other keyframe adds Br^T Bi (a = ref) or Bi^T Br (b = ref), with Br = [-A | A N0] rebuilt from the record's
        // A = Bi[:, 0:3].  Their own loop: the walk above stays free of this branch and of its loads.
        // (indices one trip ahead, as above: item -> landmark of its reference record -> landmark is two dependent fetches otherwise)
        int2 rnx = make_int2(0, 0);
        int rnx_lm = 0;
        if (im + l16 < ie) {
            rnx = items[im + l16];
            rnx_lm = B.rec_lm[d.pt0 + ((rnx.x >= d.n_obs) ? rnx.x : rnx.y) - d.n_obs];
        }
        for (int it = im + l16; it < ie; it += LP) {
            const int2 itm = rnx;
            const size_t gp = d.pt0 + rnx_lm;   // the landmark of the reference record
            if (it + LP < ie) {
                rnx = items[it + LP];
                rnx_lm = B.rec_lm[d.pt0 + ((rnx.x >= d.n_obs) ? rnx.x : rnx.y) - d.n_obs];
            }
            const int sa = itm.x, sb = itm.y;
            const bool a_ref = sa >= d.n_obs;
            const double* re = B.erec + VBA_EREC1 * (size_t)(d.obs0 + (a_ref ? sb : sa));
            const double rho = B.pt[3 * gp], xb = B.pt[3 * gp + 1], yb = B.pt[3 * gp + 2];
            const double* Ro = B.kfR + 12 * (size_t)(d.kf0 + (a_ref ? b : a));  // the observer's rotation
            double rec[6], bi0[6], bi1[6], br0[6], br1[6], rr0, rr1;
#pragma unroll
            for (int i = 0; i < 4; i++) rec[i] = re[i];
            rec[4] = rec[5] = 0.0;
            rebuild_edge(d, Ro, rec, bi0, bi1, rr0, rr1);   // (the rotation is read where it is used: nine registers fewer in flight)
            {   // N0 = R0 hat(b0) of the landmark in its reference keyframe, as the linearisation forms it (lin2_body, phase A): the
                // landmark has not moved since (the update comes after the solve).  Br = [-A | A N0], both rows, then N0 is dead.
                const double* C0 = B.kfR + 12 * (size_t)(d.kf0 + (a_ref ? a : b));
                double dd, c0[3], b0[3], Xw[3], Hb[9], N0[9];
                idp_point_world(d, C0, rho, xb, yb, dd, c0, b0, Xw);
                hat3(b0, Hb);
                mm3(C0, Hb, N0);
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    br0[k] = -bi0[k]; br1[k] = -bi1[k];
                    br0[3 + k] = bi0[0] * N0[k] + bi0[1] * N0[3 + k] + bi0[2] * N0[6 + k];
                    br1[3 + k] = bi1[0] * N0[k] + bi1[1] * N0[3 + k] + bi1[2] * N0[6 + k];
                }
            }
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const double* bi = h ? bi1 : bi0;
                const double* br = h ? br1 : br0;
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    const double x = a_ref ? br[i] : bi[i];
#pragma unroll
                    for (int j = 0; j < 6; j++) acc[6 * i + j] += x * (a_ref ? bi[j] : br[j]);
                }
            }
        }
    }
    // fixed-order sums inside each LP-lane group (skipped when no pair of the wave has an item: most keyframe pairs of a window
    // share no landmark, and in distance order they fill whole waves)
    if (__ballot(ib < ie) != 0ull) {
#pragma unroll
        for (int i = 0; i < 36; i++) acc[i] = (LP == 16) ? row16_sum(acc[i]) : wave64_sum(acc[i]);
    }
    double* blk = blk4 + g * 36;
    if (l16 == 0) {
#pragma unroll
        for (int i = 0; i < 36; i++) blk[i] = acc[i];
    }
    __syncthreads();
    if (have) schur_write_block(B, d, c, w, pr, a, b, blk, l16, LP, 6);
}
__global__ void __launch_bounds__(64, 2) k_schur_off(Batch B, int max_quads) {   // (the split form, VBA_SCHUR_SPLIT: not the default path)
    __shared__ double blk4[4 * 36];
    schur_off_body<1, 16>(B, max_quads, blk4);
}
__global__ void __launch_bounds__(64) k_schur_off_w(Batch B, int max_quads) {  // one pair per wave
    __shared__ double blk4[4 * 36];
    schur_off_body<1, 64>(B, max_quads, blk4);
}
__global__ void __launch_bounds__(64) k_schur_off3(Batch B, int max_quads) {
    __shared__ double blk4[4 * 36];
    schur_off_body<3, 16>(B, max_quads, blk4);
}
__global__ void __launch_bounds__(64) k_schur_off3_w(Batch B, int max_quads) {
    __shared__ double blk4[4 * 36];
    schur_off_body<3, 64>(B, max_quads, blk4);
}

// diagonal pairs (a,a): every slot of keyframe a; also the reduced rhs (block_solver.hpp:436-439), the
// unreduced b_p and the H_pp diagonal (LM's lambda init)
template <int LD>
DEVI void schur_diag_body(const Batch& B, int max_free, int hd_pass, double* blk, double* sh_r, double* sh_b, double* sh_h,
                          int w_in = -1, int a_in = 0) {
    constexpr int SS = (LD == 1) ? VBA_SLOT : VBA_SLOT3;
    int w = w_in, a = a_in;
    if (w_in < 0 && !schur_map(B, max_free, w, a)) return;
    const WinDesc& d = B.desc[w];
    const WinCtrl& c = B.ctrl[w];
    // hd_pass: LM's pre-trial pass for computeLambdaInit -- part of the "outer" slot, which a window that still owes a trial skips
    if (hd_pass ? (!c.active || c.lm_need_trial) : !win_on(d, c)) return;
    if (a >= d.n_free) return;
    const int t = threadIdx.x;
    const int pr = a * d.n_free - a * (a - 1) / 2;  // index of pair (a,a)
    double acc[21], rhs[6], bp[6], hd[6];  // the block is symmetric: upper triangle only, row-major (i <= j)
#pragma unroll
    for (int i = 0; i < 21; i++) acc[i] = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) rhs[i] = bp[i] = hd[i] = 0;
    // the items of the diagonal pair are the records of keyframe a: its slot records (observer) and, for inverse-depth
    // landmarks, the landmark records it is the reference keyframe of -- two index ranges, no list
    const int* kseg = B.kf_seg + d.kf0 + d.win;
    const int s0 = kseg[a], n_o = kseg[a + 1] - s0;
    const int r0s = (LD == 1) ? B.ref_seg[d.kf0 + d.win + a] : 0, n_r = (LD == 1) ? B.ref_seg[d.kf0 + d.win + a + 1] - r0s : 0;
    const double* slots = B.slot + SS * (size_t)(d.obs0 + d.pt0);
    double Ra[9];  // rotation of keyframe a (the observer of every non-reference item of its diagonal pair)
    {
        const double* Ca = B.kfR + 12 * (size_t)(d.kf0 + a);
#pragma unroll
        for (int i = 0; i < 9; i++) Ra[i] = Ca[i];
    }
    // inverse depth: a third range of items -- the run records of this keyframe's reference terms (G0: 21, g0: 6 summed over runs of
    // landmarks by k_lin2, phase E).  A lane's first run item has its index fetched now: inside the walk only the record itself is a
    // dependent fetch, like a slot record.
    int n_run = 0, run_first = -1;
    const int* run_list = nullptr;
    if (LD == 1) {
        const int* pfb = B.pref_begin + d.kf0 + d.win;
        const int m0 = pfb[a];
        n_run = pfb[a + 1] - m0;
        run_list = B.pref_list + d.pt0 + m0;
        const int j0 = (t - (n_o + n_r)) & 63;   // the first run item this lane meets in the walk
        if (j0 < n_run) run_first = run_list[j0];
    }
    for (int it = t; it < n_o + n_r + n_run; it += 64) {
        if (LD == 1 && it >= n_o + n_r) {   // a run record: direct terms only
            const int j = it - (n_o + n_r);
            const double* pr_ = B.prec + VBA_PREC * (size_t)(d.pt0 + ((j < 64) ? run_first : run_list[j]));
#pragma unroll
            for (int g = 0; g < 21; g++) acc[g] += pr_[g];
            // (no H_pp diagonal here: it feeds Levenberg-Marquardt's lambda init, and inverse-depth windows run Gauss-Newton)
#pragma unroll
            for (int i = 0; i < 6; i++) bp[i] += pr_[21 + i];
            continue;
        }
        const int sa = (it < n_o) ? s0 + it : d.n_obs + r0s + (it - n_o);
        // XYZ landmarks (Levenberg-Marquardt): the direct terms sum Bi^T Bi and b_p do not depend on the damping, so the pass
        // that opens an outer iteration (hd_pass) takes them from the edge records ONCE and parks them per keyframe (kf_dir);
        // the trials then read the slot records U(lambda) only -- 192 B instead of 384 B per record and trial
        const bool want_u = (LD == 1) || !hd_pass, want_dir = (LD == 1) || hd_pass;
        const double* qa = slots + SS * (size_t)sa;
        double UA[6 * LD], UB[6 * LD], beta[LD];   // XYZ: UA = W_a Sigma, UB = W_a, beta = t = Sigma b_l; inverse depth: UA = UB = U, beta
#pragma unroll
        for (int i = 0; i < 6 * LD; i++) UB[i] = want_u ? qa[i] : 0.0;
        if (LD == 1) {
#pragma unroll
            for (int i = 0; i < 6 * LD; i++) UA[i] = UB[i];
            beta[0] = qa[6];
        } else {
            // (the pass that opens an outer iteration reads no Sigma at all: the record may hold anything there -- after a failed
            // solve of the window that occupied this place before, a NaN, and 0 * NaN would poison the whole keyframe block)
            const double* sg = B.prec + VBA_PREC * (size_t)(d.pt0 + (want_u ? B.slot_lm[d.obs0 + sa] : 0)) + 10;
            double s00 = 0.0, s01 = 0.0, s02 = 0.0, s11 = 0.0, s12 = 0.0, s22 = 0.0;
            if (want_u) { s00 = sg[0]; s01 = sg[1]; s02 = sg[2]; s11 = sg[3]; s12 = sg[4]; s22 = sg[5]; }
#pragma unroll
            for (int l = 0; l < LD; l++) beta[l] = want_u ? sg[6 + l] : 0.0;
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const double w0 = UB[3 * i], w1 = UB[3 * i + 1], w2 = UB[3 * i + 2];
                UA[LD * i] = w0 * s00 + w1 * s01 + w2 * s02;
                UA[LD * i + 1] = w0 * s01 + w1 * s11 + w2 * s12;
                UA[LD * i + 2] = w0 * s02 + w1 * s12 + w2 * s22;
            }
        }
        if (!want_dir) {
        } else if (LD == 1 && sa >= d.n_obs) {
            // (a landmark record of this keyframe: its direct terms come summed over runs of landmarks, below)
        } else {
            const double* ra = B.erec + ((LD == 1) ? VBA_EREC1 : VBA_EREC) * (size_t)(d.obs0 + sa);
            double r0 = 0.0, r1 = 0.0;
            double b0[6], b1[6];
            if (LD == 1) {
                double rec[6];
#pragma unroll
                for (int i = 0; i < 6; i++) rec[i] = ra[i];
                rebuild_edge(d, Ra, rec, b0, b1, r0, r1);
            } else {
#pragma unroll
                for (int i = 0; i < 6; i++) { b0[i] = ra[i]; b1[i] = ra[6 + i]; }   // XYZ edge record: Bi (12), g (6)
            }
            int gi = 0;
#pragma unroll
            for (int i = 0; i < 6; i++) {
#pragma unroll
                for (int j = i; j < 6; j++) acc[gi++] += b0[i] * b0[j] + b1[i] * b1[j];
                bp[i] += (LD == 1) ? -(b0[i] * r0 + b1[i] * r1) : ra[12 + i];
            }
        }
        int gi = 0;
#pragma unroll
        for (int i = 0; i < 6; i++) {
#pragma unroll
            for (int l = 0; l < LD; l++) rhs[i] -= UB[LD * i + l] * beta[l];   // W_a t  (inverse depth: U beta)
#pragma unroll
            for (int j = i; j < 6; j++) {
#pragma unroll
                for (int l = 0; l < LD; l++) acc[gi] -= UA[LD * i + l] * UB[LD * j + l];
                gi++;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 21; i++) acc[i] = wave_sum(acc[i]);
#pragma unroll
    for (int i = 0; i < 6; i++) { rhs[i] = wave_sum(rhs[i]); bp[i] = wave_sum(bp[i]); }
    if (LD != 1) {
        double* kd = B.kf_dir + 32 * (size_t)(d.kf0 + a);
        if (hd_pass) {
            if (t == 0) {
#pragma unroll
                for (int g = 0; g < 21; g++) kd[g] = acc[g];
#pragma unroll
                for (int i = 0; i < 6; i++) kd[21 + i] = bp[i];
            }
            hd[0] = acc[0]; hd[1] = acc[6]; hd[2] = acc[11]; hd[3] = acc[15]; hd[4] = acc[18]; hd[5] = acc[20];
        } else {
#pragma unroll
            for (int g = 0; g < 21; g++) acc[g] += kd[g];
#pragma unroll
            for (int i = 0; i < 6; i++) bp[i] = kd[21 + i];
        }
    }
    const int P = d.pdim;
    for (int q = t; q < P * P; q += 64) blk[q] = 0.0;
    __syncthreads();
    if (t == 0) {
        int gi = 0;
#pragma unroll
        for (int i = 0; i < 6; i++) {
#pragma unroll
            for (int j = i; j < 6; j++) { blk[i * P + j] = acc[gi]; blk[j * P + i] = acc[gi]; gi++; }
            sh_r[i] = rhs[i]; sh_b[i] = bp[i]; sh_h[i] = hd[i];
        }
    }
    __syncthreads();
    if (!hd_pass) schur_write_block(B, d, c, w, pr, a, a, blk);   // (the hd pass only feeds lambda init and b_p)
    if (t < P) {
        double s = 0.0, sb = 0.0, h = 0.0;
        if (t < 6) { s = sh_r[t]; sb = sh_b[t]; h = sh_h[t]; }
        const int qb = B.pimu_begin[d.pair0 + d.win + pr], qe = B.pimu_begin[d.pair0 + d.win + pr + 1];
        for (int m = qb; m < qe; m++) {
            const int k = B.pimu[2 * (size_t)(d.pimu0 + m)], role = B.pimu[2 * (size_t)(d.pimu0 + m) + 1];
            const double* H = B.imuH + VBA_IMUH * (size_t)(d.imu0 + k);
            const int ro = (role & 1) ? 15 : 0;
            sb += H[900 + ro + t];
            h += H[(ro + t) * 30 + ro + t];
        }
        const int gr = vpos(d, a, t);
        const bool act = B.var_act[d.vec0 + gr] != 0;
        (B.vec + d.vec0)[gr] = act ? (sb + s) : 0.0;            // reduced rhs = b_p - sum W Dinv b_l
        (B.bpose + 2 * (size_t)d.vec0)[gr] = act ? sb : 0.0;        // unreduced b_p (LM computeScale)
        if (LD == 1 || hd_pass) (B.bpose + 2 * (size_t)d.vec0)[d.nS + gr] = act ? h : 0.0;  // H_pp diagonal (LM computeLambdaInit)
    }
}
__global__ void __launch_bounds__(64) k_schur_diag(Batch B, int max_free) {
    __shared__ double blk[15 * 15 + 16];
    __shared__ double sh_r[6], sh_b[6], sh_h[6];
    schur_diag_body<1>(B, max_free, 0, blk, sh_r, sh_b, sh_h);
}
// Both Schur kernels of the inverse-depth variant in ONE launch, a window's diagonal pairs right before its off-diagonal
// ones: they read the same slot and edge records, which are then still in the XCD's L2 for the second reader.
__global__ void __launch_bounds__(64, 3) k_schur_all(Batch B, int max_free, int max_quads) {
    __shared__ double blk[15 * 15 + 16];
    __shared__ double sh_r[6], sh_b[6], sh_h[6];
    int w, idx;
    if (!schur_map(B, max_free + max_quads, w, idx)) return;
    if (idx < max_free) schur_diag_body<1>(B, max_free, 0, blk, sh_r, sh_b, sh_h, w, idx);
    else schur_off_body<1, 16>(B, max_quads, blk, w, idx - max_free);
}
// the same for fewer than 8 windows: one off-diagonal pair per wave (latency), still one launch
__global__ void __launch_bounds__(64) k_schur_all_w(Batch B, int max_free, int max_offp) {
    __shared__ double blk[15 * 15 + 16];
    __shared__ double sh_r[6], sh_b[6], sh_h[6];
    int w, idx;
    if (!schur_map(B, max_free + max_offp, w, idx)) return;
    if (idx < max_free) schur_diag_body<1>(B, max_free, 0, blk, sh_r, sh_b, sh_h, w, idx);
    else schur_off_body<1, 64>(B, max_offp, blk, w, idx - max_free);
}
__global__ void __launch_bounds__(64) k_schur_diag3(Batch B, int max_free, int hd_pass) {
    __shared__ double blk[15 * 15 + 16];
    __shared__ double sh_r[6], sh_b[6], sh_h[6];
    schur_diag_body<3>(B, max_free, hd_pass, blk, sh_r, sh_b, sh_h);
}

// ------------------------------------------------------------------------------------------------
// Reduced-system factorisation S = L L^T on 32x32 tiles, right-looking, ONE launch per block column:
// every workgroup (one wave) of step k owns one tile pair (I,J) of the trailing update and redoes, on its
// own, the little work it depends on -- POTRF of the diagonal tile (registers + v_readlane broadcasts, no
// LDS round trips on the critical path) and the triangular solves of the two panel tiles L_Ik, L_Jk -- then
// applies C_IJ -= L_Ik L_Jk^T with v_mfma_f64_16x16x4_f64.  No inter-workgroup dependency inside a launch.
// The right-hand side rides along as one more row of the panel solve (forward substitution for free).
// Tiles that are structurally zero in L (V/Bias-first ordering) are not in the lists and never touched.
// Restates LinearSolverEigen::solve (linear_solver_eigen.h:94-124) on the dense reduced system.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_chol_step(Batch B, int k) {
    __shared__ double XI[32 * 34];   // first holds L_kk (stride 33) for the panel solves, then X_I (stride 34)
    __shared__ double XJ[32 * 34];
    __shared__ double rd[32];
    double* Lk = XI;
    const int w = blockIdx.y;
    const WinDesc& d = B.desc[w];
    WinCtrl& c = B.ctrl[w];
    if (!win_on(d, c)) return;
    if (k >= d.nb || k < d.nc) return;   // (columns [0, nc): k_chol_chain)
    const int* sb = B.tl_step_begin + d.tl_step0;
    const int npair = sb[k + 1] - sb[k];
    const int bx = blockIdx.x;
    if (bx >= (npair > 0 ? npair : 1)) return;
    const bool has_pair = bx < npair;
    int I = 0, J = 0;
    if (has_pair) {
        const int v = B.tl_pairs[d.tl_pair0 + sb[k] + bx];
        I = v >> 16;
        J = v & 0xffff;
    }
    const int lane = threadIdx.x, r = lane & 31, hi = lane >> 5;
    const int n = d.nS;
    double* S = B.S + d.S0;
    double* Lf = B.Lf + d.S0;
    double* vec = B.vec + d.vec0;
    double* yv = B.yv + d.vec0;
    const size_t dk = (size_t)k * 32;
    const bool diagp = has_pair && (I == J);
    const bool rhs_lane = (!has_pair || diagp) && lane == 32;
    const bool row_act = has_pair && (hi == 0 || !diagp);
    // panel rows first: their loads are in flight while the diagonal tile is factored
    double x[32];
    {
        const double* src = row_act ? (S + ((size_t)(hi ? J : I) * 32 + r) * n + dk) : (vec + dk);
        const bool ld = row_act || rhs_lane;
#pragma unroll
        for (int q = 0; q < 32; q++) x[q] = ld ? src[q] : 0.0;
    }
    double a[32];
    const double* arow = S + (dk + r) * n + dk;
#pragma unroll
    for (int q = 0; q < 32; q++) a[q] = (q <= r) ? arow[q] : 0.0;
    // the trailing tile C_IJ is requested now as well: it arrives while the diagonal tile is factored and the panels solved
    const int l15 = lane & 15, l4 = lane >> 4;
    d4_t cacc[2][2];
#pragma unroll
    for (int ti = 0; ti < 2; ti++)
#pragma unroll
        for (int tj = 0; tj < 2; tj++) {
            const double* C = S + ((size_t)I * 32 + 16 * ti) * n + (size_t)J * 32 + 16 * tj;
#pragma unroll
            for (int i = 0; i < 4; i++) cacc[ti][tj][i] = has_pair ? C[(size_t)(l4 + 4 * i) * n + l15] : 0.0;
        }
    // L D L^T of the diagonal tile (unit lower L, D on the diagonal), as Eigen's SimplicialLDLT does: negative
    // pivots are fine, only an exactly zero / non-finite pivot fails (linear_solver_eigen.h:105-111)
    bool bad = false;
    double rdiag = 1.0;  // 1 / d_r of this lane's row
#pragma unroll
    for (int cc = 0; cc < 32; cc++) {
        const double piv = rl64(a[cc], cc);
        bad = bad || (piv == 0.0) || !isfinite(piv);
        double y = __builtin_amdgcn_rcp(piv);  // v_rcp_f64 + two Newton steps: full double accuracy
        y = y * (2.0 - piv * y);
        y = y * (2.0 - piv * y);
        const double u = a[cc];                 // a_{r,cc} before the division = l_{r,cc} d_cc
        const double l = u * y;
        rdiag = (r == cc) ? y : rdiag;
        a[cc] = (r == cc) ? piv : l;
#pragma unroll
        for (int c2 = cc + 1; c2 < 32; c2++) a[c2] -= l * rl64(u, c2);
    }
    if (hi == 0) {
#pragma unroll
        for (int q = 0; q < 32; q++) Lk[r * 33 + q] = (q <= r) ? a[q] : 0.0;
        rd[r] = rdiag;
        if (bx == 0) {
            double* lrow = Lf + (dk + r) * n + dk;
#pragma unroll
            for (int q = 0; q < 32; q++)
                if (q <= r) lrow[q] = a[q];
        }
    }
    if (bx == 0 && lane == 0 && bad) c.chol_fail = 1;
    __syncthreads();
    // X' L_kk^T = A with unit-diagonal L (row per lane, column-oriented so the 31-q updates of a step are
    // independent); the panel factor is L_Ik = X' D^-1, the forward-substituted rhs z_k = L_kk^-1 r_k
#pragma unroll
    for (int q = 0; q < 32; q++) {
        const double xq = x[q];
#pragma unroll
        for (int c2 = q + 1; c2 < 32; c2++) x[c2] -= xq * Lk[c2 * 33 + q];
    }
    if (bx == 0 && lane == 32) {
#pragma unroll
        for (int q = 0; q < 32; q++) yv[dk + q] = x[q];  // z_k
    }
    if (!has_pair) return;
    double xs[32];   // row of L_Ik = X' D^-1
#pragma unroll
    for (int q = 0; q < 32; q++) xs[q] = x[q] * rd[q];
    double sy = 0.0;  // L_Ik row . z_k ; z_k sits in lane 32 of a diagonal pair (wave-uniform control flow here)
#pragma unroll
    for (int q = 0; q < 32; q++) sy += xs[q] * rl64(x[q], 32);
    if (diagp && hi == 0) {
        double* dst = Lf + ((size_t)I * 32 + r) * n + dk;
#pragma unroll
        for (int q = 0; q < 32; q++) dst[q] = xs[q];  // L_Ik
        vec[(size_t)I * 32 + r] -= sy;
    }
    __syncthreads();  // every lane is done reading L_kk before X_I overwrites it
    // C_IJ -= L_Ik D L_Jk^T = (X'_I D^-1) X'_J^T : XI holds the scaled rows of tile I, XJ the unscaled rows of tile J
    if (hi == 0) {
#pragma unroll
        for (int q = 0; q < 32; q++) XI[r * 34 + q] = xs[q];
        if (diagp) {
#pragma unroll
            for (int q = 0; q < 32; q++) XJ[r * 34 + q] = x[q];
        }
    } else if (!diagp) {
#pragma unroll
        for (int q = 0; q < 32; q++) XJ[r * 34 + q] = x[q];
    }
    __syncthreads();
    const double* XJp = XJ;
#pragma unroll
    for (int ti = 0; ti < 2; ti++)
#pragma unroll
        for (int tj = 0; tj < 2; tj++) {
            if (diagp && tj > ti) continue;
            double* C = S + ((size_t)I * 32 + 16 * ti) * n + (size_t)J * 32 + 16 * tj;
            d4_t acc = cacc[ti][tj];
#pragma unroll
            for (int ks = 0; ks < 8; ks++) {
                const double av = -XI[(16 * ti + l15) * 34 + 4 * ks + l4];
                const double bv = XJp[(16 * tj + l15) * 34 + 4 * ks + l4];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; i++) C[(size_t)(l4 + 4 * i) * n + l15] = acc[i];
        }
}

// ------------------------------------------------------------------------------------------------
// The same step for FEW windows (latency is everything: one window = a chain of nb dependent launches), built around
// `v_fmac_f64_dpp ... row_newbcast:n` (gfx90a+: the only DPP control the FP64 pipe takes -- lane n of every 16-lane row is
// broadcast to the row).  In the form above every element of a rank-1 update costs three instructions: two v_readlane_b32 to bring
// u_{c2} (held by lane c2) into a scalar pair, one FMA; the 32 x 32 elimination is ~1 500 of them per row set and a lone wave
// issues one every four to six cycles.  With the broadcast inside the FMA an element costs ONE instruction.  Layout:
//   lanes 0..31 of a wave hold the 32 rows of the diagonal tile, lanes 32..63 the rows of ONE panel tile, in the same 32 registers
//   t[] (t[c] = column c of the lane's row); wave 0 of the workgroup takes tile (I,k), wave 1 tile (J,k) -- both eliminate the
//   diagonal tile (the same instructions, hence the same bits) on their own SIMD, so a pivot costs 31 - cc FMACs for diagonal and
//   panel rows together (round 3's first DPP form, k_chol_step3, kept the panel rows of both tiles in a second array in one wave:
//   2 (31 - cc) FMACs per pivot, ~66 instructions; 9.8 k cycles per elimination against 7.0 k here);
//   at pivot cc, u = t[cc] of lanes 0..31 is u_0..u_31; the broadcast operand must sit in every 16-lane row, so u_0..15 / u_16..31
//   are replicated with ds_bpermute (crossbar only, no LDS memory) into uLow / uHigh and
//   t[c2] -= l * u_{c2}  is  v_fmac_f64_dpp t[c2], -(c2 < 16 ? uLow : uHigh), l  row_newbcast:(c2 & 15)
//   (the sign rides in the DPP word's source modifier: no negated copy of l).
// The hand-written instruction reads its broadcast operand through DPP: the two wait states a DPP read needs after a VALU write of
// that register (the compiler does not look into inline assembly) are the s_nop in front of each pivot's first FMAC.
// Zero / non-finite pivots are looked for once, in the d vector, after the elimination.
// Same arithmetic as k_chol_step up to the association of l = a / d: results agree to rounding (and bit for bit with k_chol_step3,
// checked before that kernel was removed: scripts/step_forms.py).
// ------------------------------------------------------------------------------------------------
template <int LANE>
DEVI double wl64(double old, double v) {  // old with lane LANE replaced by the wave-uniform value v (two v_writelane_b32)
    int lo = __double2loint(old), hi = __double2hiint(old);
    const int vlo = __builtin_amdgcn_readfirstlane(__double2loint(v)), vhi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(lo) : "s"(vlo), "n"(LANE));
    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(hi) : "s"(vhi), "n"(LANE));
    return __hiloint2double(hi, lo);
}
template <int N, bool NOP>
DEVI void fnmac_bcast(double& acc, double urep, double s) {   // acc -= urep[lane N of this lane's row] * s
    if (NOP) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(urep), "v"(s), "n"(N));
    else asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(urep), "v"(s), "n"(N));
}
DEVI double bperm64(double v, int byte_addr) {   // the value of lane byte_addr / 4
    const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
template <int C2, int CC>
struct ElimUpd {   // columns C2..31 of pivot CC
    static DEVI void run(double (&t)[32], double uLow, double uHigh, double l) {
        if constexpr (C2 < 32) {
            fnmac_bcast<(C2 & 15), false>(t[C2], (C2 < 16) ? uLow : uHigh, l);
            ElimUpd<C2 + 1, CC>::run(t, uLow, uHigh, l);
        }
    }
};
// One pivot.  No per-lane masks anywhere: the entries of a diagonal row above the diagonal are never read by another lane (u_{c2}
// comes from lane c2 > CC, the pivot from lane CC) nor stored where anybody uses them, so they may hold anything -- lane CC's own
// update with l = 1 and the lanes r < CC run through the same instructions; what a lane must KEEP from pivot CC (d_CC, z_CC) is
// captured with v_writelane into lane CC of dout / zout before the unguarded updates overwrite it.
// uLow / uHigh are column CC replicated into every 16-lane row; those of the NEXT pivot are requested as soon as its column is
// final (after the first FMAC) so that the crossbar round trip hides behind the remaining FMACs of this one.  (A fully hand-
// scheduled variant -- the reciprocal, l, z and the captures of the next pivot slotted between the FMACs -- measured the same.)
// WRITE_X: 0 nothing; 1 Xrow[CC] = l; 2 (vba_chain.h) Xrow[pc] = l and Xrow[pc + ELIM_U_OFF] = u = l d with pc = (CC & 3) * 8 + (CC >> 2): the
// eight k-steps a lane needs as an MFMA operand lie side by side, and the second operand (rows times D) needs no multiplication
#define ELIM_U_OFF (32 * 34)
template <int CC, int WRITE_X>
struct ElimStep {
    static DEVI void run(double (&t)[32], double& rr, double& dout, double& zout, double uLow, double uHigh, int aLow, int aHigh, double* Xrow) {
        if constexpr (CC < 32) {
            const double u = t[CC];                      // column CC before the division: l d
            const double piv = rl64(u, CC);              // d_CC (lane CC holds row CC of the diagonal tile)
            double y = __builtin_amdgcn_rcp(piv);        // v_rcp_f64 + two Newton steps: full double accuracy
            y = y * (2.0 - piv * y);
            y = y * (2.0 - piv * y);
            const double l = u * y;
            const double zc = rl64(rr, CC);              // z_CC is final here
            dout = wl64<CC>(dout, piv);
            zout = wl64<CC>(zout, zc);
            rr -= l * zc;
            if constexpr (WRITE_X == 1) Xrow[CC] = l;    // lanes 32..63: row r of L_Tk for the MFMA update (lanes 0..31: a scratch row)
            if constexpr (WRITE_X == 2) { Xrow[(CC & 3) * 8 + (CC >> 2)] = l; Xrow[(CC & 3) * 8 + (CC >> 2) + ELIM_U_OFF] = u; }
            t[CC] = l;
            double nLow = 0.0, nHigh = 0.0;
            if constexpr (CC < 31) {
                fnmac_bcast<((CC + 1) & 15), true>(t[CC + 1], (CC + 1 < 16) ? uLow : uHigh, l);
                if constexpr (CC + 1 < 15) nLow = bperm64(t[CC + 1], aLow);
                if constexpr (CC + 1 < 31) nHigh = bperm64(t[CC + 1], aHigh);
                __builtin_amdgcn_sched_barrier(0);       // (the requests stay in front of the remaining FMACs)
                ElimUpd<CC + 2, CC>::run(t, uLow, uHigh, l);
            }
            ElimStep<CC + 1, WRITE_X>::run(t, rr, dout, zout, nLow, nHigh, aLow, aHigh, Xrow);
        }
    }
};
template <int WRITE_X>
DEVI void elim_tile(double (&t)[32], double& rr, double& dout, double& zout, int aLow, int aHigh, double* Xrow) {
    ElimStep<0, WRITE_X>::run(t, rr, dout, zout, bperm64(t[0], aLow), bperm64(t[0], aHigh), aLow, aHigh, Xrow);
}

// ONE = a single window: what the kernel would fetch from the window descriptor and the step table (descriptor -> step table ->
// pair list -> tile rows: four dependent loads, ~600 cycles each, in front of the elimination) comes in the kernel arguments, and
// the chain is pair list -> tile rows.
struct StepOne { int algo, nS, nb, vec0, pair_off, npair; long long S0; };
template <bool ONE>
__global__ void __launch_bounds__(128) k_chol_step4(Batch B, int k, StepOne so) {
    __shared__ double XI[32 * 34];   // rows of L_Ik            (A operand: -L_Ik)
    __shared__ double XJ[32 * 34];   // rows of L_Jk            (B operand: L_Jk D_k, scaled when it is read)
    __shared__ double XD[32 * 34];   // where the diagonal lanes' Xrow stores go (never read)
    __shared__ double dg[32];
    const int w = blockIdx.y;
    WinCtrl& c = B.ctrl[w];
    int n, npair, pair_off;
    long long S0;
    int vec0;
    if constexpr (ONE) {
        if (!(c.active && (so.algo == 0 || c.lm_need_trial))) return;   // win_on
        if (k >= so.nb) return;
        n = so.nS; npair = so.npair; pair_off = so.pair_off; S0 = so.S0; vec0 = so.vec0;
    } else {
        const WinDesc& d = B.desc[w];
        if (!win_on(d, c)) return;
        if (k >= d.nb || k < d.nc) return;
        const int* sb = B.tl_step_begin + d.tl_step0;
        npair = sb[k + 1] - sb[k];
        pair_off = d.tl_pair0 + sb[k];
        n = d.nS; S0 = d.S0; vec0 = d.vec0;
    }
    const int bx = blockIdx.x;
    if (bx >= (npair > 0 ? npair : 1)) return;
    const bool has_pair = bx < npair;
    int I = 0, J = 0;
    if (has_pair) {
        const int v = B.tl_pairs[pair_off + bx];
        I = v >> 16;
        J = v & 0xffff;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5;
    double* S = B.S + S0;
    double* Lf = B.Lf + S0;
    double* vec = B.vec + vec0;
    double* yv = B.yv + vec0;
    const size_t dk = (size_t)k * 32;
    const bool diagp = has_pair && (I == J);
#ifdef VBA_STAMPS   // diagnostic build only (scripts/stamps.sh): shader-clock stamps of wave 0 of workgroup 1 of column 5 into B.dbg
    unsigned long long st_[8];
#define STAMP(i) { if (k == 5 && bx == 1 && wave == 0) { st_[i] = __builtin_amdgcn_s_memtime(); } }
    STAMP(0)
#else
#define STAMP(i)
#endif
    const int l15 = lane & 15, l4 = lane >> 4;
    const bool elim = wave == 0 || (has_pair && !diagp);
    const int T = wave ? J : I;                      // this wave's panel tile
    double t[32], rr = 0.0;
    if (elim) {
        {   // whole rows, 32-byte pieces (rows start on 256-byte boundaries: nS is a multiple of 32)
            const double4* row = reinterpret_cast<const double4*>(S + (hi ? (size_t)T * 32 + r : dk + r) * n + dk);
            if (!hi || has_pair) {
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const double4 v = row[q];
                    t[4 * q] = v.x; t[4 * q + 1] = v.y; t[4 * q + 2] = v.z; t[4 * q + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 32; q++) t[q] = 0.0;
            }
        }
        // right-hand side: one more column.  The rhs rows of a panel tile are updated by ONE workgroup, the one of its diagonal pair
        rr = hi ? ((diagp && wave == 0) ? vec[(size_t)I * 32 + r] : 0.0) : vec[dk + r];
    }
    // this wave's half of the trailing tile C_IJ (rows 16 wave .. 16 wave + 15), requested behind the rows the elimination waits
    // for: it arrives while the elimination runs
    d4_t cacc[2];
#pragma unroll
    for (int tj = 0; tj < 2; tj++) {
        const double* C = S + ((size_t)I * 32 + 16 * wave) * n + (size_t)J * 32 + 16 * tj;
#pragma unroll
        for (int i = 0; i < 4; i++) cacc[tj][i] = has_pair ? C[(size_t)(l4 + 4 * i) * n + l15] : 0.0;
    }
    if (elim) {
        double dout = 1.0, zout = 0.0;               // lane r < 32 ends up with d_r and z_r
#ifdef VBA_STAMPS
        { double sink = t[0] + t[31] + rr; asm volatile("" :: "v"(sink)); }   // the loads have landed
#endif
        STAMP(1)
        elim_tile<1>(t, rr, dout, zout, 4 * l15, 4 * (16 + l15), (hi ? (wave ? XJ : XI) : XD) + r * 34);
        STAMP(2)
        if (wave == 0) {
            if (bx == 0 && !hi) {                    // the factor's diagonal tile (unit L below the diagonal, D on it) and z_k = L_kk^-1 r_k
                double4* lrow = reinterpret_cast<double4*>(Lf + (dk + r) * n + dk);
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    double4 v;
                    // (what the elimination leaves above the diagonal is arbitrary: zeros go into the factor)
                    v.x = (4 * q == r) ? dout : ((4 * q < r) ? t[4 * q] : 0.0);
                    v.y = (4 * q + 1 == r) ? dout : ((4 * q + 1 < r) ? t[4 * q + 1] : 0.0);
                    v.z = (4 * q + 2 == r) ? dout : ((4 * q + 2 < r) ? t[4 * q + 2] : 0.0);
                    v.w = (4 * q + 3 == r) ? dout : ((4 * q + 3 < r) ? t[4 * q + 3] : 0.0);
                    lrow[q] = v;
                }
                yv[dk + r] = zout;
            }
            if (bx == 0) {
                const bool badl = !hi && (dout == 0.0 || !isfinite(dout));
                if (__ballot(badl) != 0ull && lane == 0) c.chol_fail = 1;
            }
            if (diagp && hi) {                       // tile (I,k) of the factor and the rhs rows it has updated
                double4* dst = reinterpret_cast<double4*>(Lf + ((size_t)I * 32 + r) * n + dk);
#pragma unroll
                for (int q = 0; q < 8; q++) dst[q] = make_double4(t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]);
                vec[(size_t)I * 32 + r] = rr;
            }
            if (!hi) dg[r] = dout;
        }
        STAMP(3)
    }
    if (!has_pair) return;                           // (the same for every thread of the workgroup)
    // C_IJ -= L_Ik D_k L_Jk^T: XI / XJ hold the rows of L_Ik / L_Jk (written pivot by pivot above), D_k is dg; 16 rows per wave.
    // (LDS-only barrier: the stores of the factor tiles above drain while the MFMAs run)
    lds_barrier();
    const double* XJp = diagp ? XI : XJ;
#pragma unroll
    for (int ks = 0; ks < 8; ks++) {    // (the wave's two products side by side: independent accumulators)
        const double av = -XI[(16 * wave + l15) * 34 + 4 * ks + l4];
        const double dvk = dg[4 * ks + l4];
#pragma unroll
        for (int tj = 0; tj < 2; tj++) {
            const double bv = XJp[(16 * tj + l15) * 34 + 4 * ks + l4] * dvk;
            cacc[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, cacc[tj], 0, 0, 0);
        }
    }
#pragma unroll
    for (int tj = 0; tj < 2; tj++) {
        if (diagp && tj > wave) continue;
        double* C = S + ((size_t)I * 32 + 16 * wave) * n + (size_t)J * 32 + 16 * tj;
#pragma unroll
        for (int i = 0; i < 4; i++) C[(size_t)(l4 + 4 * i) * n + l15] = cacc[tj][i];
    }
#ifdef VBA_STAMPS
    STAMP(4)
    if (wave == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); STAMP(5) }
    if (k == 5 && bx == 1 && threadIdx.x == 0)
        for (int i = 0; i < 6; i++) B.dbg[i] = (double)st_[i];
#endif
#undef STAMP
}

// ------------------------------------------------------------------------------------------------
// Left-looking form of the tile factorisation for LARGE batches.  The right-looking kernels above re-read and
// re-write every trailing tile once per step (~19 MB of HBM traffic per C3 factorisation); here a tile is read from S
// once, accumulates ALL its updates C_IJ = S_IJ - sum_k L_Ik D_k L_Jk^T in MFMA registers (same k order, hence the same
// rounding as the right-looking sweep), and is written once.  S itself is never modified.  Per block column J:
//   diagonal tile : C_JJ, its L D L^T, the forward-substituted rhs y_J, and W_J = L_JJ^-T D_J^-1 (so that the panel
//                   needs no serial triangular solve)
//   panel tile    : C_IJ, then L_IJ = C_IJ W_J as one more MFMA product.
// ------------------------------------------------------------------------------------------------
// In this mode the factor is stored TILE BY TILE -- tile (I,J) at 1024 (I nb + J) doubles -- in the order the MFMA operand
// registers want it: element (r, c) of a tile belongs to lane (c & 3) * 16 + (r & 15), the lane's 16 elements being the two row
// halves x eight k-steps of v_mfma_f64_16x16x4 (operand A[row][k]: row = lane & 15, k = 4 ks + (lane >> 4)), two k-steps per 16-byte
// piece.  A panel wave then loads both operand tiles of a product with 8 + 8 fully coalesced global_load_dwordx4 straight into
// the registers the MFMAs read: no LDS, no barrier, and the occupancy is set by the registers alone.
DEVI int ll_pk(int r, int c) { return (((((r >> 4) * 4 + (c >> 3)) * 64) + ((c & 3) * 16 + (r & 15))) * 2) + ((c >> 2) & 1); }
DEVI size_t ll_tile(const WinDesc& d, int I, int J) { return 1024 * ((size_t)I * d.nb + J); }

// diagonal tile of block column J: C_JJ = S_JJ - sum_k L_Jk D_k L_Jk^T and the dot products L_Jk y_k for the forward substitution.
// Both MFMA operands are the one packed tile L_Jk (A = -L_Jk, B = L_Jk D_k), loaded straight into registers.
DEVI void ll_diag_accumulate(const Batch& B, const WinDesc& d, int J, int kb, int ke, d4_t (&acc)[2][2], double& sdot_out) {
    const int lane = threadIdx.x, n = d.nS;
    const double* S = B.S + d.S0;
    const double* Lf = B.Lf + d.S0;
    const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
    for (int ti = 0; ti < 2; ti++)
#pragma unroll
        for (int tj = 0; tj < 2; tj++) {
            const double* C = S + ((size_t)J * 32 + 16 * ti) * n + (size_t)J * 32 + 16 * tj;
#pragma unroll
            for (int i = 0; i < 4; i++) acc[ti][tj][i] = C[(size_t)(l4 + 4 * i) * n + l15];
        }
    const int* kl = B.tl_kl + d.tl_k0;
    double p0 = 0.0, p1 = 0.0;   // this lane's share of (L_Jk y_k) for rows l15 and 16 + l15: columns 4 ks + l4
    // one wave per window, a chain of dependent products: the tile of step e+1 is in flight (registers) while the MFMAs of step e run
    double2 nx[8];
    double ndv[8], nyk[8];
    auto fetch = [&](int k) {
        const double2* t = reinterpret_cast<const double2*>(Lf + ll_tile(d, J, k)) + lane;
        const double* sd = B.dvec + d.vec0 + (size_t)k * 32 + l4;
        const double* sy = B.yv + d.vec0 + (size_t)k * 32 + l4;
#pragma unroll
        for (int q = 0; q < 8; q++) { nx[q] = t[64 * q]; ndv[q] = sd[4 * q]; nyk[q] = sy[4 * q]; }
    };
    if (kb < ke) fetch(kl[kb]);
    for (int e = kb; e < ke; e++) {
        double2 x[8];
        double dv[8], yk[8];
#pragma unroll
        for (int q = 0; q < 8; q++) { x[q] = nx[q]; dv[q] = ndv[q]; yk[q] = nyk[q]; }
        if (e + 1 < ke) fetch(kl[e + 1]);
#pragma unroll
        for (int ks = 0; ks < 8; ks++) {
            const double2 a0 = x[ks >> 1], a1 = x[4 + (ks >> 1)];
            p0 += ((ks & 1) ? a0.y : a0.x) * yk[ks];
            p1 += ((ks & 1) ? a1.y : a1.x) * yk[ks];
        }
        // (k-step outermost: four independent accumulators back to back -- a chain of dependent FP64 MFMAs issues at about two
        // thirds of the rate; every accumulator still sums its k-steps in ascending order)
#pragma unroll
        for (int ks = 0; ks < 8; ks++)
#pragma unroll
            for (int ti = 0; ti < 2; ti++)
#pragma unroll
                for (int tj = 0; tj < 2; tj++) {
                    const double2 pi = x[ti * 4 + (ks >> 1)], pj = x[tj * 4 + (ks >> 1)];
                    const double av = -((ks & 1) ? pi.y : pi.x);
                    const double bv = ((ks & 1) ? pj.y : pj.x) * dv[ks];
                    acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[ti][tj], 0, 0, 0);
                }
    }
    p0 += __shfl_xor(p0, 16, 64); p0 += __shfl_xor(p0, 32, 64);
    p1 += __shfl_xor(p1, 16, 64); p1 += __shfl_xor(p1, 32, 64);
    sdot_out = (lane & 16) ? p1 : p0;   // lane r < 32 holds the sum of row r (lanes 32..63 mirror them)
}

// Diagonal tile of block column J: C_JJ, its L D L^T, y_J, D_J and W_J = (L_JJ^-T D_J^-1)^T (packed like a factor tile), with the
// elimination of k_chol_step4 (v_fmac_f64_dpp row_newbcast: one instruction per element of a rank-1 update instead of two
// v_readlane + one FMA: the first form of this kernel, 6 054 instructions; diagonal rows in lanes 0..31 and the rows that ride
// along in lanes 32..63 of ONE register array).  The rows that ride along with the diagonal tile are the rows of the IDENTITY: a row P of a
// panel comes out of the elimination as P L^-T D^-1, so the identity comes out as L_JJ^-T D_J^-1 -- W_J itself, for the price of the
// ride (496 FMACs) instead of a 32-column forward substitution against L_JJ in LDS (in-kernel stamps of ll_diag: 12-18 k cycles
// LDL^T + 5 k y_J + 13 k W_J per column; here ~10 k for all three).  The right-hand side rides along as one more column (y_J).
// Both tiles leave through LDS in 16-byte pieces of the packed order (ll_diag: 64 scattered 8-byte stores per lane).
DEVI void ll_diag2(const Batch& B, const WinDesc& d, WinCtrl& c, int w, int J, double* CT, double* WT) {
    const int* pb = B.tl_pan_begin + d.tl_step0;
    const int* klb = B.tl_kl_begin + d.tl_kb0;
    const int ent = pb[J] + J;  // column entry of (J,J)
    const int lane = threadIdx.x, r = lane & 31, hi = lane >> 5;
    const int l15 = lane & 15, l4 = lane >> 4;
    d4_t acc[2][2];
    double sdot = 0.0;
    ll_diag_accumulate(B, d, J, klb[ent], klb[ent + 1], acc, sdot);
    // C_JJ through LDS into one row per lane of the lower half of the wave (what lies above the diagonal is never used); the upper
    // half holds the rows of the identity
#pragma unroll
    for (int ti = 0; ti < 2; ti++)
#pragma unroll
        for (int tj = 0; tj < 2; tj++)
#pragma unroll
            for (int i = 0; i < 4; i++) CT[(16 * ti + l4 + 4 * i) * 34 + 16 * tj + l15] = acc[ti][tj][i];
    lds_barrier();
    double t[32];
#pragma unroll
    for (int q = 0; q < 32; q++) {
        const double cv = CT[r * 34 + q];
        t[q] = hi ? ((q == r) ? 1.0 : 0.0) : cv;
    }
    const size_t dk = (size_t)J * 32;
    const double rh = (B.vec + d.vec0)[dk + r] - __shfl(sdot, r, 64);   // b_J - sum_k L_Jk y_k
    double rr = hi ? 0.0 : rh, dout = 1.0, zout = 0.0;
    lds_barrier();                                    // every lane has its row: CT is free
    elim_tile<0>(t, rr, dout, zout, 4 * l15, 4 * (16 + l15), nullptr);
    {
        const bool badl = !hi && (dout == 0.0 || !isfinite(dout));
        if (__ballot(badl) != 0ull && lane == 0) c.chol_fail = 1;
    }
    if (!hi) {
        B.dvec[d.vec0 + dk + r] = dout;
        (B.yv + d.vec0)[dk + r] = zout;
    }
    // rows of L_JJ (unit lower, D on the diagonal, zeros above) into CT and the identity rows -- row r of L^-T D^-1 = column r of W_J^T --
    // into WT, one row per lane, no divergence: CT[r][c] = L[r][c], WT[r][c] = W^T[c][r]
    double* rowp = (hi ? WT : CT) + r * 34;
#pragma unroll
    for (int q = 0; q < 32; q++) rowp[q] = (hi || q < r) ? t[q] : ((q == r) ? dout : 0.0);
    lds_barrier();
    // packed order (ll_pk): piece (q, lane) = { M[row][c0], M[row][c0 + 4] }, row = 16 (q >> 2) + (lane & 15), c0 = 8 (q & 3) + (lane >> 4)
    double2* Ljj = reinterpret_cast<double2*>(B.Lf + d.S0 + ll_tile(d, J, J)) + lane;
    double2* Wd = reinterpret_cast<double2*>(B.winv + 1024 * (size_t)d.win) + lane;   // d.win: the batch-wide window index
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int row = 16 * (q >> 2) + l15, c0 = 8 * (q & 3) + l4;
        Ljj[64 * q] = make_double2(CT[row * 34 + c0], CT[row * 34 + c0 + 4]);
        Wd[64 * q] = make_double2(WT[c0 * 34 + row], WT[(c0 + 4) * 34 + row]);
    }
}

// (A fused variant -- the wave that owns tile (J+1,J) going on to factor diagonal tile J+1 -- was measured: no gain at 512
// windows)
__global__ void __launch_bounds__(64) k_chol_diag_ll2(Batch B, int J) {
    __shared__ double CT[32 * 34];
    __shared__ double WT[32 * 34];
    const int w = blockIdx.x;
    if (w >= B.n_win) return;
    const WinDesc& d = B.desc[w];
    WinCtrl& c = B.ctrl[w];
    if (!win_on(d, c)) return;
    if (J >= d.nb || J < d.nc) return;
    ll_diag2(B, d, c, w, J, CT, WT);
}

// panel tile (I,J): C_IJ = S_IJ - sum_k L_Ik D_k L_Jk^T, then L_IJ = C_IJ W_J.  The wave accumulates the TRANSPOSED tile
// (A operand = L_Jk D_k, B operand = -L_Ik: the same products in the same order as the row-major form, so the same bits): the
// accumulator registers of C^T are then exactly the B operand of L_IJ^T = W_J^T C_IJ^T, and that product's result registers are
// exactly the packed order of tile (I,J) -- nothing passes through LDS.
DEVI void ll_panel_init(const double* S, int n, int I, int J, int l15, int l4, d4_t (&acc)[2][2]) {
#pragma unroll
    for (int tj = 0; tj < 2; tj++)
#pragma unroll
        for (int ti = 0; ti < 2; ti++) {
            const double* C = S + ((size_t)I * 32 + 16 * ti + l15) * n + (size_t)J * 32 + 16 * tj + l4;
#pragma unroll
            for (int i = 0; i < 4; i++) acc[tj][ti][i] = C[4 * i];   // acc[tj][ti][i] = C[32 I + 16 ti + l15][32 J + 16 tj + 4 i + l4]
        }
}
DEVI void ll_panel_mfma(const double (&av)[2][8], const double2 (&xi)[8], d4_t (&acc)[2][2]) {
#pragma unroll
    for (int ks = 0; ks < 8; ks++)      // (k-step outermost: the four accumulators are independent of one another)
#pragma unroll
        for (int tj = 0; tj < 2; tj++)
#pragma unroll
            for (int ti = 0; ti < 2; ti++) {
                const double2 pi = xi[ti * 4 + (ks >> 1)];
                const double bv = -((ks & 1) ? pi.y : pi.x);
                acc[tj][ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[tj][ks], bv, acc[tj][ti], 0, 0, 0);
            }
}
// L_IJ^T = W_J^T C_IJ^T: A operand = the packed W tile, B operand = the accumulators; the result registers are the packed tile
DEVI void ll_panel_finish_keep(const double2 (&xw)[8], const d4_t (&acc)[2][2], double2* out, double2 (&xo)[8]);
DEVI void ll_panel_finish(const double2 (&xw)[8], const d4_t (&acc)[2][2], double2* out) {
    double2 xo[8];
    ll_panel_finish_keep(xw, acc, out, xo);
}

// the same, also handing the tile back as the operand pieces of a later product (k_chol_chain_panel)
DEVI void ll_panel_finish_keep(const double2 (&xw)[8], const d4_t (&acc)[2][2], double2* out, double2 (&xo)[8]) {
    d4_t o[2][2];
#pragma unroll
    for (int tk = 0; tk < 2; tk++)
#pragma unroll
        for (int ti = 0; ti < 2; ti++) o[tk][ti] = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 8; ks++)      // (k-step outermost: four independent accumulators)
#pragma unroll
        for (int tk = 0; tk < 2; tk++)
#pragma unroll
            for (int ti = 0; ti < 2; ti++) {
                const double2 pw = xw[tk * 4 + (ks >> 1)];
                const double av = (ks & 1) ? pw.y : pw.x;
                const double bv = acc[ks >> 2][ti][ks & 3];
                o[tk][ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, o[tk][ti], 0, 0, 0);
            }
#pragma unroll
    for (int tk = 0; tk < 2; tk++)
#pragma unroll
        for (int ti = 0; ti < 2; ti++) {
            // o[i] = L[16 ti + l15][16 tk + 4 i + l4]: k-steps 4 tk + i of row half ti
            xo[ti * 4 + 2 * tk] = make_double2(o[tk][ti][0], o[tk][ti][1]);
            xo[ti * 4 + 2 * tk + 1] = make_double2(o[tk][ti][2], o[tk][ti][3]);
            out[64 * (ti * 4 + 2 * tk)] = xo[ti * 4 + 2 * tk];
            out[64 * (ti * 4 + 2 * tk + 1)] = xo[ti * 4 + 2 * tk + 1];
        }
}

// One wave per tile.  (Measured and not kept, all within +-3 % of this form at 4096 windows: two tiles of the column per wave sharing
// L_Jk D_k and W_J -- a quarter less traffic; one 32-byte record per tile instead of the chain column row -> panel entry -> list
// bounds -> list; a register double buffer for the next product's tiles; 2, 3 or 4 waves per SIMD.  See DESIGN.md section 6.)
__global__ void __launch_bounds__(64) k_chol_panel_ll(Batch B, int J, int per_win) {
    // the waves of one window sit on one XCD (schur_map)
    int w, bx;
    if (!schur_map(B, per_win, w, bx)) return;
    const WinDesc& d = B.desc[w];
    if (!win_on(d, B.ctrl[w])) return;
    if (J >= d.nb || J < d.nc) return;
    const int* pb = B.tl_pan_begin + d.tl_step0;
    const int* pan = B.tl_pan + d.tl_pan0;
    const int npan = pb[J + 1] - pb[J];
    if (bx >= npan) return;
    const int I = pan[pb[J] + bx];
    const int* klb = B.tl_kl_begin + d.tl_kb0;
    const int ent = pb[J] + J + 1 + bx;
    const int lane = threadIdx.x, n = d.nS;
    const int l15 = lane & 15, l4 = lane >> 4;
    const double* S = B.S + d.S0;
    double* Lf = B.Lf + d.S0;
    d4_t acc[2][2];
    ll_panel_init(S, n, I, J, l15, l4, acc);
    const int* kl = B.tl_kl + d.tl_k0;
    const int kb = klb[ent], ke = klb[ent + 1];
    for (int e = kb; e < ke; e++) {
        const int k = kl[e];
        const double2* tj_ = reinterpret_cast<const double2*>(Lf + ll_tile(d, J, k)) + lane;
        const double2* ti_ = reinterpret_cast<const double2*>(Lf + ll_tile(d, I, k)) + lane;
        const double* sd = B.dvec + d.vec0 + (size_t)k * 32 + l4;
        double2 xj[8], xi[8];
        double dv[8];
#pragma unroll
        for (int q = 0; q < 8; q++) { xj[q] = tj_[64 * q]; xi[q] = ti_[64 * q]; dv[q] = sd[4 * q]; }
        double av[2][8];
#pragma unroll
        for (int tj = 0; tj < 2; tj++)
#pragma unroll
            for (int ks = 0; ks < 8; ks++) {
                const double2 pj = xj[tj * 4 + (ks >> 1)];
                av[tj][ks] = ((ks & 1) ? pj.y : pj.x) * dv[ks];
            }
        ll_panel_mfma(av, xi, acc);
    }
    const double2* tw = reinterpret_cast<const double2*>(B.winv + 1024 * (size_t)d.win) + lane;
    double2 xw[8];
#pragma unroll
    for (int q = 0; q < 8; q++) xw[q] = tw[64 * q];
    ll_panel_finish(xw, acc, reinterpret_cast<double2*>(Lf + ll_tile(d, I, J)) + lane);
}

// K_trsv: L^T x = y.  One 256-thread workgroup per window walks the block columns from the bottom: the
// column's nonzero tiles are gathered by the four waves (one tile per wave and trip), the 32x32 triangular solve runs in wave 0 on
// registers (column of L per lane, v_readlane broadcasts).
__global__ void __launch_bounds__(256) k_trsv(Batch B) {
    extern __shared__ double xs[];  // nS doubles + 8*32 partials + 32*33 diagonal tile
    const int w = blockIdx.x;
    const WinDesc& d = B.desc[w];
    if (!win_on(d, B.ctrl[w])) return;
    const int n = d.nS, t = threadIdx.x;
    double* part = xs + n;
    double* Lt = part + 256;
    const double* S = B.Lf + d.S0;
    const bool pk = B.l_packed;   // left-looking mode: the factor is stored tile by tile (ll_pk)
    double* vec = B.vec + d.vec0;
    const double* yv = B.yv + d.vec0;
    const int* pb = B.tl_pan_begin + d.tl_step0;
    const int* pan = B.tl_pan + d.tl_pan0;
    for (int q = t; q < n; q += 256) xs[q] = yv[q];
    for (int q = t; q < 256; q += 256) part[q] = 0.0;
    __syncthreads();
    // the column's panel tiles are dealt to the four waves; a lane takes 16 elements of a tile, so its loads are in flight
    // together and a column with m tiles costs ceil(m / 4) memory round trips, not m
    const int cc = t & 31, rl = t >> 5, wave = t >> 6, half = (t >> 5) & 1;
    const int lane = t & 63, l15 = lane & 15, l4 = lane >> 4;
    // the diagonal tile of the next block column is fetched while this one is solved; element u of a thread sits at (dr[u], dc[u])
    double nx[4];
    int dpos[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int q = t + 256 * u;   // row-major: element (q >> 5, q & 31); packed: piece q >> 1 of lane (q >> 1) & 63, half q & 1
        const int pr = 16 * (q >> 9) + ((q >> 1) & 15), pc = 8 * ((q >> 7) & 3) + 4 * (q & 1) + ((q >> 5) & 3);
        dpos[u] = pk ? pr * 33 + pc : (q >> 5) * 33 + (q & 31);
    }
    auto diag_fetch = [&](int k) {
        if (pk) {
            const double* tdk = S + ll_tile(d, k, k);
#pragma unroll
            for (int u = 0; u < 4; u++) nx[u] = tdk[t + 256 * u];
        } else {
            const size_t dk = (size_t)k * 32;
#pragma unroll
            for (int u = 0; u < 4; u++) { const int q = t + 256 * u; nx[u] = S[(dk + (q >> 5)) * n + dk + (q & 31)]; }
        }
    };
    diag_fetch(d.nb - 1);
#ifdef VBA_STAMPS
    unsigned long long st_[8];
#define TSTAMP(i) { if (k == 10) st_[i] = __builtin_amdgcn_s_memtime(); }
#else
#define TSTAMP(i)
#endif
    for (int k = d.nb - 1; k >= 0; k--) {
        const size_t dk = (size_t)k * 32;
        const int i0 = pb[k], m = pb[k + 1] - i0;
        TSTAMP(0)
        if (pk) {
            // a wave per tile: 8 coalesced 16-byte loads per lane; the lane's 16 elements are rows l15 and 16 + l15 of the
            // columns 4 ks + l4 -- eight column sums per lane, reduced over the 16 lanes of a DPP row at the end
            double cs[8];
#pragma unroll
            for (int q = 0; q < 8; q++) cs[q] = 0.0;
            for (int i = wave; i < m; i += 4) {
                const int I = pan[i0 + i];
                const double2* tp = reinterpret_cast<const double2*>(S + ll_tile(d, I, k)) + lane;
                double2 lv[8];
#pragma unroll
                for (int q = 0; q < 8; q++) lv[q] = tp[64 * q];
                const double x0 = xs[I * 32 + l15], x1 = xs[I * 32 + 16 + l15];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    cs[2 * j] += lv[j].x * x0;
                    cs[2 * j + 1] += lv[j].y * x0;
                    cs[2 * j] += lv[4 + j].x * x1;
                    cs[2 * j + 1] += lv[4 + j].y * x1;
                }
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const double v = row16_sum(cs[q]);
                if (l15 == 0) part[wave * 32 + 4 * q + l4] = v;
            }
        } else {
            double s = 0.0;
            for (int i = wave; i < m; i += 4) {
                const int I = pan[i0 + i];
                const int r0 = I * 32 + half * 16;
                double lv[16];
#pragma unroll
                for (int rr = 0; rr < 16; rr++) lv[rr] = S[(size_t)(r0 + rr) * n + dk + cc];
#pragma unroll
                for (int rr = 0; rr < 16; rr++) s += lv[rr] * xs[r0 + rr];
            }
            part[rl * 32 + cc] = s;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) Lt[dpos[u]] = nx[u];
        TSTAMP(1)
        if (k > 0) diag_fetch(k - 1);
        __syncthreads();
        TSTAMP(2)
        if (t < 64) {
            const int c = t & 31;
            double v;
            double col[32];
#pragma unroll
            for (int j = 0; j < 32; j++) col[j] = Lt[j * 33 + c];  // column c of L_kk (rows j >= c are the factor)
            double dgc = 1.0;
#pragma unroll
            for (int j = 0; j < 32; j++) dgc = (j == c) ? col[j] : dgc;  // d_c sits on the diagonal of the tile
            v = xs[dk + c] / dgc;                                         // D^-1 z
#pragma unroll
            for (int q = 0; q < 8; q++) v -= part[q * 32 + c];
#ifdef VBA_STAMPS
            asm volatile("" :: "v"(v));
#endif
            TSTAMP(3)
#pragma unroll
            for (int j = 31; j >= 0; j--) {                               // unit-diagonal L_kk^T x_k = v
                const double xj = rl64(v, j);
                v = (c < j) ? v - col[j] * xj : v;
            }
            if (t < 32) xs[dk + c] = v;
            TSTAMP(4)
        }
        __syncthreads();
        TSTAMP(5)
#ifdef VBA_STAMPS
        if (k == 10 && t == 0) for (int i = 0; i < 6; i++) B.dbg[16 + i] = (double)st_[i];
#endif
    }
#undef TSTAMP
    for (int q = t; q < n; q += 256) vec[q] = xs[q];
}

// The back-substitution for the row-major factor of the few-window kernels, where its latency counts (one window: 8-10 calls per
// solve, 23 block columns each), as a two-stage pipeline (8 waves).  In k_trsv a block column is a chain  products of ALL its tiles
// with x -> barrier -> 32-step triangular solve -> barrier  (2.3 us per column at C3 size even with the tile lists in LDS and the
// tiles prefetched: round 3's k_trsv_w, 52 us per call), although only ONE of those tiles, (k+1, k), needs the x_{k+1} that the
// previous column has just produced.  Here
//   * wave 0 only solves: column K in 32 steps of  x_j = readlane(acc, j); acc -= reg[j] * x_j.  Lanes 0..31 hold column c of the
//     (strictly lower) diagonal tile in reg[] and the running right-hand side in acc; lanes 32..63 hold column c of tile (K, K-1)
//     and accumulate -sum_j L[32K + j][32(K-1) + c] x_j with the SAME instruction: the one product that needs x_K is finished when x_K
//     is, and seeds column K - 1;
//   * waves 1..7 work one column ahead: the products of column K - 1 with the tiles I >= K + 1 (x_I is final), summed to one partial
//     row per wave; they also bring the diagonal tile and tile (K-1, K-2) of the next solve into LDS (zeros above the diagonal, so the
//     solve needs no mask) and divide z by d (the solve starts from D^-1 z); their tiles are requested one phase earlier still
//     (registers: eight 16-byte loads per tile from one wave-uniform base + 32-bit lane offsets), so no load waits inside a phase.
// One LDS-only barrier per column; the chain per column is wave 0's ~10 LDS reads + 32 x 3 instructions.  Fixed summation order.
// Measured at C3 size: 37 -> 30 us per call (k_trsv_w: 52).
#define TRSV_P_PF 4
#define TRSV_P_DW 7
__global__ void __launch_bounds__(512) k_trsv_p(Batch B) {
    extern __shared__ double xs[];  // nS doubles | 2 x 7 x 32 partial rows | 2 x 32 x 65 solve blocks | 32 seeds | tile lists (ints)
    const int w = blockIdx.x;
    const WinDesc& d = B.desc[w];
    if (!win_on(d, B.ctrl[w])) return;
    const int n = d.nS, nb = d.nb, t = threadIdx.x;
    double* part = xs + n;
    double* R = part + 2 * TRSV_P_DW * 32;
    double* seed = R + 2 * 32 * 65;
    int* lpb = reinterpret_cast<int*>(seed + 32);   // [nb + 1] column starts, then the tile rows
    int* lpan = lpb + nb + 1;
    const double* S = B.Lf + d.S0;
    double* vec = B.vec + d.vec0;
    const double* yv = B.yv + d.vec0;
    {
        const int* pb = B.tl_pan_begin + d.tl_step0;
        const int* pan = B.tl_pan + d.tl_pan0;
        const int np = pb[nb];
        for (int q = t; q <= nb; q += 512) lpb[q] = pb[q];
        for (int q = t; q < np; q += 512) lpan[q] = pan[q];
    }
    for (int q = t; q < n; q += 512) xs[q] = yv[q];
    for (int q = t; q < 2 * TRSV_P_DW * 32 + 2 * 32 * 65 + 32; q += 512) part[q] = 0.0;
    __syncthreads();
    const int wave = t >> 6, lane = t & 63;
    if (wave == 0) {
        // ---------------- the solving wave ----------------
        const int c = lane & 31, hiL = lane >> 5;
        lds_barrier();                               // (the loaders' prologue)
        for (int K = nb - 1; K >= 0; K--) {
            const double* Rk = R + (K & 1) * (32 * 65);
            const double* pk = part + (K & 1) * (TRSV_P_DW * 32);
            double reg[32];
#pragma unroll
            for (int j = 0; j < 32; j++) reg[j] = Rk[j * 65 + lane];
            double acc = 0.0;
            if (!hiL) {
                acc = xs[K * 32 + c] + seed[c];      // D^-1 z (divided by the loaders) - the product with tile (K+1, K)
#pragma unroll
                for (int q = 0; q < TRSV_P_DW; q++) acc -= pk[q * 32 + c];
            }
#pragma unroll
            for (int j = 31; j >= 0; j--) {          // unit-diagonal L_KK^T x_K = acc (reg is zero on and above the diagonal)
                const double xj = rl64(acc, j);
                acc = __builtin_fma(-reg[j], xj, acc);
            }
            if (hiL) seed[c] = acc; else xs[K * 32 + c] = acc;
            lds_barrier();
        }
    } else {
        // ---------------- the seven waves that work ahead ----------------
        const int dw = wave - 1, t2 = t - 64;        // 0..447
        const int cp = lane & 15, rg = lane >> 4;   // column pair, group of eight rows of a tile
        const unsigned rowb = (unsigned)n * 8u;      // row pitch in bytes
        const unsigned lane_off = (unsigned)(rg * 8) * rowb + (unsigned)cp * 16u;
        double2 nlv[TRSV_P_PF][8];
        int nI[TRSV_P_PF];
        double rst[5];
        auto tile_load = [&](int I, size_t dk, double2 (&lv)[8]) {
            const char* tb = reinterpret_cast<const char*>(S + (size_t)I * 32 * n + dk);   // wave-uniform
#pragma unroll
            for (int rr = 0; rr < 8; rr++) lv[rr] = *reinterpret_cast<const double2*>(tb + (lane_off + (unsigned)rr * rowb));
        };
        auto tile_dot = [&](int I, const double2 (&lv)[8], double& s0, double& s1) {
            const double* x = xs + I * 32 + rg * 8;
#pragma unroll
            for (int rr = 0; rr < 8; rr++) { s0 += lv[rr].x * x[rr]; s1 += lv[rr].y * x[rr]; }
        };
        // the tiles (I, col) with I >= col + 2 (tile (col + 1, col), if the column has it, belongs to the solving wave)
        auto dot_range = [&](int col, int& i0, int& m) {
            i0 = lpb[col]; m = lpb[col + 1] - i0;
            if (m > 0 && lpan[i0] == col + 1) { i0++; m--; }
        };
        auto prefetch_dots = [&](int col) {
#pragma unroll
            for (int s = 0; s < TRSV_P_PF; s++) nI[s] = -1;
            if (col < 0) return;
            int i0, m;
            dot_range(col, i0, m);
#pragma unroll
            for (int s = 0; s < TRSV_P_PF; s++) {
                const int i = dw + TRSV_P_DW * s;
                nI[s] = (i < m) ? lpan[i0 + i] : -1;      // (wave-uniform)
                if (nI[s] >= 0) tile_load(nI[s], (size_t)col * 32, nlv[s]);
            }
        };
        // solve block of column K: element e = 1024 tile + 32 j + c; tile 0 = (K, K), tile 1 = (K, K - 1) (zeros if it is not in the
        // factor).  What a lane fetches and where it goes do not depend on K: byte offsets from a wave-uniform base, computed once.
        unsigned roff[5];
        int ridx[5];
        bool rkeep[5];
#pragma unroll
        for (int u = 0; u < 5; u++) {
            const int e = t2 + 448 * u;
            const int tile = (e >> 10) & 1, j = (e >> 5) & 31, cc = e & 31;
            roff[u] = (unsigned)j * rowb + (unsigned)(cc + 32 - 32 * tile) * 8u;   // from 32 columns left of the diagonal tile
            ridx[u] = (e < 2048) ? j * 65 + 32 * tile + cc : -1;
            rkeep[u] = tile == 1 || cc < j;
        }
        const unsigned doff = (unsigned)(lane & 31) * (rowb + 8u) + 256u;           // the diagonal (wave 7 divides z by it)
        double rdg = 1.0;
        auto prefetch_R = [&](int K) {
            if (K < 0) return;
            const bool has_nxt = K >= 1 && lpb[K] > lpb[K - 1] && lpan[lpb[K - 1]] == K;
            const char* base = reinterpret_cast<const char*>(S + (size_t)K * 32 * n + (size_t)K * 32) - 256;   // wave-uniform
#pragma unroll
            for (int u = 0; u < 5; u++) {
                const bool want = ridx[u] >= 0 && (has_nxt || ridx[u] % 65 < 32);   // (ridx % 65 = 32 tile + c)
                rst[u] = want ? *reinterpret_cast<const double*>(base + roff[u]) : 0.0;
            }
            if (dw == TRSV_P_DW - 1) rdg = *reinterpret_cast<const double*>(base + doff);
        };
        auto stage_R = [&](int K) {
            if (K < 0) return;
            double* Rk = R + (K & 1) * (32 * 65);
#pragma unroll
            for (int u = 0; u < 5; u++)
                if (ridx[u] >= 0) Rk[ridx[u]] = rkeep[u] ? rst[u] : 0.0;
            if (dw == TRSV_P_DW - 1 && lane < 32) xs[K * 32 + lane] = xs[K * 32 + lane] / rdg;   // D^-1 z
        };
        // prologue: the block of the last column; nothing to multiply yet
        prefetch_R(nb - 1);
        stage_R(nb - 1);
        prefetch_dots(nb - 2);
        prefetch_R(nb - 2);
        lds_barrier();
        for (int K = nb - 1; K >= 0; K--) {
            const int col = K - 1;                   // the column whose products are formed in this phase
            if (col >= 0) {
                int i0, m;
                dot_range(col, i0, m);
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int q = 0; q < TRSV_P_PF; q++)  // tile i of the column goes to wave 1 + i % 7
                    if (nI[q] >= 0) tile_dot(nI[q], nlv[q], s0, s1);
                for (int i = dw + TRSV_P_DW * TRSV_P_PF; i < m; i += TRSV_P_DW) {
                    const int I = lpan[i0 + i];
                    double2 lv[8];
                    tile_load(I, (size_t)col * 32, lv);
                    tile_dot(I, lv, s0, s1);
                }
                s0 += __shfl_xor(s0, 16, 64); s1 += __shfl_xor(s1, 16, 64);
                s0 += __shfl_xor(s0, 32, 64); s1 += __shfl_xor(s1, 32, 64);
                if (rg == 0) *reinterpret_cast<double2*>(part + (col & 1) * (TRSV_P_DW * 32) + dw * 32 + 2 * cp) = make_double2(s0, s1);
                stage_R(col);
                prefetch_dots(col - 1);
                prefetch_R(col - 1);
            }
            lds_barrier();
        }
    }
    __syncthreads();
    for (int q = t; q < n; q += 512) vec[q] = xs[q];
}

// ------------------------------------------------------------------------------------------------
// K_update: landmark back-substitution x_l = Dinv (b_l - W^T x_p) (block_solver.hpp:461-481) fused with the
// vertex retractions (SparseOptimizer::update -> oplusImpl): blocks [0,nblk_pt) landmarks, then keyframes.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_update(Batch B, int nblk_pt) {
    const int w = blockIdx.y;
    const WinDesc& d = B.desc[w];
    const WinCtrl& c = B.ctrl[w];
    if (!c.active || c.chol_fail) return;
    const double* x = B.vec + d.vec0;
    const int P = d.pdim;
    if ((int)blockIdx.x < nblk_pt) {
        const int p = blockIdx.x * 64 + threadIdx.x;
        if (p >= d.n_pt) return;
        const size_t gp = d.pt0 + p;
        const double* slots = B.slot + VBA_SLOT * (size_t)(d.obs0 + d.pt0);
        const double* sr = slots + VBA_SLOT * (size_t)(d.n_obs + B.pt_perm[gp]);
        const double sD = sr[7];
        if (!(sD > 0.0)) return;  // landmark outside the active set
        double cl = sr[6];        // beta - sum U . x_p
        const int rf = B.pt_ref[gp];
        if (rf < d.n_free)
#pragma unroll
            for (int i = 0; i < 6; i++) cl -= sr[i] * x[vpos(d, rf, i)];
        const int* ob = B.pt_obs_begin + d.pt0 + d.win;
        // two edges per trip, their indices fetched one trip ahead: per landmark ~n/2 + 1 dependent round trips instead of 2 n
        const int o0 = ob[p], o1 = ob[p + 1];
        int kfn[2] = {0, 0}, spn[2] = {0, 0};
#pragma unroll
        for (int u = 0; u < 2; u++)
            if (o0 + u < o1) { kfn[u] = B.obs_kf[d.obs0 + o0 + u]; spn[u] = B.slot_perm[d.obs0 + o0 + u]; }
        for (int o = o0; o < o1; o += 2) {
            const int kf0 = kfn[0], kf1 = kfn[1], sp0 = spn[0], sp1 = spn[1];
            const bool two = o + 1 < o1;
#pragma unroll
            for (int u = 0; u < 2; u++)
                if (o + 2 + u < o1) { kfn[u] = B.obs_kf[d.obs0 + o + 2 + u]; spn[u] = B.slot_perm[d.obs0 + o + 2 + u]; }
            const bool on0 = kf0 < d.n_free, on1 = two && kf1 < d.n_free;
            const double* sl0 = slots + VBA_SLOT * (size_t)sp0;
            const double* sl1 = slots + VBA_SLOT * (size_t)(on1 ? sp1 : sp0);
            double u0[6], u1[6], x0[6], x1[6];
#pragma unroll
            for (int i = 0; i < 6; i++) {
                u0[i] = sl0[i]; u1[i] = sl1[i];
                x0[i] = x[vpos(d, on0 ? kf0 : 0, i)]; x1[i] = x[vpos(d, on1 ? kf1 : 0, i)];
            }
            if (on0) {
#pragma unroll
                for (int i = 0; i < 6; i++) cl -= u0[i] * x0[i];
            }
            if (on1) {
#pragma unroll
                for (int i = 0; i < 6; i++) cl -= u1[i] * x1[i];
            }
        }
        double rho = B.pt[3 * gp] + sD * cl;
        if (rho < 1e-6) rho = 1e-6;  // VertexIDP::oplusImpl, g2otypes.h:50-55
        B.pt[3 * gp] = rho;
    } else {
        const int a = (blockIdx.x - nblk_pt) * 64 + threadIdx.x;
        if (a >= d.n_free) return;
        const int* va = B.var_act + d.vec0;
        double dx[15];
        for (int i = 0; i < P; i++) dx[i] = x[vpos(d, a, i)];
        const size_t gk = d.kf0 + a;
        if (va[vpos(d, a, 0)]) {  // NavState::IncSmallPR, NavState.cpp:63-70
            double* T = B.pose + 7 * gk;
            T[0] += dx[0]; T[1] += dx[1]; T[2] += dx[2];
            double dq[4], qn[4];
            so3exp(dx + 3, dq);
            so3mul(T + 3, dq, qn);
            T[3] = qn[0]; T[4] = qn[1]; T[5] = qn[2]; T[6] = qn[3];
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

// ------------------------------------------------------------------------------------------------
// K_classify: outlier pass between the stages (src/Optimizer.cpp:475-490): chi2 from the stored error,
// depth and rho at the current estimate.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_classify(Batch B) {
    const int w = blockIdx.y;
    const WinDesc& d = B.desc[w];
    const WinCtrl& c = B.ctrl[w];
    if (!c.active) return;  // bDoMore == false (latched by k_stage_clear(1)) or aborted on entry
    const int o = blockIdx.x * 64 + threadIdx.x;
    if (o >= d.n_obs) return;
    const size_t go = d.obs0 + o;
    double s, depth;
    if (d.variant == 2) {   // inverse depth: evaluated here (idp_edge_eval); the value is kept for the edges this pass takes out --
        idp_edge_eval(B, d, go, s, depth);   // e->chi2() of a level-1 edge stays what it was when the edge left the optimisation
        B.chi2_e[go] = s;
    } else { s = B.chi2_e[go]; depth = B.depth_e[go]; }
    bool bad = (s > d.chi2_th) || !(depth > d.depth_min);
    if (d.variant == 2 && B.pt[3 * (size_t)(d.pt0 + B.obs_pt[go])] < d.rho_min) bad = true;
    if (bad) B.lvl[go] = 1;
}

// K_final: erase list + chi2 of the level-0 edges at the final estimates (src/Optimizer.cpp:496-517)
__global__ void __launch_bounds__(64) k_final_edges(Batch B) {
    __shared__ double sm[64];
    const int w = blockIdx.y;
    const WinDesc& d = B.desc[w];
    const int o = blockIdx.x * 64 + threadIdx.x;
    if ((int)blockIdx.x * 64 >= d.n_obs) return;
    // a window stopped at the very first terminate() of its first optimize() has evaluated nothing: e->chi2() then reads an _error
    // no computeError() ever wrote (uninitialised in g2o; zero here and in the oracle) -- the launch schedule has linearised once
    // before that poll, but those values do not exist for the protocol
    const bool never = B.ctrl[w].n_trace == 0;
    double chi = 0.0, cnt = 0.0;
    if (o < d.n_obs) {
        const size_t go = d.obs0 + o;
        // e->chi2() at the final estimates (what the last linearisation computed); for an edge the outlier pass took out: its value
        // at that pass (k_classify); the depth is evaluated at the final estimates for every edge (isDepthPositive())
        double stored, depth;
        if (d.variant == 2) {
            idp_edge_eval(B, d, go, stored, depth);
            if (B.lvl[go]) stored = B.chi2_e[go];
        } else { stored = B.chi2_e[go]; depth = B.depth_e[go]; }
        const double s = never ? 0.0 : stored;
        bool bad = (s > d.chi2_th) || !(depth > d.depth_min);
        if (d.variant == 2 && (B.pt[3 * (size_t)(d.pt0 + B.obs_pt[go])] < d.rho_min || B.lvl[go])) bad = true;
        if (d.protocol == 1) bad = false;  // global BA classifies nothing
        B.out_outlier[go] = bad ? 1 : 0;
        B.out_chi2[go] = s;
        if (!B.lvl[go]) chi = B.chi2_f ? B.chi2_f[go] : stored;
        cnt = bad ? 1.0 : 0.0;
    }
    const double tc = block_sum<64>(chi, sm);
    const double tn = block_sum<64>(cnt, sm);
    if (threadIdx.x == 0) {
        B.part[d.part0 + 2 * blockIdx.x] = tc;
        B.part[d.part0 + 2 * blockIdx.x + 1] = tn;
    }
}

__global__ void __launch_bounds__(64) k_final_sum(Batch B) {
    __shared__ double sm[64];
    const int w = blockIdx.x;
    const WinDesc& d = B.desc[w];
    WinCtrl& c = B.ctrl[w];
    const int t = threadIdx.x;
    const int nblk = (d.n_obs + 63) / 64;
    double s = 0, nn = 0, sp = 0, sbias = 0;
    for (int k = t; k < nblk; k += 64) { s += B.part[d.part0 + 2 * k]; nn += B.part[d.part0 + 2 * k + 1]; }
    for (int k = t; k < d.n_imu; k += 64) {
        const int i = B.imu_i[d.imu0 + k], j = B.imu_j[d.imu0 + k];
        if (i < d.n_free || j < d.n_free) { sp += B.imu_chi[4 * (size_t)(d.imu0 + k) + 2]; sbias += B.imu_chi[4 * (size_t)(d.imu0 + k) + 3]; }
    }
    s = block_sum<64>(s, sm);
    nn = block_sum<64>(nn, sm);
    sp = block_sum<64>(sp, sm);
    sbias = block_sum<64>(sbias, sm);
    if (t == 0) {
        if (c.status == 2) { c.n_outliers = 0; return; }
        c.chi2_vis = s; c.chi2_prv = sp; c.chi2_bias = sbias; c.n_outliers = (int)(nn + 0.5);
    }
}
