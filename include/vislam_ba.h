/*
 * vislam_ba.h -- C ABI of the MI355X local bundle-adjustment backend.
 *
 * Drop-in boundary.  The reference (mc275/MC_SLAM) has no FFI layer; the boundary
 * is the point where its host code hands a freshly built factor graph to g2o:
 *
 *     optimizer.initializeOptimization(); optimizer.optimize(5); ... optimizer.optimize(10);
 *       src/Optimizer.cpp:458-493   (Optimizer::LocalBAPRVIDP,            variant 2)
 *       src/Optimizer.cpp:1259-1317 (Optimizer::LocalBundleAdjustmentNavStatePRV, variant 1)
 *       src/Optimizer.cpp:4093-4143 (Optimizer::LocalBundleAdjustment,    variant 0)
 *
 * Everything g2o does between those lines (active-set construction, residuals + analytic Jacobians,
 * Huber weighting, H/b assembly, Schur complement, reduced solve, back-substitution, manifold update,
 * GN / LM control flow, the two-stage outlier protocol) happens behind vba_solve().  The host keeps graph
 * extraction (src/Optimizer.cpp:49-451) and write-back (:496-623).
 *
 * Plain C: flat caller-owned arrays, f64 + i32, no C++ / torch types across the line.
 */
#ifndef VISLAM_BA_H
#define VISLAM_BA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VBA_VARIANT_SE3_XYZ 0 /* VertexSE3Expmap + VertexSBAPointXYZ + EdgeSE3ProjectXYZ (types_six_dof_expmap.h:80) */
#define VBA_VARIANT_PRV_XYZ 1 /* VertexNavStatePR/V/Bias + XYZ + EdgeNavStatePRPointXYZ (g2otypes.h:255)          */
#define VBA_VARIANT_PRV_IDP 2 /* VertexNavStatePR/V/Bias + VertexIDP + EdgePRIDP (g2otypes.h:22,65)                */

#define VBA_ALGO_GN 0 /* OptimizationAlgorithmGaussNewton with the |dchi2|<1e-3 stop (gauss_newton.cpp:97) */
#define VBA_ALGO_LM 1 /* OptimizationAlgorithmLevenberg, g2o lambda/rho schedule (levenberg.cpp:61-164)    */

#define VBA_PROTO_LOCAL 0  /* optimize(its_stage1); outlier pass; optimize(its_stage2): the LocalBA functions */
#define VBA_PROTO_SINGLE 1 /* one optimize(its_stage1), no outlier pass: BundleAdjustment / GlobalBundleAdjustmentNavStatePRV
                            * (src/Optimizer.cpp:3377-3607, 629-933) */

#define VBA_SOLVER_LDLT 0 /* dense LDL^T on 32x32 tiles of the reduced system: what LinearSolverEigen does (linear_solver_eigen.h:94-124) */
#define VBA_SOLVER_PCG 1  /* block-Jacobi preconditioned conjugate gradients on the reduced system (not in the reference: BASELINE
                           * north_star / configs[3] "Schur + PCG"); tolerance 1e-10, so the outer iterations follow the direct path */

#define VBA_IMU_MEAS_STRIDE 61 /* dt, dP(3), dV(3), dR(9 row-major), JPg, JPa, JVg, JVa, JRg (9 each, row-major) */
#define VBA_TRACE_MAX 64

/* status values of vba_result.status */
#define VBA_OK 0
#define VBA_ABORTED_AFTER_STAGE1 1 /* stop flag seen after optimize(5): src/Optimizer.cpp:464-470 */
#define VBA_ABORTED_BEFORE 2       /* stop flag set on entry: src/Optimizer.cpp:453-455, nothing touched */
#define VBA_SOLVER_FAILED (-2)     /* reduced system not positive definite (linear_solver_eigen.h:105-111 -> Fail) */

typedef struct vba_problem {
    int32_t variant;          /* VBA_VARIANT_* */
    int32_t n_kf, n_kf_free;  /* free keyframes first (hessian order = caller order), fixed after */
    int32_t n_pt, n_obs, n_imu;
    /* keyframe state, updated IN PLACE for the free entries.
     * variant 0: T_cw as SE3Quat  (tx ty tz qx qy qz qw)          se3quat.h:40-47
     * variant 1,2: NavState P, R  (px py pz qx qy qz qw) = T_wb   src/IMU/NavState.h:124-138 */
    double *kf_pose;          /* [n_kf][7] */
    double *kf_vel;           /* [n_kf][3]  (variants 1,2; only KFs touched by IMU edges are read) */
    double *kf_bias;          /* [n_kf][12] bg(3) ba(3) dbg(3) dba(3); only dbg,dba are optimised */
    /* landmarks, updated IN PLACE.  variant 0,1: world xyz.  variant 2: rho, xbar, ybar (rho updated;
     * xbar,ybar = normalised ref-KF pixel, src/Optimizer.cpp:382-385) */
    double *pt;               /* [n_pt][3] */
    const int32_t *pt_ref_kf; /* [n_pt] variant 2: reference keyframe index (may be a fixed KF) */
    const int32_t *pt_obs_begin; /* [n_pt+1] CSR: observations of point p are [begin[p], begin[p+1]) */
    const int32_t *obs_kf;    /* [n_obs] observing keyframe (variant 2: never the reference KF, :395-398) */
    const double *obs_uv;     /* [n_obs][2] undistorted pixel (kpUn.pt) */
    const double *obs_w;      /* [n_obs] invSigma2 of the keypoint octave (information = w * I2) */
    double K[4];              /* fx fy cx cy */
    double T_cb[7];           /* camera<-body extrinsic, t_cb(3) q_cb(4) (variants 1,2; ConfigParam::GetEigT_cb) */
    double g_w[3];            /* gravity in world (variants 1,2) */
    const int32_t *imu_kf_i;  /* [n_imu] keyframe i (previous) */
    const int32_t *imu_kf_j;  /* [n_imu] keyframe j (owner of the preintegrator, pKF1) */
    const double *imu_meas;   /* [n_imu][VBA_IMU_MEAS_STRIDE] */
    const double *imu_info_prv; /* [n_imu][81] row-major information of EdgeNavStatePRV in P,phi,V order
                                 * (= inverse of the V/phi-swapped covariance, src/Optimizer.cpp:273-280) */
    double inv_bg_rw2, inv_ba_rw2; /* 1/IMUData::getGyrBiasRW2(), 1/getAccBiasRW2(); bias info = diag/dt (:244-249,302) */
    double huber_vis, huber_prv, huber_bias; /* Huber deltas (float-rounded sqrt(5.991), sqrt(2166.6), sqrt(1681.2)) */
    int32_t algo;             /* VBA_ALGO_* */
    int32_t its_stage1, its_stage2; /* 5, 10 */
    double chi2_th;           /* 5.991 */
    double depth_min;         /* isDepthPositive threshold: 0.01 (EdgePRIDP, g2otypes.h:122-127) or 0.0 */
    double rho_min;           /* 2e-6 (variant 2 outlier gate, src/Optimizer.cpp:484) */
    /* --- global bundle adjustment; all zero / NULL = the local-BA protocol above --- */
    int32_t protocol;         /* VBA_PROTO_* */
    int32_t robust;           /* VBA_PROTO_SINGLE only: bRobust -- Huber on every edge (1) or none at all (0) */
    const uint8_t *kf_fix;    /* NULL or [n_kf]: per-vertex setFixed() of the keyframes listed as free: bit0 PR, bit1 V,
                               * bit2 Bias (GlobalBundleAdjustmentNavStatePRV fixes PR and Bias of keyframe 0 but not
                               * its V: src/Optimizer.cpp:667-685) */
    int32_t solver;           /* VBA_SOLVER_*: how the reduced system is solved (0 = LDL^T, the reference's choice) */
} vba_problem;

typedef struct vba_result {
    double chi2_vis;   /* sum e'We over level-0 vision edges at the final estimates (N3 in SURVEY 8a) */
    double chi2_prv;   /* same for EdgeNavStatePRV */
    double chi2_bias;  /* same for EdgeNavStateBias */
    int32_t its_done[2];  /* outer iterations executed by optimize(5) / optimize(10) (cjIterations) */
    int32_t n_outliers;   /* number of set entries of obs_outlier */
    int32_t status;       /* VBA_OK / VBA_ABORTED_* / VBA_SOLVER_FAILED */
    uint8_t *obs_outlier; /* [n_obs] caller-allocated or NULL: 1 = host must erase (src/Optimizer.cpp:509-514) */
    double *obs_chi2;     /* [n_obs] caller-allocated or NULL: e->chi2() as the reference's erase loop reads it */
    int32_t n_trace;      /* entries of chi2_trace */
    double chi2_trace[VBA_TRACE_MAX]; /* activeRobustChi2 after every accepted/terminating evaluation (diagnostic) */
    double lambda_final;  /* LM only */
    int32_t lin_iterations; /* VBA_SOLVER_PCG: conjugate-gradient iterations summed over all solves of the window (0 for LDL^T) */
} vba_result;

/* Per-kernel-class device time of the last vba_batch_run, measured with HIP events on the backend's
 * own stream (only filled when profiling is enabled with vba_set_profile). */
#define VBA_PROF_LINEARIZE 0
#define VBA_PROF_CONTROL 1
#define VBA_PROF_SCHUR 2
#define VBA_PROF_FACTOR 3
#define VBA_PROF_TRSV 4
#define VBA_PROF_UPDATE 5
#define VBA_PROF_MISC 6
#define VBA_PROF_N 8
typedef struct vba_profile {
    double ms[VBA_PROF_N];       /* summed device time per class */
    int64_t launches[VBA_PROF_N];
    double bytes[VBA_PROF_N];    /* algorithmic bytes moved per class (SURVEY 8d accounting) */
    double total_ms;             /* first launch -> last launch of the run */
    double factor_flops;         /* FP64 flop the factorisation class executed on MFMA: 2*32^3 per tile product of the
                                    symbolic tile lists, per solve (structurally zero tiles are never touched) */
    int64_t kernel_launches;     /* kernel launches the last vba_batch_run / vba_solve enqueued (filled with or without profiling) */
} vba_profile;

/* One handle per host thread / GPU; owns device buffers and a stream.  Errors: nonzero return, message
 * via vba_last_error.  Never throws, never aborts. */
int vba_create(int device, void **handle);
int vba_destroy(void *handle);
const char *vba_last_error(void *handle);

/* Replaces optimizer.initializeOptimization(); optimize(its_stage1); <outlier pass>; optimize(its_stage2)
 * (src/Optimizer.cpp:458-493 / 4093-4143).  stop_flag (may be NULL) plays g2o's forceStopFlag
 * (sparse_optimizer.h:188): polled before the solve, between outer iterations and between the stages. */
int vba_solve(void *handle, vba_problem *inout, vba_result *out, const volatile int *stop_flag);

/* Batched / device-resident form: independent windows solved in lock-step on one GPU.
 * upload = H2D + structure build (g2o buildStructure, block_solver.hpp:143-295);
 * run    = the whole two-stage solve from the uploaded initial state, everything HBM-resident;
 * download = D2H of states and per-edge results into the caller's arrays.
 * run may be repeated (each run restarts from the uploaded state). */
int vba_batch_upload(void *handle, int32_t n_windows, vba_problem *const *problems);
int vba_batch_run(void *handle, const volatile int *stop_flag);
int vba_batch_download(void *handle, int32_t n_windows, vba_problem *const *inout, vba_result *const *out);

/* Streamed form of the same: n FRESH windows in, n solved windows out (states updated in place, results filled), for callers
 * that have many independent windows at once (map-server replays, multi-session back-ends).  Equivalent to
 * upload + run + download of the whole batch, but the call cuts the batch into chunks and keeps several in flight, so the
 * host packing, the PCIe transfers and the structure build of one chunk overlap the solve of another. */
int vba_batch_solve(void *handle, int32_t n_windows, vba_problem *const *inout, vba_result *const *out,
                    const volatile int *stop_flag);

/* The same three entry points for a caller whose flag is a C++ `bool` (one byte): the reference hands `bool* pbStopFlag` =
 * &LocalMapping::mbAbortBA down to g2o (include/Optimizer.h:22-24; written by the Tracking thread, src/LocalMapping.cpp:1769-1772).
 * The flag is read at its own width, so `Optimizer::LocalBAPRVIDP` passes its argument straight through -- no mirror word, no
 * helper thread.  (sizeof(bool) == 1 on every ABI the library is built for; the facade static_asserts it.) */
int vba_solve_b(void *handle, vba_problem *inout, vba_result *out, const volatile unsigned char *stop_flag);
int vba_batch_run_b(void *handle, const volatile unsigned char *stop_flag);
int vba_batch_solve_b(void *handle, int32_t n_windows, vba_problem *const *inout, vba_result *const *out,
                      const volatile unsigned char *stop_flag);

/* On-device IMU preintegration: IMUPreintegrator::update (src/IMU/IMUPreintegrator.cpp:63-112) applied
 * over the samples of n_edges keyframe intervals, the way KeyFrame::ComputePreInt feeds it (src/KeyFrame.cpp:195-252:
 * the caller lists the samples and their dt, including the duplicated first sample).  gyr/acc are bias-corrected
 * (measurement - bias of the previous keyframe).  Outputs, per interval: imu_meas[VBA_IMU_MEAS_STRIDE] and the 9x9
 * covariance in P,V,phi order as the reference keeps it; imu_info_prv (may be NULL) = inverse of the V/phi-swapped
 * covariance, i.e. exactly what vba_problem.imu_info_prv expects (src/Optimizer.cpp:273-280). */
int vba_preintegrate(void *handle, int32_t n_edges, const int32_t *sample_begin, const double *gyr, const double *acc,
                     const double *dt, double gyr_meas_cov, double acc_meas_cov, double *imu_meas, double *cov_pvphi,
                     double *imu_info_prv);

/* ---- IMU-aided per-frame pose optimisation ----
 * Optimizer::PoseOptimization(Frame*, KeyFrame* pLastKF, IMUPreintegrator, gw, bComputeMarg)   src/Optimizer.cpp:2046-2317
 * Optimizer::PoseOptimization(Frame*, Frame*   pLastFrame, IMUPreintegrator, gw, bComputeMarg) src/Optimizer.cpp:1671-2044
 * Optimizer::PoseOptimization(Frame*)  (vision only, BASELINE configs[0])                        src/Optimizer.cpp:3610-3835
 * Everything between the vertex set-up and the write-back: the four optimize(10) rounds of Levenberg-Marquardt on the
 * 6-, 15- or 30-dimensional system, the chi2 > 5.991 reclassification after every round, the kernel removal after the third,
 * and computeMarginals.  One call solves a batch of independent frames (one workgroup per frame). */
#define VBA_FRAME_KF 0
#define VBA_FRAME_FRAME 1
#define VBA_FRAME_VISION 2
#define VBA_NAV_STRIDE 22 /* NavState: P(3) q(4, xyzw) V(3) bg(3) ba(3) dbg(3) dba(3)   src/IMU/NavState.h:124-138 */
typedef struct vba_frame_problem {
    int32_t last_is_frame;  /* VBA_FRAME_*: 0 last keyframe, fixed (:2082-2097); 1 last frame, free, tied to its marginal prior
                             * (:1710-1747); 2 vision only: nav[0..6] is T_cw as SE3Quat (t, q xyzw), one VertexSE3Expmap +
                             * EdgeSE3ProjectXYZOnlyPose edges (:3623-3672), no IMU fields are read */
    int32_t compute_marg;   /* bComputeMarg */
    int32_t n_obs;          /* monocular correspondences of the frame (mvpMapPoints[i] != NULL, mvuRight[i] < 0) */
    int32_t n_obs_last;     /* those of the last frame (last_is_frame only) */
    double nav[VBA_NAV_STRIDE];      /* in: pFrame->GetNavState(); out: the optimised state (P, R, V, dbg, dba change) */
    double nav_last[VBA_NAV_STRIDE]; /* pLastKF / pLastFrame NavState; never written back by the reference */
    const double *obs_pw;   /* [n_obs][3] MapPoint world positions */
    const double *obs_uv;   /* [n_obs][2] undistorted keypoints */
    const double *obs_w;    /* [n_obs] invSigma2 */
    const double *last_pw, *last_uv, *last_w; /* the same for the last frame */
    double K[4];            /* fx fy cx cy */
    double T_cb[7];         /* as in vba_problem */
    double g_w[3];
    double imu_meas[VBA_IMU_MEAS_STRIDE]; /* imupreint (last -> current) */
    double imu_cov_pvphi[81];             /* its covariance; information of EdgeNavStatePVR = inverse, same P,V,phi order */
    double prior_nav[VBA_NAV_STRIDE];     /* pLastFrame->mNavStatePrior (last_is_frame) */
    double prior_info[225];               /* pLastFrame->mMargCovInv, row-major 15x15, order P V phi bg ba */
    double inv_bg_rw2, inv_ba_rw2;
} vba_frame_problem;

typedef struct vba_frame_result {
    int32_t n_inliers;      /* the function's return value: nInitialCorrespondences - nBad (0 if fewer than 3 correspondences) */
    int32_t status;         /* VBA_OK */
    int32_t its_done[4];    /* LM iterations of each round */
    uint8_t *outlier;       /* [n_obs] caller-allocated: pFrame->mvbOutlier */
    uint8_t *outlier_last;  /* [n_obs_last] caller-allocated or NULL: pLastFrame->mvbOutlier */
    double chi2_round[4];   /* activeRobustChi2 at the end of each round (diagnostic) */
    double marg_cov_inv[225]; /* pFrame->mMargCovInv when compute_marg (pFrame->mNavStatePrior = the returned nav) */
} vba_frame_result;

int vba_pose_optimize(void *handle, int32_t n_frames, vba_frame_problem *const *inout, vba_frame_result *const *out);

/* ---- on-disk problem format: one vba_problem per file, so that windows recorded from a live system can
 * be replayed as fixtures.  Little-endian; header "VBAP" u32 version(=2) then the scalar fields in struct order
 * (i32 variant n_kf n_kf_free n_pt n_obs n_imu algo its_stage1 its_stage2 protocol robust has_kf_fix solver reserved(0), f64 K[4]
 * T_cb[7] g_w[3] inv_bg_rw2 inv_ba_rw2 huber_vis huber_prv huber_bias chi2_th depth_min rho_min), then the arrays in struct order
 * with the sizes of the struct comments.  Version 1 (no solver / reserved ints; solver = VBA_SOLVER_LDLT) is still read.
 * vba_problem_load allocates one block that vba_problem_free releases. */
int vba_problem_save(const char *path, const vba_problem *p);
int vba_problem_load(const char *path, vba_problem **out);
void vba_problem_free(vba_problem *p);

/* Host threads one handle of this process uses for packing, the host half of the structure build and the scatter of the results:
 * this rank's share of the cores it may run on (VBA_UPLOAD_THREADS, else cores / LOCAL_WORLD_SIZE, at most 16). */
int vba_host_threads(void);

int vba_set_profile(void *handle, int32_t enable);
int vba_get_profile(void *handle, vba_profile *out);

#ifdef __cplusplus
}
#endif
#endif /* VISLAM_BA_H */
