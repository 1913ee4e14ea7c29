"""ctypes mirror of include/vislam_ba.h (the C-ABI of the local-BA backend).

`Problem` keeps the caller-owned arrays of one local-BA window as numpy arrays (the layout the host
facade hands over after graph extraction, src/Optimizer.cpp:49-451 of the reference) and exposes them as
a `vba_problem` struct without copying.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

VARIANT_SE3_XYZ, VARIANT_PRV_XYZ, VARIANT_PRV_IDP = 0, 1, 2
PROTO_LOCAL, PROTO_SINGLE = 0, 1
SOLVER_LDLT, SOLVER_PCG = 0, 1
ALGO_GN, ALGO_LM = 0, 1
IMU_MEAS_STRIDE = 61
TRACE_MAX = 64
PROF_N = 8
PROF_NAMES = ["linearize", "control", "schur", "factor", "trsv", "update", "misc", "_"]

# float-rounded Huber deltas exactly as the reference builds them (const float th = sqrt(...)):
# src/Optimizer.cpp:241-242, 327
HUBER_VIS = float(np.float32(np.sqrt(5.991)))
HUBER_PRV = float(np.float32(np.sqrt(100 * 21.666)))
HUBER_BIAS = float(np.float32(np.sqrt(100 * 16.812)))
# src/IMU/imudata.cpp:25-31
GYR_BIAS_RW2 = 2.0e-5 * 2.0e-5
ACC_BIAS_RW2 = 5.0e-3 * 5.0e-3
GYR_MEAS_COV = 1.7e-4 * 1.7e-4 / 0.005
ACC_MEAS_COV = 2.0e-3 * 2.0e-3 / 0.005 * 100

_pd = C.POINTER(C.c_double)
_pi = C.POINTER(C.c_int32)
_pu8 = C.POINTER(C.c_uint8)


class vba_problem(C.Structure):
    _fields_ = [
        ("variant", C.c_int32), ("n_kf", C.c_int32), ("n_kf_free", C.c_int32),
        ("n_pt", C.c_int32), ("n_obs", C.c_int32), ("n_imu", C.c_int32),
        ("kf_pose", _pd), ("kf_vel", _pd), ("kf_bias", _pd), ("pt", _pd),
        ("pt_ref_kf", _pi), ("pt_obs_begin", _pi), ("obs_kf", _pi),
        ("obs_uv", _pd), ("obs_w", _pd),
        ("K", C.c_double * 4), ("T_cb", C.c_double * 7), ("g_w", C.c_double * 3),
        ("imu_kf_i", _pi), ("imu_kf_j", _pi), ("imu_meas", _pd), ("imu_info_prv", _pd),
        ("inv_bg_rw2", C.c_double), ("inv_ba_rw2", C.c_double),
        ("huber_vis", C.c_double), ("huber_prv", C.c_double), ("huber_bias", C.c_double),
        ("algo", C.c_int32), ("its_stage1", C.c_int32), ("its_stage2", C.c_int32),
        ("chi2_th", C.c_double), ("depth_min", C.c_double), ("rho_min", C.c_double),
        ("protocol", C.c_int32), ("robust", C.c_int32), ("kf_fix", _pu8), ("solver", C.c_int32),
    ]


class vba_result(C.Structure):
    _fields_ = [
        ("chi2_vis", C.c_double), ("chi2_prv", C.c_double), ("chi2_bias", C.c_double),
        ("its_done", C.c_int32 * 2), ("n_outliers", C.c_int32), ("status", C.c_int32),
        ("obs_outlier", _pu8), ("obs_chi2", _pd),
        ("n_trace", C.c_int32), ("chi2_trace", C.c_double * TRACE_MAX),
        ("lambda_final", C.c_double), ("lin_iterations", C.c_int32),
    ]


class vba_profile(C.Structure):
    _fields_ = [
        ("ms", C.c_double * PROF_N), ("launches", C.c_int64 * PROF_N),
        ("bytes", C.c_double * PROF_N), ("total_ms", C.c_double), ("factor_flops", C.c_double),
        ("kernel_launches", C.c_int64),
    ]


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


@dataclass
class Problem:
    """One local-BA window in the flat SoA form of `vba_problem`."""
    variant: int
    n_kf_free: int
    kf_pose: np.ndarray            # [n_kf,7]
    pt: np.ndarray                 # [n_pt,3]
    pt_obs_begin: np.ndarray       # [n_pt+1]
    obs_kf: np.ndarray             # [n_obs]
    obs_uv: np.ndarray             # [n_obs,2]
    obs_w: np.ndarray              # [n_obs]
    K: np.ndarray                  # [4]
    kf_vel: Optional[np.ndarray] = None    # [n_kf,3]
    kf_bias: Optional[np.ndarray] = None   # [n_kf,12]
    pt_ref_kf: Optional[np.ndarray] = None
    T_cb: np.ndarray = field(default_factory=lambda: np.array([0, 0, 0, 0, 0, 0, 1.0]))
    g_w: np.ndarray = field(default_factory=lambda: np.zeros(3))
    imu_kf_i: Optional[np.ndarray] = None
    imu_kf_j: Optional[np.ndarray] = None
    imu_meas: Optional[np.ndarray] = None      # [n_imu,61]
    imu_info_prv: Optional[np.ndarray] = None  # [n_imu,81]
    algo: int = ALGO_GN
    its_stage1: int = 5
    its_stage2: int = 10
    chi2_th: float = 5.991
    depth_min: float = 0.0
    rho_min: float = 2e-6
    huber_vis: float = HUBER_VIS
    huber_prv: float = HUBER_PRV
    huber_bias: float = HUBER_BIAS
    protocol: int = 0                          # PROTO_LOCAL / PROTO_SINGLE
    robust: int = 1
    kf_fix: Optional[np.ndarray] = None        # [n_kf] uint8: bit0 PR, bit1 V, bit2 Bias fixed
    solver: int = 0                            # SOLVER_LDLT / SOLVER_PCG
    truth: dict = field(default_factory=dict)  # generator ground truth (not part of the ABI)

    def __post_init__(self):
        self.kf_pose = _f64(self.kf_pose, (-1, 7))
        self.pt = _f64(self.pt, (-1, 3))
        self.pt_obs_begin = _i32(self.pt_obs_begin)
        self.obs_kf = _i32(self.obs_kf)
        self.obs_uv = _f64(self.obs_uv, (-1, 2))
        self.obs_w = _f64(self.obs_w)
        self.K = _f64(self.K)
        n_kf = self.n_kf
        self.kf_vel = _f64(self.kf_vel if self.kf_vel is not None else np.zeros((n_kf, 3)), (-1, 3))
        self.kf_bias = _f64(self.kf_bias if self.kf_bias is not None else np.zeros((n_kf, 12)), (-1, 12))
        self.pt_ref_kf = _i32(self.pt_ref_kf if self.pt_ref_kf is not None else np.zeros(self.n_pt))
        self.T_cb = _f64(self.T_cb)
        self.g_w = _f64(self.g_w)
        self.imu_kf_i = _i32(self.imu_kf_i if self.imu_kf_i is not None else [])
        self.imu_kf_j = _i32(self.imu_kf_j if self.imu_kf_j is not None else [])
        self.imu_meas = _f64(self.imu_meas if self.imu_meas is not None else np.zeros((0, IMU_MEAS_STRIDE)),
                             (-1, IMU_MEAS_STRIDE))
        self.imu_info_prv = _f64(self.imu_info_prv if self.imu_info_prv is not None else np.zeros((0, 81)), (-1, 81))

    n_kf = property(lambda self: self.kf_pose.shape[0])
    n_pt = property(lambda self: self.pt.shape[0])
    n_obs = property(lambda self: self.obs_kf.shape[0])
    n_imu = property(lambda self: self.imu_kf_i.shape[0])

    def copy(self) -> "Problem":
        import copy as _c
        q = _c.copy(self)
        for k in ("kf_pose", "kf_vel", "kf_bias", "pt"):
            setattr(q, k, getattr(self, k).copy())
        return q

    def as_struct(self) -> vba_problem:
        s = vba_problem()
        s.variant, s.n_kf, s.n_kf_free = self.variant, self.n_kf, self.n_kf_free
        s.n_pt, s.n_obs, s.n_imu = self.n_pt, self.n_obs, self.n_imu
        p = lambda a, t: a.ctypes.data_as(t)
        s.kf_pose, s.kf_vel, s.kf_bias, s.pt = (p(self.kf_pose, _pd), p(self.kf_vel, _pd),
                                                p(self.kf_bias, _pd), p(self.pt, _pd))
        s.pt_ref_kf, s.pt_obs_begin, s.obs_kf = p(self.pt_ref_kf, _pi), p(self.pt_obs_begin, _pi), p(self.obs_kf, _pi)
        s.obs_uv, s.obs_w = p(self.obs_uv, _pd), p(self.obs_w, _pd)
        s.K[:] = self.K.tolist()
        s.T_cb[:] = self.T_cb.tolist()
        s.g_w[:] = self.g_w.tolist()
        s.imu_kf_i, s.imu_kf_j = p(self.imu_kf_i, _pi), p(self.imu_kf_j, _pi)
        s.imu_meas, s.imu_info_prv = p(self.imu_meas, _pd), p(self.imu_info_prv, _pd)
        s.inv_bg_rw2, s.inv_ba_rw2 = 1.0 / GYR_BIAS_RW2, 1.0 / ACC_BIAS_RW2
        s.huber_vis, s.huber_prv, s.huber_bias = self.huber_vis, self.huber_prv, self.huber_bias
        s.algo, s.its_stage1, s.its_stage2 = self.algo, self.its_stage1, self.its_stage2
        s.chi2_th, s.depth_min, s.rho_min = self.chi2_th, self.depth_min, self.rho_min
        s.protocol, s.robust = self.protocol, self.robust
        s.solver = self.solver
        if self.kf_fix is not None:
            self.kf_fix = np.ascontiguousarray(self.kf_fix, dtype=np.uint8)
            assert self.kf_fix.shape == (self.n_kf,)
            s.kf_fix = p(self.kf_fix, _pu8)
        return s


@dataclass
class Result:
    chi2_vis: float
    chi2_prv: float
    chi2_bias: float
    its_done: tuple
    n_outliers: int
    status: int
    obs_outlier: np.ndarray
    obs_chi2: np.ndarray
    chi2_trace: np.ndarray
    lambda_final: float
    lin_iterations: int = 0


class ResultBuf:
    """Caller-allocated result storage for one window."""

    def __init__(self, n_obs: int, want_chi2: bool = True):
        """want_chi2 = False: vba_result.obs_chi2 stays NULL -- the per-edge chi2 is an optional output (the reference's caller reads
        the erase list only: the classification of src/Optimizer.cpp:496-517 happens inside the call)"""
        self.outlier = np.zeros(max(n_obs, 1), dtype=np.uint8)
        self.chi2 = np.zeros(max(n_obs, 1) if want_chi2 else 1, dtype=np.float64)
        self.n_obs = n_obs
        self.want_chi2 = want_chi2
        self.s = vba_result()
        self.s.obs_outlier = self.outlier.ctypes.data_as(_pu8)
        if want_chi2:
            self.s.obs_chi2 = self.chi2.ctypes.data_as(_pd)

    def get(self) -> Result:
        s = self.s
        return Result(s.chi2_vis, s.chi2_prv, s.chi2_bias, (s.its_done[0], s.its_done[1]), s.n_outliers, s.status,
                      self.outlier[:self.n_obs].copy(), self.chi2[:self.n_obs].copy() if self.want_chi2 else None,
                      np.array(s.chi2_trace[:s.n_trace]), s.lambda_final, s.lin_iterations)


# ---- IMU-aided per-frame pose optimisation (include/vislam_ba.h: vba_frame_problem / vba_frame_result) ----
NAV_STRIDE = 22


class vba_frame_problem(C.Structure):
    _fields_ = [
        ("last_is_frame", C.c_int32), ("compute_marg", C.c_int32), ("n_obs", C.c_int32), ("n_obs_last", C.c_int32),
        ("nav", C.c_double * NAV_STRIDE), ("nav_last", C.c_double * NAV_STRIDE),
        ("obs_pw", _pd), ("obs_uv", _pd), ("obs_w", _pd), ("last_pw", _pd), ("last_uv", _pd), ("last_w", _pd),
        ("K", C.c_double * 4), ("T_cb", C.c_double * 7), ("g_w", C.c_double * 3),
        ("imu_meas", C.c_double * IMU_MEAS_STRIDE), ("imu_cov_pvphi", C.c_double * 81),
        ("prior_nav", C.c_double * NAV_STRIDE), ("prior_info", C.c_double * 225),
        ("inv_bg_rw2", C.c_double), ("inv_ba_rw2", C.c_double),
    ]


class vba_frame_result(C.Structure):
    _fields_ = [
        ("n_inliers", C.c_int32), ("status", C.c_int32), ("its_done", C.c_int32 * 4),
        ("outlier", _pu8), ("outlier_last", _pu8), ("chi2_round", C.c_double * 4), ("marg_cov_inv", C.c_double * 225),
    ]


@dataclass
class FrameProblem:
    """One PoseOptimization(Frame*, KeyFrame*|Frame*, IMUPreintegrator, gw, bComputeMarg) call as flat arrays."""
    nav: np.ndarray
    nav_last: np.ndarray
    obs_pw: np.ndarray
    obs_uv: np.ndarray
    obs_w: np.ndarray
    K: np.ndarray
    T_cb: np.ndarray
    g_w: np.ndarray
    imu_meas: np.ndarray
    imu_cov_pvphi: np.ndarray
    last_is_frame: int = 0
    compute_marg: int = 1
    last_pw: Optional[np.ndarray] = None
    last_uv: Optional[np.ndarray] = None
    last_w: Optional[np.ndarray] = None
    prior_nav: Optional[np.ndarray] = None
    prior_info: Optional[np.ndarray] = None
    truth: dict = field(default_factory=dict)

    def __post_init__(self):
        self.nav = _f64(self.nav); self.nav_last = _f64(self.nav_last)
        self.obs_pw = _f64(self.obs_pw, (-1, 3)); self.obs_uv = _f64(self.obs_uv, (-1, 2)); self.obs_w = _f64(self.obs_w)
        self.last_pw = _f64(self.last_pw if self.last_pw is not None else np.zeros((0, 3)), (-1, 3))
        self.last_uv = _f64(self.last_uv if self.last_uv is not None else np.zeros((0, 2)), (-1, 2))
        self.last_w = _f64(self.last_w if self.last_w is not None else np.zeros(0))
        self.prior_nav = _f64(self.prior_nav if self.prior_nav is not None else np.zeros(NAV_STRIDE))
        self.prior_info = _f64(self.prior_info if self.prior_info is not None else np.zeros((15, 15)), (15, 15))
        self.K = _f64(self.K); self.T_cb = _f64(self.T_cb); self.g_w = _f64(self.g_w)
        self.imu_meas = _f64(self.imu_meas); self.imu_cov_pvphi = _f64(self.imu_cov_pvphi, (9, 9))

    n_obs = property(lambda self: self.obs_pw.shape[0])
    n_obs_last = property(lambda self: self.last_pw.shape[0])

    def copy(self):
        import copy as _c
        q = _c.copy(self)
        q.nav = self.nav.copy()
        return q

    def as_struct(self) -> vba_frame_problem:
        s = vba_frame_problem()
        s.last_is_frame, s.compute_marg, s.n_obs, s.n_obs_last = self.last_is_frame, self.compute_marg, self.n_obs, self.n_obs_last
        s.nav[:] = self.nav.tolist(); s.nav_last[:] = self.nav_last.tolist()
        p = lambda a: a.ctypes.data_as(_pd)
        s.obs_pw, s.obs_uv, s.obs_w = p(self.obs_pw), p(self.obs_uv), p(self.obs_w)
        s.last_pw, s.last_uv, s.last_w = p(self.last_pw), p(self.last_uv), p(self.last_w)
        s.K[:] = self.K.tolist(); s.T_cb[:] = self.T_cb.tolist(); s.g_w[:] = self.g_w.tolist()
        s.imu_meas[:] = self.imu_meas.tolist(); s.imu_cov_pvphi[:] = self.imu_cov_pvphi.reshape(-1).tolist()
        s.prior_nav[:] = self.prior_nav.tolist(); s.prior_info[:] = self.prior_info.reshape(-1).tolist()
        s.inv_bg_rw2, s.inv_ba_rw2 = 1.0 / GYR_BIAS_RW2, 1.0 / ACC_BIAS_RW2
        return s


@dataclass
class FrameResult:
    n_inliers: int
    status: int
    its_done: tuple
    outlier: np.ndarray
    outlier_last: np.ndarray
    chi2_round: np.ndarray
    marg_cov_inv: np.ndarray
    nav: np.ndarray


class FrameResultBuf:
    def __init__(self, f: FrameProblem):
        self.o = np.zeros(max(f.n_obs, 1), dtype=np.uint8)
        self.ol = np.zeros(max(f.n_obs_last, 1), dtype=np.uint8)
        self.n, self.nl = f.n_obs, f.n_obs_last
        self.s = vba_frame_result()
        self.s.outlier = self.o.ctypes.data_as(_pu8)
        self.s.outlier_last = self.ol.ctypes.data_as(_pu8)

    def get(self, st: vba_frame_problem) -> FrameResult:
        s = self.s
        return FrameResult(s.n_inliers, s.status, tuple(s.its_done), self.o[:self.n].copy(), self.ol[:self.nl].copy(),
                           np.array(s.chi2_round[:]), np.array(s.marg_cov_inv[:]).reshape(15, 15), np.array(st.nav[:]))


# ---- on-disk problem format "VBAP" v2 (include/vislam_ba.h: vba_problem_save / vba_problem_load); v1 = the same without the
# two ints `solver`, `reserved0` behind has_kf_fix: still read ----
_HDR = np.dtype([("magic", "S4"), ("version", "<u4"), ("i", "<i4", 14), ("K", "<f8", 4), ("T_cb", "<f8", 7), ("g_w", "<f8", 3),
                 ("s", "<f8", 8)])
_HDR1 = np.dtype([("magic", "S4"), ("version", "<u4"), ("i", "<i4", 12), ("K", "<f8", 4), ("T_cb", "<f8", 7), ("g_w", "<f8", 3),
                  ("s", "<f8", 8)])


def save_problem(path, p: "Problem"):
    """numpy twin of vba_problem_save (same bytes)."""
    s = p.as_struct()
    hd = np.zeros(1, dtype=_HDR)
    hd["magic"] = b"VBAP"; hd["version"] = 2
    hd["i"] = [p.variant, p.n_kf, p.n_kf_free, p.n_pt, p.n_obs, p.n_imu, p.algo, p.its_stage1, p.its_stage2, p.protocol, p.robust,
               1 if p.kf_fix is not None else 0, p.solver, 0]
    hd["K"], hd["T_cb"], hd["g_w"] = p.K, p.T_cb, p.g_w
    hd["s"] = [s.inv_bg_rw2, s.inv_ba_rw2, p.huber_vis, p.huber_prv, p.huber_bias, p.chi2_th, p.depth_min, p.rho_min]
    with open(path, "wb") as f:
        f.write(hd.tobytes())
        for a in (p.kf_pose, p.kf_vel, p.kf_bias, p.pt, p.pt_ref_kf, p.pt_obs_begin, p.obs_kf, p.obs_uv, p.obs_w, p.imu_kf_i, p.imu_kf_j,
                  p.imu_meas, p.imu_info_prv):
            f.write(np.ascontiguousarray(a).tobytes())
        if p.kf_fix is not None:
            f.write(np.ascontiguousarray(p.kf_fix, dtype=np.uint8).tobytes())


def load_problem(path) -> "Problem":
    """numpy twin of vba_problem_load."""
    raw = open(path, "rb").read()
    if len(raw) < 8 or raw[:4] != b"VBAP" or int(np.frombuffer(raw[4:8], "<u4")[0]) not in (1, 2):
        raise ValueError("not a VBAP v1 / v2 file")
    hdt = _HDR if int(np.frombuffer(raw[4:8], "<u4")[0]) == 2 else _HDR1
    hd = np.frombuffer(raw[:hdt.itemsize], dtype=hdt)[0]
    variant, n_kf, n_free, n_pt, n_obs, n_imu, algo, its1, its2, proto, robust, has_fix = [int(x) for x in hd["i"][:12]]
    solver = int(hd["i"][12]) if hdt is _HDR else 0
    off = [hdt.itemsize]

    def take(n, dt):
        a = np.frombuffer(raw, dtype=dt, count=n, offset=off[0]).copy()
        off[0] += a.nbytes
        return a
    pose, vel, bias, pt = take(7 * n_kf, "<f8"), take(3 * n_kf, "<f8"), take(12 * n_kf, "<f8"), take(3 * n_pt, "<f8")
    ref, beg, okf = take(n_pt, "<i4"), take(n_pt + 1, "<i4"), take(n_obs, "<i4")
    uv, w = take(2 * n_obs, "<f8"), take(n_obs, "<f8")
    ii, ij, meas, info = take(n_imu, "<i4"), take(n_imu, "<i4"), take(61 * n_imu, "<f8"), take(81 * n_imu, "<f8")
    fix = take(n_kf, "u1") if has_fix else None
    if off[0] != len(raw):
        raise ValueError("trailing bytes")
    sc = hd["s"]
    return Problem(variant=variant, n_kf_free=n_free, kf_pose=pose, pt=pt, pt_obs_begin=beg, obs_kf=okf, obs_uv=uv, obs_w=w, K=hd["K"].copy(),
                   kf_vel=vel, kf_bias=bias, pt_ref_kf=ref, T_cb=hd["T_cb"].copy(), g_w=hd["g_w"].copy(), imu_kf_i=ii, imu_kf_j=ij,
                   imu_meas=meas, imu_info_prv=info, algo=algo, its_stage1=its1, its_stage2=its2, chi2_th=float(sc[5]),
                   depth_min=float(sc[6]), rho_min=float(sc[7]), huber_vis=float(sc[2]), huber_prv=float(sc[3]), huber_bias=float(sc[4]),
                   protocol=proto, robust=robust, kf_fix=fix, solver=solver)
