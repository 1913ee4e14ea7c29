"""ctypes loader for the CPU oracle (oracle/libvba_oracle.so).  TEST INFRASTRUCTURE ONLY: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

from mc_slam_amd import abi

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "libvba_oracle.so")
_lib = None
_pd = C.POINTER(C.c_double)


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(_ROOT, "oracle")])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.vba_oracle_solve.argtypes = [C.POINTER(abi.vba_problem), C.POINTER(abi.vba_result), C.c_void_p, C.c_int]
        _lib.vba_oracle_solve.restype = C.c_int
        _lib.vba_oracle_solve_ex.argtypes = [C.POINTER(abi.vba_problem), C.POINTER(abi.vba_result), C.c_void_p, C.c_int, C.c_int]
        _lib.vba_oracle_solve_ex.restype = C.c_int
        _lib.vba_oracle_linearize.argtypes = [C.POINTER(abi.vba_problem), C.c_double, _pd, _pd, _pd, _pd]
        _lib.vba_oracle_linearize.restype = C.c_int
    return _lib


def P(a):
    return a.ctypes.data_as(_pd)


_native = None


def timing_lib():
    """The oracle for bench.py's cpu_baseline leg: compiled ON THIS MACHINE with `-O3 -march=native`, the flags of the reference
    (CMakeLists.txt:19, Thirdparty/g2o/CMakeLists.txt:61), into a temporary directory -- the in-tree
    libvba_oracle.so is built `-march=x86-64-v3` so that it runs on whatever box the snapshot lands on.  Falls back to the portable
    build when no compiler is at hand.  Returns (CDLL, "native" | "x86-64-v3")."""
    global _native
    if _native is not None:
        return _native
    import hashlib
    import tempfile
    src = os.path.join(_ROOT, "oracle", "vba_oracle.c")
    tag = hashlib.sha256(open(src, "rb").read()).hexdigest()[:12]
    out = os.path.join(tempfile.gettempdir(), "vba_oracle_native_%s_%d.so" % (tag, os.getuid()))
    try:
        if not os.path.exists(out):
            tmp = out + ".%d.tmp" % os.getpid()
            subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-std=c11", "-fno-fast-math", "-shared",
                                   "-I", os.path.join(_ROOT, "include"), "-o", tmp, src, "-lm"],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
            os.replace(tmp, out)
        l = C.CDLL(out)
        l.vba_oracle_solve_ex.argtypes = [C.POINTER(abi.vba_problem), C.POINTER(abi.vba_result), C.c_void_p, C.c_int, C.c_int]
        l.vba_oracle_solve_ex.restype = C.c_int
        _native = (l, "native")
    except Exception:
        _native = (lib(), "x86-64-v3")
    return _native


def solve_timed(prob: abi.Problem, solver_mode=0):
    """one solve with the timing build: (seconds, Result) -- the clock brackets the C call only"""
    import time
    l, _ = timing_lib()
    q = prob.copy()
    s = q.as_struct()
    rb = abi.ResultBuf(q.n_obs)
    t0 = time.perf_counter()
    rc = l.vba_oracle_solve_ex(C.byref(s), C.byref(rb.s), None, solver_mode, -1)
    dt = time.perf_counter() - t0
    if rc != 0:
        raise RuntimeError("oracle failed rc=%d" % rc)
    return dt, q, rb.get()


def solve(prob: abi.Problem, solver_mode=0, stop=None, stop_after=-1):
    """Runs the oracle on a COPY of prob; returns (solved problem copy, Result).
    stop_after >= 0: the stop flag reads 1 from that terminate() poll on (counted from the first poll of the first optimize();
    the bDoMore check between the stages counts) -- the twin of the backend's vba_debug_set_stop_after."""
    q = prob.copy()
    s = q.as_struct()
    rb = abi.ResultBuf(q.n_obs)
    stop_ptr = None
    if stop is not None:
        stop_ptr = C.cast(C.pointer(stop), C.c_void_p)
    rc = lib().vba_oracle_solve_ex(C.byref(s), C.byref(rb.s), stop_ptr, solver_mode, int(stop_after))
    if rc != 0:
        raise RuntimeError("oracle failed rc=%d" % rc)
    return q, rb.get()


def evaluate(prob: abi.Problem, robust_vis=True):
    """residual-only evaluation at prob's current state: (robust chi2, chi2_vis, chi2_prv, chi2_bias, per-edge chi2, depth)"""
    out = np.zeros(4)
    ch = np.zeros(max(prob.n_obs, 1)); dp = np.zeros(max(prob.n_obs, 1))
    s = prob.as_struct()
    f = lib().vba_oracle_eval
    f.restype = C.c_int
    f(C.byref(s), C.c_int(1 if robust_vis else 0), P(out), P(ch), P(dp))
    return out[0], out[1], out[2], out[3], ch[:prob.n_obs], dp[:prob.n_obs]


def linearize(prob: abi.Problem, lam=0.0, want_H=True):
    """Dense H, b and the Schur solution at prob's current state (all edges active, Huber on)."""
    pdim = 6 if prob.variant == abi.VARIANT_SE3_XYZ else 15
    ldim = 1 if prob.variant == abi.VARIANT_PRV_IDP else 3
    n = pdim * prob.n_kf_free + ldim * prob.n_pt
    H = np.zeros((n, n)) if want_H else None
    b = np.zeros(n)
    x = np.zeros(n)
    chi = C.c_double(0)
    s = prob.as_struct()
    rc = lib().vba_oracle_linearize(C.byref(s), lam, P(H) if want_H else None, P(b), P(x), C.cast(C.pointer(chi), _pd))
    return rc, H, b, x, chi.value


def linearize_ex(prob: abi.Problem, robust_vis=True, lvl=None, lam=0.0):
    """Dense H, b and the active robust chi2 at prob's current state, with the stage's settings spelled out: Huber on the vision
    edges or not, g2o level per vision edge (None: all 0).  Returns (H, b, chi2)."""
    pdim = 6 if prob.variant == abi.VARIANT_SE3_XYZ else 15
    ldim = 1 if prob.variant == abi.VARIANT_PRV_IDP else 3
    n = pdim * prob.n_kf_free + ldim * prob.n_pt
    H = np.zeros((n, n)); b = np.zeros(n); chi = C.c_double(0)
    lv = None if lvl is None else np.ascontiguousarray(lvl, dtype=np.uint8)
    s = prob.as_struct()
    f = lib().vba_oracle_linearize_ex
    f.restype = C.c_int
    f(C.byref(s), C.c_double(lam), C.c_int(1 if robust_vis else 0), None if lv is None else lv.ctypes.data_as(C.POINTER(C.c_uint8)),
      P(H), P(b), None, C.cast(C.pointer(chi), _pd))
    return H, b, chi.value


# ---- unit-level hooks -------------------------------------------------------------------------
def _v(n):
    return np.zeros(n, dtype=np.float64)


def _a(x):
    return np.ascontiguousarray(x, dtype=np.float64)


def call(name, *arrs):
    """call a void vbo_* hook whose arguments are all double* (numpy arrays) or python floats"""
    f = getattr(lib(), name)
    args = []
    for a in arrs:
        if a is None:
            args.append(None)
        elif isinstance(a, (float, int)):
            args.append(C.c_double(a))
        else:
            args.append(P(a))
    f.restype = None
    f(*args)


def so3_exp(w):
    q = _v(4); call("vbo_so3_exp", _a(w), q); return q


def so3_log(q):
    w = _v(3); call("vbo_so3_log", _a(q), w); return w


def so3_jr(w):
    J = _v(9); call("vbo_so3_jr", _a(w), J); return J.reshape(3, 3)


def so3_jrinv(w):
    J = _v(9); call("vbo_so3_jrinv", _a(w), J); return J.reshape(3, 3)


def se3_exp(u):
    o = _v(7); call("vbo_se3_exp", _a(u), o); return o


def quat_to_R(q):
    R = _v(9); call("vbo_quat_to_R", _a(q), R); return R.reshape(3, 3)


def R_to_quat(R):
    q = _v(4); call("vbo_R_to_quat", _a(R).reshape(-1), q); return q


def huber(e, delta):
    r = _v(3); call("vbo_huber", float(e), float(delta), r); return r


def oplus_pr(pose7, d6):
    p = _a(pose7).copy(); call("vbo_oplus_pr", p, _a(d6)); return p


def oplus_se3(T7, d6):
    p = _a(T7).copy(); call("vbo_oplus_se3", p, _a(d6)); return p


def edge_idp(pt3, ref7, obs7, Tcb7, K, uv, jac=True):
    e, Pc = _v(2), _v(3)
    if jac:
        Jr, J1, J2 = _v(2), _v(12), _v(12)
        call("vbo_edge_idp", _a(pt3), _a(ref7), _a(obs7), _a(Tcb7), _a(K), _a(uv), e, Pc, Jr, J1, J2)
        return e, Pc, Jr, J1.reshape(2, 6), J2.reshape(2, 6)
    call("vbo_edge_idp", _a(pt3), _a(ref7), _a(obs7), _a(Tcb7), _a(K), _a(uv), e, Pc, None, None, None)
    return e, Pc


def edge_prxyz(Pw, kf7, Tcb7, K, uv, jac=True):
    e, Pc = _v(2), _v(3)
    if jac:
        Jp, Jk = _v(6), _v(12)
        call("vbo_edge_prxyz", _a(Pw), _a(kf7), _a(Tcb7), _a(K), _a(uv), e, Pc, Jp, Jk)
        return e, Pc, Jp.reshape(2, 3), Jk.reshape(2, 6)
    call("vbo_edge_prxyz", _a(Pw), _a(kf7), _a(Tcb7), _a(K), _a(uv), e, Pc, None, None)
    return e, Pc


def edge_se3xyz(Pw, T7, K, uv, jac=True):
    e, Pc = _v(2), _v(3)
    if jac:
        Jp, Jk = _v(6), _v(12)
        call("vbo_edge_se3xyz", _a(Pw), _a(T7), _a(K), _a(uv), e, Pc, Jp, Jk)
        return e, Pc, Jp.reshape(2, 3), Jk.reshape(2, 6)
    call("vbo_edge_se3xyz", _a(Pw), _a(T7), _a(K), _a(uv), e, Pc, None, None)
    return e, Pc


def edge_prv_error(pi, pj, vi, vj, bi, meas, g):
    e = _v(9); call("vbo_edge_prv_error", _a(pi), _a(pj), _a(vi), _a(vj), _a(bi), _a(meas), _a(g), e); return e


def edge_prv_jac(pi, pj, vi, vj, bi, meas, g, err):
    J = [_v(54), _v(54), _v(27), _v(27), _v(54)]
    call("vbo_edge_prv_jac", _a(pi), _a(pj), _a(vi), _a(vj), _a(bi), _a(meas), _a(g), _a(err), *J)
    return J[0].reshape(9, 6), J[1].reshape(9, 6), J[2].reshape(9, 3), J[3].reshape(9, 3), J[4].reshape(9, 6)


def edge_bias_error(bi, bj):
    e = _v(6); call("vbo_edge_bias_error", _a(bi), _a(bj), e); return e


def edge_navstate_error(navi, navj, meas, g):
    """A6 EdgeNavState::computeError: 15 (rP rV rPhi rBg rBa)"""
    e = _v(15); call("vbo_edge_navstate_error", _a(navi), _a(navj), _a(meas), _a(g), e); return e


def edge_navstate_jac(navi, navj, meas, g, err):
    """A6 EdgeNavState::linearizeOplus: two 15x15, columns P V Phi dBg dBa"""
    Ji, Jj = _v(225), _v(225)
    call("vbo_edge_navstate_jac", _a(navi), _a(navj), _a(meas), _a(g), _a(err), Ji, Jj)
    return Ji.reshape(15, 15), Jj.reshape(15, 15)


def oplus_navstate(nav, upd15):
    out = np.array(nav, dtype=np.float64).copy(); call("vbo_oplus_navstate", out, _a(upd15)); return out


def preint(omega, acc, dts, gyr_cov=abi.GYR_MEAS_COV, acc_cov=abi.ACC_MEAS_COV):
    meas, cov = _v(abi.IMU_MEAS_STRIDE), _v(81)
    call("vbo_preint_reset", meas, cov)
    for w, a, dt in zip(omega, acc, dts):
        call("vbo_preint_update", meas, cov, _a(w), _a(a), float(dt), float(gyr_cov), float(acc_cov))
    return meas, cov.reshape(9, 9)


def prv_information(cov):
    info = _v(81)
    f = lib().vbo_prv_information
    f.restype = C.c_int
    rc = f(P(_a(cov).reshape(-1)), P(info))
    assert rc == 0
    return info.reshape(9, 9)


# ---- IMU-aided per-frame pose optimisation ----
def pose_optimize(f: abi.FrameProblem):
    """vba_oracle_pose_optimize on a copy: FrameResult (with the optimised nav)"""
    s = f.as_struct()
    rb = abi.FrameResultBuf(f)
    fn = lib().vba_oracle_pose_optimize
    fn.restype = C.c_int
    rc = fn(C.byref(s), C.byref(rb.s))
    assert rc == 0
    return rb.get(s)


def frame_linearize(f: abi.FrameProblem, want_H=True):
    n = {0: 15, 1: 30, 2: 6}[int(f.last_is_frame)]
    H = np.zeros((n, n)); b = np.zeros(n); chi = C.c_double(0)
    s = f.as_struct()
    fn = lib().vba_oracle_frame_linearize
    fn.restype = C.c_int
    fn(C.byref(s), P(H) if want_H else None, P(b), C.cast(C.pointer(chi), _pd))
    return H, b, chi.value


def nav_oplus(nav, dpvr, dbias):
    out = np.array(nav, dtype=np.float64).copy()
    call("vbo_nav_oplus", out, _a(dpvr), _a(dbias))
    return out
