"""Host-side binding of the HIP backend (mc_slam_amd/csrc/libvislam_ba.so) through its C-ABI.

There is NO CPU fallback: if the shared library is missing or no HIP device is present, construction
fails loudly.  The library is built in-tree by `python -c "import __graft_entry__ as g; g.build()"`
(or `make -C mc_slam_amd/csrc`).
"""
import ctypes as C
import os

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VBA_LIB", os.path.join(_HERE, "csrc", "libvislam_ba.so"))   # VBA_LIB: A/B builds in experiments
# the same library built with -DVBA_TEST_HOOKS (vba_debug_*: test / diagnostic hooks that the shipped library does not export)
HOOKS_LIB_PATH = os.environ.get("VBA_LIB", os.path.join(_HERE, "csrc", "libvislam_ba_hooks.so"))
_lib = None
_libs = {}

EXPORTS = ["vba_create", "vba_destroy", "vba_last_error", "vba_solve", "vba_batch_upload", "vba_batch_run",
           "vba_batch_download", "vba_batch_solve", "vba_solve_b", "vba_batch_run_b", "vba_batch_solve_b", "vba_preintegrate", "vba_pose_optimize", "vba_problem_save", "vba_problem_load", "vba_problem_free", "vba_set_profile", "vba_get_profile", "vba_host_threads"]


def load_library(hooks=False):
    """dlopen the in-tree HIP library and declare every entry point of include/vislam_ba.h.  hooks=True: the flavour with the
    test / diagnostic hooks (a second, independent instance of the library)."""
    global _lib
    path = HOOKS_LIB_PATH if hooks else LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise RuntimeError("HIP backend not built: %s is missing (run __graft_entry__.build())" % path)
    lib = C.CDLL(path)
    PP = C.POINTER(C.POINTER(abi.vba_problem))
    PR = C.POINTER(C.POINTER(abi.vba_result))
    lib.vba_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.vba_destroy.argtypes = [C.c_void_p]
    lib.vba_last_error.argtypes = [C.c_void_p]
    lib.vba_last_error.restype = C.c_char_p
    lib.vba_solve.argtypes = [C.c_void_p, C.POINTER(abi.vba_problem), C.POINTER(abi.vba_result), C.c_void_p]
    lib.vba_batch_upload.argtypes = [C.c_void_p, C.c_int32, PP]
    lib.vba_batch_run.argtypes = [C.c_void_p, C.c_void_p]
    lib.vba_batch_download.argtypes = [C.c_void_p, C.c_int32, PP, PR]
    lib.vba_batch_solve.argtypes = [C.c_void_p, C.c_int32, PP, PR, C.c_void_p]
    lib.vba_solve_b.argtypes = [C.c_void_p, C.POINTER(abi.vba_problem), C.POINTER(abi.vba_result), C.c_void_p]
    lib.vba_batch_run_b.argtypes = [C.c_void_p, C.c_void_p]
    lib.vba_batch_solve_b.argtypes = [C.c_void_p, C.c_int32, PP, PR, C.c_void_p]
    _pd, _pi = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    lib.vba_preintegrate.argtypes = [C.c_void_p, C.c_int32, _pi, _pd, _pd, _pd, C.c_double, C.c_double, _pd, _pd, _pd]
    lib.vba_pose_optimize.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.POINTER(abi.vba_frame_problem)), C.POINTER(C.POINTER(abi.vba_frame_result))]
    lib.vba_set_profile.argtypes = [C.c_void_p, C.c_int32]
    lib.vba_get_profile.argtypes = [C.c_void_p, C.POINTER(abi.vba_profile)]
    for n in EXPORTS:
        if n != "vba_last_error":
            getattr(lib, n).restype = C.c_int
    _libs[path] = lib
    if not hooks:
        _lib = lib
    return lib


class LocalBA:
    """One backend handle = one GPU + one stream (vba_create).  Mirrors how the reference owns one
    function-local g2o::SparseOptimizer per call (src/Optimizer.cpp:130), but keeps device buffers alive."""

    def __init__(self, device=0, hooks=False):
        self.lib = load_library(hooks)
        self.h = C.c_void_p()
        rc = self.lib.vba_create(device, C.byref(self.h))
        if rc != 0:
            raise RuntimeError("vba_create(device=%d) failed (rc=%d): no usable HIP device -- the backend has no CPU path"
                               % (device, rc))
        self._keep = None

    def close(self):
        if self.h:
            self.lib.vba_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _err(self, what):
        return RuntimeError("%s: %s" % (what, self.lib.vba_last_error(self.h).decode()))

    @staticmethod
    def _stop_ptr(stop):
        return C.cast(C.pointer(stop), C.c_void_p) if stop is not None else None

    def solve(self, prob: abi.Problem, stop=None):
        """vba_solve on a COPY of prob: returns (solved copy, Result)."""
        q = prob.copy()
        s = q.as_struct()
        rb = abi.ResultBuf(q.n_obs)
        if self.lib.vba_solve(self.h, C.byref(s), C.byref(rb.s), self._stop_ptr(stop)) != 0:
            raise self._err("vba_solve")
        return q, rb.get()

    # ---- device-resident batch interface -------------------------------------------------------
    def upload(self, probs):
        self._probs = [p.copy() for p in probs]
        self._structs = [p.as_struct() for p in self._probs]
        n = len(probs)
        arr = (C.POINTER(abi.vba_problem) * n)(*[C.pointer(s) for s in self._structs])
        self._parr = arr
        if self.lib.vba_batch_upload(self.h, n, arr) != 0:
            raise self._err("vba_batch_upload")

    def run(self, stop=None):
        if self.lib.vba_batch_run(self.h, self._stop_ptr(stop)) != 0:
            raise self._err("vba_batch_run")

    def download(self):
        n = len(self._probs)
        rbs = [abi.ResultBuf(p.n_obs) for p in self._probs]
        rarr = (C.POINTER(abi.vba_result) * n)(*[C.pointer(r.s) for r in rbs])
        if self.lib.vba_batch_download(self.h, n, self._parr, rarr) != 0:
            raise self._err("vba_batch_download")
        return self._probs, [r.get() for r in rbs]

    # ---- fresh windows in, solved windows out (vba_batch_solve: chunks of the batch in flight concurrently) ----
    def pack(self, probs, want_chi2=True):
        """private copies of the windows + the ctypes views vba_batch_solve needs (kept alive by the returned dict)"""
        n = len(probs)
        own = [p.copy() for p in probs]
        structs = [p.as_struct() for p in own]
        rbs = [abi.ResultBuf(p.n_obs, want_chi2) for p in own]
        return dict(n=n, src=list(probs), own=own, structs=structs, rbs=rbs,
                    parr=(C.POINTER(abi.vba_problem) * n)(*[C.pointer(s) for s in structs]),
                    rarr=(C.POINTER(abi.vba_result) * n)(*[C.pointer(r.s) for r in rbs]))

    @staticmethod
    def pack_reset(packed):
        """the solve updates the states in place: put the initial states back (outside any timed region)"""
        for q, p in zip(packed["own"], packed["src"]):
            for k in ("kf_pose", "kf_vel", "kf_bias", "pt"):
                getattr(q, k)[...] = getattr(p, k)

    def solve_packed(self, packed, stop=None):
        if self.lib.vba_batch_solve(self.h, packed["n"], packed["parr"], packed["rarr"], self._stop_ptr(stop)) != 0:
            raise self._err("vba_batch_solve")

    @staticmethod
    def pack_results(packed):
        return packed["own"], [r.get() for r in packed["rbs"]]

    def solve_batch(self, probs, stop=None):
        """vba_batch_solve on copies of probs: (solved copies, Results)"""
        packed = self.pack(probs)
        self.solve_packed(packed, stop)
        return self.pack_results(packed)

    def preintegrate(self, sample_begin, gyr, acc, dt, want_info=True):
        """vba_preintegrate: (imu_meas [E,61], cov_PVphi [E,9,9], info_PphiV [E,9,9] or None)"""
        import numpy as np
        sb = np.ascontiguousarray(sample_begin, dtype=np.int32)
        g = np.ascontiguousarray(gyr, dtype=np.float64).reshape(-1, 3)
        a = np.ascontiguousarray(acc, dtype=np.float64).reshape(-1, 3)
        d = np.ascontiguousarray(dt, dtype=np.float64)
        E = len(sb) - 1
        meas = np.zeros((E, abi.IMU_MEAS_STRIDE)); cov = np.zeros((E, 81)); info = np.zeros((E, 81))
        P = lambda x, t: x.ctypes.data_as(C.POINTER(t))
        rc = self.lib.vba_preintegrate(self.h, E, P(sb, C.c_int32), P(g, C.c_double), P(a, C.c_double), P(d, C.c_double),
                                       abi.GYR_MEAS_COV, abi.ACC_MEAS_COV, P(meas, C.c_double), P(cov, C.c_double),
                                       P(info, C.c_double) if want_info else None)
        if rc != 0:
            raise self._err("vba_preintegrate")
        return meas, cov.reshape(E, 9, 9), (info.reshape(E, 9, 9) if want_info else None)

    def pose_pack(self, frames):
        """ctypes views of a list of abi.FrameProblem for vba_pose_optimize (kept alive by the returned tuple)"""
        n = len(frames)
        structs = [f.as_struct() for f in frames]
        bufs = [abi.FrameResultBuf(f) for f in frames]
        pp = (C.POINTER(abi.vba_frame_problem) * n)(*[C.pointer(s) for s in structs])
        rr = (C.POINTER(abi.vba_frame_result) * n)(*[C.pointer(b.s) for b in bufs])
        return n, structs, bufs, pp, rr, frames

    def pose_reset(self, packed):
        """vba_pose_optimize updates nav in place: put the frames' initial states back before the next run"""
        for s, f in zip(packed[1], packed[5]):
            C.memmove(C.addressof(s) + abi.vba_frame_problem.nav.offset, f.nav.ctypes.data, 8 * abi.NAV_STRIDE)

    def pose_call(self, packed):
        if self.lib.vba_pose_optimize(self.h, packed[0], packed[3], packed[4]) != 0:
            raise self._err("vba_pose_optimize")

    def pose_run(self, packed):
        self.pose_reset(packed)
        self.pose_call(packed)

    def pose_optimize(self, frames):
        """vba_pose_optimize on copies of the FrameProblems: list of abi.FrameResult (with the optimised nav)"""
        packed = self.pose_pack(frames)
        self.pose_run(packed)
        return [b.get(s) for b, s in zip(packed[2], packed[1])]

    def set_profile(self, on=True):
        self.lib.vba_set_profile(self.h, 1 if on else 0)

    def get_profile(self):
        pf = abi.vba_profile()
        self.lib.vba_get_profile(self.h, C.byref(pf))
        d = {abi.PROF_NAMES[i]: dict(ms=pf.ms[i], launches=pf.launches[i], bytes=pf.bytes[i]) for i in range(7)}
        d["factor"]["flops"] = pf.factor_flops
        d["total_ms"] = pf.total_ms
        d["kernel_launches"] = int(pf.kernel_launches)
        return d
