"""ctypes driver of the C++ host facade (mc_slam_amd/host/libvba_facade.so): builds a KeyFrame / MapPoint map
from a synthetic Problem the way ORB-SLAM holds it, calls Optimizer::LocalBAPRVIDP / LocalBundleAdjustment."""
import ctypes as C
import os

import numpy as np

from mc_slam_amd import abi, synth

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "mc_slam_amd", "host", "libvba_facade.so")
_lib = None
_pd = C.POINTER(C.c_double)
_pf = C.POINTER(C.c_float)
_pl = C.POINTER(C.c_long)


def lib():
    global _lib
    if _lib is None:
        l = C.CDLL(_SO)
        l.fc_create.restype = C.c_void_p
        l.fc_destroy.argtypes = [C.c_void_p]
        l.fc_set_tbc.argtypes = [_pd, _pd]
        l.fc_add_keyframe.argtypes = [C.c_void_p, C.c_long, _pd, _pd, C.c_long, C.c_int]
        l.fc_set_pose_tcw.argtypes = [C.c_void_p, C.c_long, _pf]
        l.fc_set_covisible.argtypes = [C.c_void_p, C.c_long, _pl, C.c_int]
        l.fc_set_preint.argtypes = [C.c_void_p, C.c_long, _pd, _pd]
        l.fc_add_mappoint.argtypes = [C.c_void_p, C.c_long, _pf, C.c_long]
        l.fc_add_observation.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_float, C.c_float, C.c_int]
        l.fc_local_ba_prvidp.argtypes = [C.c_void_p, _pl, C.c_int, _pd, C.c_int, C.c_int]
        l.fc_local_ba_vision.argtypes = [C.c_void_p, C.c_long, C.c_int]
        l.fc_local_ba_prvidp_flag.argtypes = [C.c_void_p, _pl, C.c_int, _pd, C.POINTER(C.c_bool)]
        l.fc_last_timing.argtypes = [_pd]
        l.fc_last_timing.restype = None
        l.fc_local_ba_prv_xyz.argtypes = [C.c_void_p, _pl, C.c_int, _pd, C.c_int, C.c_int]
        l.fc_local_ba_vision_list.argtypes = [C.c_void_p, _pl, C.c_int, C.c_int, C.c_int]
        l.fc_global_ba_prv.argtypes = [C.c_void_p, _pd, C.c_int, C.c_long, C.c_int, C.c_int, C.c_int]
        l.fc_global_ba_vision.argtypes = [C.c_void_p, C.c_int, C.c_long, C.c_int, C.c_int, C.c_int]
        l.fc_get_gba.argtypes = [C.c_void_p, C.c_long, _pd, _pf, _pl]
        l.fc_get_mappoint_gba.argtypes = [C.c_void_p, C.c_long, _pf, _pl]
        l.fc_get_nav.argtypes = [C.c_void_p, C.c_long, _pd, _pf]
        l.fc_get_mappoint.argtypes = [C.c_void_p, C.c_long, _pf, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        l.fc_map_updated.argtypes = [C.c_void_p]
        l.fc_last_problem.restype = C.POINTER(abi.vba_problem)
        l.fc_last_result.restype = C.POINTER(abi.vba_result)
        _lib = l
    return _lib


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(_pd)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32).ctypes.data_as(_pf)


def octave_of(w):
    return int(round(np.log(1.0 / w) / (2 * np.log(1.2))))


class FacadeMap:
    """A map holding exactly one synthetic window.  Keyframe mnId = time index of the generator."""

    def __init__(self, p: abi.Problem):
        L = lib()
        self.L, self.p = L, p
        self.m = L.fc_create()
        R_bc, p_bc = p.truth["R_bc"], p.truth["p_bc"]
        L.fc_set_tbc(_d(R_bc.reshape(-1)), _d(p_bc))
        self.tidx = list(p.truth["time_index"])              # problem index -> time index (= mnId)
        order = sorted(range(p.n_kf), key=lambda i: self.tidx[i])
        for i in order:
            nav = np.concatenate([p.kf_pose[i], p.kf_vel[i], p.kf_bias[i]])
            L.fc_add_keyframe(self.m, self.tidx[i], _d(nav), _d(p.K), self.tidx[i] - 1, 0)
        if p.variant != abi.VARIANT_SE3_XYZ:
            perm = [0, 1, 2, 6, 7, 8, 3, 4, 5]
            for k in range(p.n_imu):
                cov = np.linalg.inv(p.imu_info_prv[k].reshape(9, 9))[np.ix_(perm, perm)]
                L.fc_set_preint(self.m, self.tidx[p.imu_kf_j[k]], _d(p.imu_meas[k]), _d(cov.reshape(-1)))
        else:
            for i in range(p.n_kf):
                T = np.eye(4, dtype=np.float32)
                T[:3, :3] = synth.quat_to_rot(p.kf_pose[i, 3:]); T[:3, 3] = p.kf_pose[i, :3]
                L.fc_set_pose_tcw(self.m, self.tidx[i], _f(T.reshape(-1)))
        fx, fy, cx, cy = p.K
        for q in range(p.n_pt):
            edges = range(p.pt_obs_begin[q], p.pt_obs_begin[q + 1])
            if p.variant == abi.VARIANT_PRV_IDP:
                ref = p.pt_ref_kf[q]
                T = self.pose_tcw(self.tidx[ref]).astype(np.float64)
                rho, xb, yb = p.pt[q]
                Pc = np.array([xb, yb, 1.0]) / rho
                Pw = T[:3, :3].T @ (Pc - T[:3, 3])
                L.fc_add_mappoint(self.m, q, _f(Pw), self.tidx[ref])
                L.fc_add_observation(self.m, q, self.tidx[ref], np.float32(xb * fx + cx), np.float32(yb * fy + cy), 0)
            else:
                first = p.obs_kf[p.pt_obs_begin[q]]
                L.fc_add_mappoint(self.m, q, _f(p.pt[q]), self.tidx[first])
            for o in edges:
                L.fc_add_observation(self.m, q, self.tidx[p.obs_kf[o]], np.float32(p.obs_uv[o, 0]), np.float32(p.obs_uv[o, 1]),
                                     octave_of(p.obs_w[o]))

    def close(self):
        self.L.fc_destroy(self.m)

    def nav(self, kf_id):
        nav = np.zeros(22); T = np.zeros(16, dtype=np.float32)
        self.L.fc_get_nav(self.m, kf_id, _d(nav) if False else nav.ctypes.data_as(_pd), T.ctypes.data_as(_pf))
        return nav, T.reshape(4, 4)

    def pose_tcw(self, kf_id):
        return self.nav(kf_id)[1]

    def mappoint(self, q):
        Pw = np.zeros(3, dtype=np.float32); n = C.c_int(0); u = C.c_int(0)
        self.L.fc_get_mappoint(self.m, q, Pw.ctypes.data_as(_pf), C.byref(n), C.byref(u))
        return Pw, n.value, u.value

    def window_ids(self):
        p = self.p
        ids = np.array(sorted(self.tidx[i] for i in range(p.n_kf_free)), dtype=np.int64)
        return ids

    def local_ba_prvidp(self, stop=0, extract_only=False):
        ids = self.window_ids()
        return self.L.fc_local_ba_prvidp(self.m, ids.ctypes.data_as(_pl), len(ids), _d(self.p.g_w), stop, 1 if extract_only else 0)

    def local_ba_prvidp_flag(self, flag):
        """Optimizer::LocalBAPRVIDP with the caller's own `bool` (a ctypes c_bool another thread may raise while the call runs)"""
        ids = self.window_ids()
        return self.L.fc_local_ba_prvidp_flag(self.m, ids.ctypes.data_as(_pl), len(ids), _d(self.p.g_w), C.byref(flag))

    def last_timing(self):
        """wall-clock split of the last LocalBAPRVIDP of this thread (ms): extraction, solve, erase + write-back, total"""
        t = np.zeros(4)
        self.L.fc_last_timing(t.ctypes.data_as(_pd))
        return dict(extract_ms=t[0], solve_ms=t[1], writeback_ms=t[2], total_ms=t[3])

    def global_ba_prv(self, n_it=20, loop_kf=0, robust=True, stop=0, extract_only=False):
        return self.L.fc_global_ba_prv(self.m, _d(self.p.g_w), n_it, loop_kf, int(robust), stop, 1 if extract_only else 0)

    def global_ba_vision(self, n_it=20, loop_kf=0, robust=True, stop=0, extract_only=False):
        return self.L.fc_global_ba_vision(self.m, n_it, loop_kf, int(robust), stop, 1 if extract_only else 0)

    def gba(self, kf_id):
        nav = np.zeros(22); T = np.zeros(16, dtype=np.float32); n = C.c_long(0)
        self.L.fc_get_gba(self.m, kf_id, nav.ctypes.data_as(_pd), T.ctypes.data_as(_pf), C.byref(n))
        return nav, T.reshape(4, 4), n.value

    def mappoint_gba(self, q):
        Pw = np.zeros(3, dtype=np.float32); n = C.c_long(0)
        self.L.fc_get_mappoint_gba(self.m, q, Pw.ctypes.data_as(_pf), C.byref(n))
        return Pw, n.value

    def local_ba_prv_xyz(self, stop=0, extract_only=False):
        ids = self.window_ids()
        return self.L.fc_local_ba_prv_xyz(self.m, ids.ctypes.data_as(_pl), len(ids), _d(self.p.g_w), stop, 1 if extract_only else 0)

    def local_ba_vision_list(self, ids, stop=0, extract_only=False):
        ids = np.asarray(ids, dtype=np.int64)
        return self.L.fc_local_ba_vision_list(self.m, ids.ctypes.data_as(_pl), len(ids), stop, 1 if extract_only else 0)

    def local_ba_vision(self, stop=0):
        p = self.p
        free = sorted(self.tidx[i] for i in range(p.n_kf_free))
        cur = free[-1]
        cov = np.array(free[:-1], dtype=np.int64)
        self.L.fc_set_covisible(self.m, cur, cov.ctypes.data_as(_pl), len(cov))
        return self.L.fc_local_ba_vision(self.m, cur, stop)


def last_ids():
    """(MapPoint mnId per landmark row, KeyFrame mnId per keyframe row) of the last packed window"""
    L = lib()
    L.fc_last_mp_ids.argtypes = [_pl, C.c_int]; L.fc_last_kf_ids.argtypes = [_pl, C.c_int]
    a = np.zeros(1 << 20, dtype=np.int64); b = np.zeros(4096, dtype=np.int64)
    n = L.fc_last_mp_ids(a.ctypes.data_as(_pl), len(a)); m = L.fc_last_kf_ids(b.ctypes.data_as(_pl), len(b))
    return a[:n].copy(), b[:m].copy()


def last_problem() -> abi.Problem:
    """copy of the arrays the facade packed for vba_solve"""
    s = lib().fc_last_problem().contents
    def arr(ptr, n, dt=np.float64):
        if n == 0:
            return np.zeros(0, dtype=dt)
        return np.ctypeslib.as_array(ptr, shape=(n,)).copy()
    pr = abi.Problem(
        variant=s.variant, n_kf_free=s.n_kf_free, kf_pose=arr(s.kf_pose, 7 * s.n_kf), pt=arr(s.pt, 3 * s.n_pt),
        pt_obs_begin=arr(s.pt_obs_begin, s.n_pt + 1, np.int32), obs_kf=arr(s.obs_kf, s.n_obs, np.int32),
        obs_uv=arr(s.obs_uv, 2 * s.n_obs), obs_w=arr(s.obs_w, s.n_obs), K=np.array(s.K[:]),
        kf_vel=arr(s.kf_vel, 3 * s.n_kf), kf_bias=arr(s.kf_bias, 12 * s.n_kf), pt_ref_kf=arr(s.pt_ref_kf, s.n_pt, np.int32),
        T_cb=np.array(s.T_cb[:]), g_w=np.array(s.g_w[:]), imu_kf_i=arr(s.imu_kf_i, s.n_imu, np.int32),
        imu_kf_j=arr(s.imu_kf_j, s.n_imu, np.int32), imu_meas=arr(s.imu_meas, 61 * s.n_imu), imu_info_prv=arr(s.imu_info_prv, 81 * s.n_imu),
        algo=s.algo, its_stage1=s.its_stage1, its_stage2=s.its_stage2, chi2_th=s.chi2_th, depth_min=s.depth_min, rho_min=s.rho_min,
        huber_vis=s.huber_vis, huber_prv=s.huber_prv, huber_bias=s.huber_bias, protocol=s.protocol, robust=s.robust,
        kf_fix=(np.ctypeslib.as_array(s.kf_fix, shape=(s.n_kf,)).copy() if s.kf_fix else None))
    return pr


class FrameScene:
    """The map points, the last keyframe / frame and the current frame of one synthetic FrameProblem, held the way the
    tracker holds them (float32 map points and keypoints); unmatched keypoints are interleaved so that the keypoint
    index <-> correspondence index maps of the facade are exercised."""
    CUR, LAST = 1, 2

    def __init__(self, f: abi.FrameProblem, R_bc, p_bc):
        L = lib()
        L.fc_add_frame.argtypes = [C.c_void_p, C.c_long, _pd, _pd]
        L.fc_frame_set_tcw.argtypes = [C.c_void_p, C.c_long, _pf]
        L.fc_frame_add_obs.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_float, C.c_float, C.c_int]
        L.fc_frame_set_prior.argtypes = [C.c_void_p, C.c_long, _pd, _pd]
        L.fc_pose_optimization.argtypes = [C.c_void_p, C.c_int, C.c_long, C.c_long, _pd, _pd, _pd, C.c_int]
        L.fc_frame_get.argtypes = [C.c_void_p, C.c_long, _pd, _pf, _pd, _pd, C.POINTER(C.c_uint8), C.c_int]
        self.L, self.f = L, f
        self.kind = int(f.last_is_frame)
        self.m = L.fc_create()
        L.fc_set_tbc(_d(R_bc.reshape(-1)), _d(p_bc))
        ident = np.zeros(22); ident[6] = 1.0
        L.fc_add_keyframe(self.m, 0, _d(f.nav_last if self.kind == 0 else ident), _d(f.K), -1, 0)   # last keyframe / map-point owner
        pts = np.vstack([f.obs_pw, f.last_pw]) if f.n_obs_last else f.obs_pw
        for i, P in enumerate(pts):
            L.fc_add_mappoint(self.m, i, _f(P), 0)
        if self.kind == 2:
            L.fc_add_frame(self.m, self.CUR, _d(ident), _d(f.K))
            T = np.eye(4, dtype=np.float32); T[:3, :3] = synth.quat_to_rot(f.nav[3:7]); T[:3, 3] = f.nav[:3]
            L.fc_frame_set_tcw(self.m, self.CUR, _f(T.reshape(-1)))
        else:
            L.fc_add_frame(self.m, self.CUR, _d(f.nav), _d(f.K))
        self.cur_index = self._fill(self.CUR, range(f.n_obs), f.obs_uv, f.obs_w)
        self.last_index = []
        if self.kind == 1:
            L.fc_add_frame(self.m, self.LAST, _d(f.nav_last), _d(f.K))
            L.fc_frame_set_prior(self.m, self.LAST, _d(f.prior_nav), _d(f.prior_info.reshape(-1)))
            self.last_index = self._fill(self.LAST, range(f.n_obs, f.n_obs + f.n_obs_last), f.last_uv, f.last_w)

    def _fill(self, fid, mp_ids, uv, w):
        index, n = [], 0
        for i, mp in enumerate(mp_ids):
            if i % 5 == 0:   # a keypoint without a map point
                self.L.fc_frame_add_obs(self.m, fid, -1, 10.0, 10.0, 0); n += 1
            index.append(n)
            self.L.fc_frame_add_obs(self.m, fid, mp, np.float32(uv[i, 0]), np.float32(uv[i, 1]), octave_of(w[i])); n += 1
        return np.array(index, dtype=np.int64)

    def run(self):
        f = self.f
        return self.L.fc_pose_optimization(self.m, self.kind, self.CUR, 0 if self.kind == 0 else self.LAST, _d(f.imu_meas),
                                           _d(f.imu_cov_pvphi.reshape(-1)), _d(f.g_w), int(f.compute_marg))

    def get(self, fid):
        nav = np.zeros(22); T = np.zeros(16, dtype=np.float32); marg = np.zeros(225); prior = np.zeros(22); o = np.zeros(8192, dtype=np.uint8)
        n = self.L.fc_frame_get(self.m, fid, nav.ctypes.data_as(_pd), T.ctypes.data_as(_pf), marg.ctypes.data_as(_pd), prior.ctypes.data_as(_pd),
                                o.ctypes.data_as(C.POINTER(C.c_uint8)), len(o))
        return nav, T.reshape(4, 4), marg.reshape(15, 15), prior, o[:n].astype(bool)

    def close(self):
        self.L.fc_destroy(self.m)
