"""On-disk problem format: the C writer/reader of the library and the numpy twin agree byte for byte;
no GPU involved (the functions are plain host code of the C-ABI library)."""
import ctypes as C
import filecmp

import numpy as np
import pytest

from mc_slam_amd import abi, synth, backend


def _lib():
    l = C.CDLL(backend.LIB_PATH)
    l.vba_problem_save.argtypes = [C.c_char_p, C.POINTER(abi.vba_problem)]
    l.vba_problem_load.argtypes = [C.c_char_p, C.POINTER(C.POINTER(abi.vba_problem))]
    l.vba_problem_free.argtypes = [C.POINTER(abi.vba_problem)]
    return l


@pytest.mark.parametrize("variant", [2, 1, 0])
def test_c_and_numpy_writers_agree_and_round_trip(tmp_path, variant):
    p = synth.make_window(variant, n_kf=7, n_fixed=1 if variant else 2, n_pt=90, n_obs=420, seed=70 + variant)
    if variant == 1:
        p.protocol, p.robust, p.its_stage1, p.its_stage2 = abi.PROTO_SINGLE, 0, 20, 0
        p.kf_fix = np.zeros(p.n_kf, np.uint8); p.kf_fix[0] = 5
    l = _lib()
    a, b, c = str(tmp_path / "a.vbap"), str(tmp_path / "b.vbap"), str(tmp_path / "c.vbap")
    s = p.as_struct()
    assert l.vba_problem_save(a.encode(), C.byref(s)) == 0
    abi.save_problem(b, p)
    assert filecmp.cmp(a, b, shallow=False)
    # C reader -> C writer reproduces the file; numpy reader reproduces the arrays
    q = C.POINTER(abi.vba_problem)()
    assert l.vba_problem_load(a.encode(), C.byref(q)) == 0
    assert l.vba_problem_save(c.encode(), q) == 0
    assert filecmp.cmp(a, c, shallow=False)
    assert (q.contents.n_kf, q.contents.n_obs, q.contents.protocol) == (p.n_kf, p.n_obs, p.protocol)
    l.vba_problem_free(q)
    r = abi.load_problem(a)
    for k in ("kf_pose", "kf_vel", "kf_bias", "pt", "pt_ref_kf", "pt_obs_begin", "obs_kf", "obs_uv", "obs_w", "imu_kf_i", "imu_kf_j",
              "imu_meas", "imu_info_prv", "K", "T_cb", "g_w"):
        np.testing.assert_array_equal(getattr(r, k), getattr(p, k))
    assert (r.variant, r.n_kf_free, r.algo, r.its_stage1, r.its_stage2, r.protocol, r.robust) == (p.variant, p.n_kf_free, p.algo, p.its_stage1, p.its_stage2, p.protocol, p.robust)
    assert (r.chi2_th, r.depth_min, r.rho_min, r.huber_vis, r.huber_prv, r.huber_bias) == (p.chi2_th, p.depth_min, p.rho_min, p.huber_vis, p.huber_prv, p.huber_bias)
    assert (r.kf_fix is None) == (p.kf_fix is None)


def test_solver_field_round_trips_and_v1_files_still_load(tmp_path):
    """version 2 of the format carries vba_problem.solver; a version-1 file (no solver field) loads as VBA_SOLVER_LDLT"""
    p = synth.make_window(2, n_kf=6, n_fixed=1, n_pt=60, n_obs=240, seed=31)
    p.solver = abi.SOLVER_PCG
    l = _lib()
    a, b, v1 = str(tmp_path / "a.vbap"), str(tmp_path / "b.vbap"), str(tmp_path / "v1.vbap")
    s = p.as_struct()
    assert l.vba_problem_save(a.encode(), C.byref(s)) == 0
    abi.save_problem(b, p)
    assert filecmp.cmp(a, b, shallow=False)
    q = C.POINTER(abi.vba_problem)()
    assert l.vba_problem_load(a.encode(), C.byref(q)) == 0 and q.contents.solver == abi.SOLVER_PCG
    l.vba_problem_free(q)
    assert abi.load_problem(a).solver == abi.SOLVER_PCG
    # the same window as a version-1 file: header without the two trailing ints
    raw = open(a, "rb").read()
    open(v1, "wb").write(raw[:4] + (1).to_bytes(4, "little") + raw[8:56] + raw[64:])
    q = C.POINTER(abi.vba_problem)()
    assert l.vba_problem_load(v1.encode(), C.byref(q)) == 0
    assert q.contents.solver == abi.SOLVER_LDLT and q.contents.n_obs == p.n_obs and q.contents.huber_vis == p.huber_vis
    l.vba_problem_free(q)
    r = abi.load_problem(v1)
    assert r.solver == abi.SOLVER_LDLT
    np.testing.assert_array_equal(r.obs_uv, p.obs_uv)


def test_reader_rejects_damaged_files(tmp_path):
    p = synth.make_window(2, n_kf=6, n_fixed=1, n_pt=60, n_obs=240, seed=31)
    l = _lib()
    a = str(tmp_path / "a.vbap")
    abi.save_problem(a, p)
    raw = open(a, "rb").read()
    q = C.POINTER(abi.vba_problem)()
    for name, data in (("trunc", raw[:-9]), ("extra", raw + b"x"), ("magic", b"XBAP" + raw[4:]), ("ver", raw[:4] + b"\x03\x00\x00\x00" + raw[8:])):
        f = str(tmp_path / name)
        open(f, "wb").write(data)
        assert l.vba_problem_load(f.encode(), C.byref(q)) != 0 and not q
    assert l.vba_problem_load(str(tmp_path / "missing").encode(), C.byref(q)) != 0


def test_landmarks_in_the_callers_order_are_grouped_by_first_local_keyframe():
    """synth.make_window(landmark_order="caller") hands landmarks over the way the reference's caller does (lLocalMapPoints is filled
    keyframe by keyframe over lLocalKeyFrames, src/Optimizer.cpp:59-78): the first FREE keyframe (problem index) that observes a
    landmark -- its own reference observation included -- never decreases along the list; same landmarks, same sizes as the random order"""
    for variant, kw in ((abi.VARIANT_PRV_IDP, dict(n_kf=12, n_fixed=2, n_pt=300, n_obs=1500)), (abi.VARIANT_SE3_XYZ, dict(algo=abi.ALGO_LM, n_kf=10, n_fixed=2, n_pt=200, n_obs=1000))):
        p = synth.make_window(variant, seed=77, landmark_order="caller", **kw)
        q = synth.make_window(variant, seed=77, **kw)
        assert (p.n_pt, p.n_obs, p.n_kf, p.n_kf_free) == (q.n_pt, q.n_obs, q.n_kf, q.n_kf_free)
        first = []
        for i in range(p.n_pt):
            ks = list(p.obs_kf[p.pt_obs_begin[i]:p.pt_obs_begin[i + 1]])
            if variant == abi.VARIANT_PRV_IDP:
                ks.append(int(p.pt_ref_kf[i]))
            first.append(min(k for k in ks if k < p.n_kf_free))
        assert all(a <= b for a, b in zip(first, first[1:]))
        assert sorted(np.diff(p.pt_obs_begin)) == sorted(np.diff(q.pt_obs_begin))     # the same tracks, another order
