"""The C++ host facade (mc_slam_amd/host): same static Optimizer API as the reference, KeyFrame / MapPoint in,
KeyFrame / MapPoint out.  CPU part: graph extraction reproduces the window the map was built from.
GPU part: the facade's in-place results equal a direct C-ABI solve of the extracted window."""
import ctypes as C

import numpy as np
import pytest

from mc_slam_amd import abi, synth

facade = pytest.importorskip("facade_lib")


def _mini(variant):
    if variant == abi.VARIANT_PRV_IDP:
        return synth.make_window(variant, n_kf=9, n_fixed=1, n_pt=250, n_obs=1200, seed=61)
    return synth.make_window(variant, n_kf=9, n_fixed=2, n_pt=250, n_obs=1400, seed=62)


def test_extraction_reproduces_the_window():
    p = _mini(abi.VARIANT_PRV_IDP)
    fm = facade.FacadeMap(p)
    assert fm.local_ba_prvidp(extract_only=True) == 0
    e = facade.last_problem()
    fm_tidx = list(fm.tidx)
    assert (e.variant, e.algo, e.n_kf_free, e.n_kf, e.n_pt, e.n_obs, e.n_imu) == (2, 0, p.n_kf_free, p.n_kf, p.n_pt, p.n_obs, p.n_imu)
    assert (e.its_stage1, e.its_stage2, e.chi2_th, e.depth_min, e.rho_min) == (5, 10, 5.991, 0.01, 2e-6)
    assert (e.huber_vis, e.huber_prv, e.huber_bias) == (abi.HUBER_VIS, abi.HUBER_PRV, abi.HUBER_BIAS)
    mp_ids, kf_ids = facade.last_ids()
    # free keyframes keep the window order; states are passed through unchanged (double NavState)
    assert list(kf_ids[:p.n_kf_free]) == sorted(fm_tidx[i] for i in range(p.n_kf_free))
    row_of = {fm_tidx[i]: i for i in range(p.n_kf)}          # mnId -> row of the generator's problem
    rows = [row_of[t] for t in kf_ids]
    np.testing.assert_array_equal(e.kf_pose, p.kf_pose[rows])
    np.testing.assert_array_equal(e.kf_vel[:p.n_kf_free], p.kf_vel[:p.n_kf_free])
    np.testing.assert_array_equal(e.kf_bias, p.kf_bias[rows])
    # landmarks come in the order the window's keyframes list them (lLocalMapPoints, src/Optimizer.cpp:59-79):
    # a permutation of the generator's rows; per landmark the same edges in ascending keyframe id
    assert sorted(mp_ids) == list(range(p.n_pt))
    inv_row = {r: i for i, r in enumerate(rows)}
    for j, q in enumerate(mp_ids):
        a, b = e.pt_obs_begin[j], e.pt_obs_begin[j + 1]
        c, d = p.pt_obs_begin[q], p.pt_obs_begin[q + 1]
        assert b - a == d - c
        order = np.argsort([fm_tidx[k] for k in p.obs_kf[c:d]], kind="stable")   # map order = ascending mnId
        np.testing.assert_array_equal(e.obs_kf[a:b], [inv_row[k] for k in p.obs_kf[c:d][order]])
        np.testing.assert_array_equal(e.obs_uv[a:b], p.obs_uv[c:d][order])
        np.testing.assert_allclose(e.obs_w[a:b], p.obs_w[c:d][order], rtol=2e-7)   # float32 pyramid table
        assert e.pt_ref_kf[j] == inv_row[p.pt_ref_kf[q]]
        # inverse depth / bearing are recomputed from the float32 map (src/Optimizer.cpp:355-385)
        np.testing.assert_allclose(e.pt[j], p.pt[q], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(e.T_cb, p.T_cb, atol=1e-12)
    np.testing.assert_array_equal(e.imu_meas, p.imu_meas)
    np.testing.assert_allclose(e.imu_info_prv, p.imu_info_prv, rtol=1e-5)


def test_stop_flag_returns_before_touching_anything():
    p = _mini(abi.VARIANT_PRV_IDP)
    fm = facade.FacadeMap(p)
    before = [fm.nav(t)[0].copy() for t in fm.window_ids()]
    fm.local_ba_prvidp(stop=1)     # src/Optimizer.cpp:453-455 -- needs no GPU: returns before vba_solve
    after = [fm.nav(t)[0] for t in fm.window_ids()]
    assert all((a == b).all() for a, b in zip(before, after)) and fm.L.fc_map_updated(fm.m) == 0
    fm.close()


@pytest.mark.gpu
def test_bool_flag_raised_from_another_thread_inside_local_ba_prvidp(oracle):
    """The path the live system takes: Tracking calls LocalMapping::InterruptBA, which sets the `bool` mbAbortBA that LocalMapping
    handed to Optimizer::LocalBAPRVIDP as pbStopFlag (/root/reference/src/LocalMapping.cpp:1769-1772, 1035) -- while the solve
    runs.  The facade passes the bool* straight to vba_solve_b (no mirror thread); the backend forwards it to the device whenever
    it enqueues an iteration and while it waits.  Whatever the moment, the map must come back as the oracle's result for SOME
    poll count; at least one of the delays must land inside the solve (status 1: stage 2 skipped, stage-1 state written back)."""
    import threading, time
    p = synth.config_c3(seed=21, n_kf=30, n_pt=2500, n_obs=14000)
    fm0 = facade.FacadeMap(p)
    fm0.local_ba_prvidp(extract_only=True)
    e = facade.last_problem()
    fm0.close()
    table = {}
    for n in list(range(0, 17)) + [-1]:
        qo, ro = oracle.solve(e, stop_after=n)
        table.setdefault((ro.status, tuple(ro.its_done)), (qo, ro))
    full = oracle.solve(e)[1]
    assert full.status == 0 and full.its_done[1] >= 1
    # calibration: how long extraction and solve take here (the flag has to go up after the first and before the end of the second)
    for _ in range(2):    # (the first call of the thread creates the backend handle; a map serves one call: the second finds its map points marked)
        fm = facade.FacadeMap(p)
        fm.local_ba_prvidp_flag(C.c_bool(False))
        tm = fm.last_timing()
        fm.close()
    assert tm["solve_ms"] > 0
    seen = set()
    for frac in (0.3, 0.5, 0.15, 0.7, 0.4, 0.25, 0.6, 0.1, 0.85, 0.05):
        delay = (tm["extract_ms"] + frac * tm["solve_ms"]) * 1e-3
        fm = facade.FacadeMap(p)
        flag = C.c_bool(False)
        th = threading.Thread(target=lambda: (time.sleep(delay), setattr(flag, "value", True)))
        th.start()
        fm.local_ba_prvidp_flag(flag)
        th.join()
        if fm.last_timing()["solve_ms"] == 0.0:      # raised before the solve began: the call returned at src/Optimizer.cpp:453-455
            assert fm.L.fc_map_updated(fm.m) == 0
            fm.close()
            continue
        r = facade.lib().fc_last_result().contents
        key = (r.status, (r.its_done[0], r.its_done[1]))
        assert key in table, (delay, key, sorted(table))
        qo, ro = table[key]
        for i, t in enumerate(fm.window_ids()):      # the keyframes hold the oracle's state of that poll count
            nav, _T = fm.nav(int(t))
            assert np.abs(nav[:3] - qo.kf_pose[i, :3]).max() <= 1e-6 and np.abs(nav[7:10] - qo.kf_vel[i]).max() <= 1e-5
        seen.add(key)
        fm.close()
        if any(k[0] == 1 and k[1][0] >= 1 for k in seen):
            break
    assert any(k[0] == 1 and k[1][0] >= 1 for k in seen), ("the flag never arrived inside the solve", sorted(seen), tm)


@pytest.mark.gpu
def test_facade_prvidp_equals_direct_solve():
    from mc_slam_amd import backend
    p = _mini(abi.VARIANT_PRV_IDP)
    fm = facade.FacadeMap(p)
    fm.local_ba_prvidp(extract_only=True)
    e = facade.last_problem()
    ba = backend.LocalBA(0)
    q, r = ba.solve(e)
    ba.close()
    fm2 = facade.FacadeMap(p)
    n_obs_before = [fm2.mappoint(i)[1] for i in range(p.n_pt)]
    fm2.local_ba_prvidp()
    assert fm2.L.fc_map_updated(fm2.m) == 1
    for i, t in enumerate(fm2.window_ids()):
        nav, T = fm2.nav(int(t))
        assert (nav[:7] == q.kf_pose[i]).all() and (nav[7:10] == q.kf_vel[i]).all() and (nav[16:22] == q.kf_bias[i, 6:]).all()
        # float32 Tcw refreshed from the NavState (KeyFrame::UpdatePoseFromNS)
        Rwb = synth.quat_to_rot(nav[3:7]); R_bc, p_bc = p.truth["R_bc"], p.truth["p_bc"]
        Rcw = (Rwb @ R_bc).T
        np.testing.assert_allclose(T[:3, :3], Rcw, atol=2e-6)
        np.testing.assert_allclose(T[:3, 3], -Rcw @ (Rwb @ p_bc + nav[:3]), atol=2e-5)
    # erased observations = the erase list; landmarks rewritten from rho through the updated reference pose
    erased = sum(n_obs_before) - sum(fm2.mappoint(i)[1] for i in range(p.n_pt))
    assert erased == int(r.obs_outlier.sum()) > 0
    mp_ids, kf_ids = facade.last_ids()
    for j in range(0, len(mp_ids), 17):
        Pw, _n, upd = fm2.mappoint(int(mp_ids[j]))
        assert upd == 1
        T = fm2.pose_tcw(int(kf_ids[e.pt_ref_kf[j]])).astype(np.float64)
        Pc = np.array([e.pt[j, 1], e.pt[j, 2], 1.0]) / q.pt[j, 0]
        np.testing.assert_allclose(Pw, T[:3, :3].T @ (Pc - T[:3, 3]), rtol=1e-5, atol=1e-5)
    fm.close(); fm2.close()


@pytest.mark.gpu
def test_facade_vision_equals_direct_solve():
    from mc_slam_amd import backend
    p = _mini(abi.VARIANT_SE3_XYZ)
    fm = facade.FacadeMap(p)
    fm.local_ba_vision()
    res = facade.lib().fc_last_result().contents
    assert res.status == 0 and res.its_done[0] >= 1
    # expected: the same window packed by hand from the generator's arrays in the facade's row order (cur KF,
    # covisibles, fixed observers), with the float32 narrowing the map applies, solved directly through the C-ABI
    ba = backend.LocalBA(0)
    order = [sorted(fm.tidx[i] for i in range(p.n_kf_free))[-1]] + sorted(fm.tidx[i] for i in range(p.n_kf_free))[:-1]
    q, r = ba.solve(_reordered(p, fm, order))
    ba.close()
    assert tuple(res.its_done) == r.its_done and res.n_outliers == r.n_outliers
    for row, t in enumerate(order):
        T = fm.pose_tcw(t)
        np.testing.assert_allclose(T[:3, 3], q.kf_pose[row, :3].astype(np.float32), atol=1e-6)
    fm.close()


def _reordered(p, fm, order):
    """the Problem with keyframe rows in the order the facade packs them (cur KF, covisibles, then fixed observers in
    first-encounter order) and float32-narrowed poses / points, as the map stores them"""
    t2i = {fm.tidx[i]: i for i in range(p.n_kf)}
    fixed = []
    seen = set(order)
    # fixed cameras: in the order the local map points' observations meet them (src/Optimizer.cpp:3898-3915)
    # GetMapPointMatches() order = keypoint order = order in which observations were added = landmark index order
    pts_in_order = []
    mark = set()
    for t in order:
        i = t2i[t]
        qs = sorted({int(np.searchsorted(p.pt_obs_begin, o, side="right") - 1) for o in np.nonzero(p.obs_kf == i)[0]})
        for q in qs:
            if q not in mark:
                mark.add(q); pts_in_order.append(q)
    for q in pts_in_order:
        for o in range(p.pt_obs_begin[q], p.pt_obs_begin[q + 1]):
            t = fm.tidx[p.obs_kf[o]]
            if t not in seen:
                seen.add(t); fixed.append(t)
    rows = [t2i[t] for t in order + fixed]
    inv = {old: new for new, old in enumerate(rows)}
    pose = []
    for i in rows:
        T = np.eye(4, dtype=np.float32)
        T[:3, :3] = synth.quat_to_rot(p.kf_pose[i, 3:]); T[:3, 3] = p.kf_pose[i, :3]
        Td = T.astype(np.float64)
        qd = synth.rot_to_quat(Td[:3, :3])
        pose.append(np.concatenate([Td[:3, 3], qd]))
    begin = [0]; okf = []; ouv = []; ow = []; pts = []
    for q in pts_in_order:
        for o in range(p.pt_obs_begin[q], p.pt_obs_begin[q + 1]):
            okf.append(inv[p.obs_kf[o]]); ouv.append(p.obs_uv[o]); ow.append(p.obs_w[o])
        begin.append(len(okf)); pts.append(np.float32(p.pt[q]).astype(np.float64))
    return abi.Problem(variant=0, n_kf_free=len(order), kf_pose=np.array(pose), pt=np.array(pts), pt_obs_begin=begin, obs_kf=okf,
                       obs_uv=np.array(ouv), obs_w=ow, K=np.float32(p.K).astype(np.float64), algo=abi.ALGO_LM, depth_min=0.0)


# ---- global bundle adjustment through the facade ----
def _gmap(variant):
    return synth.make_window(variant, algo=abi.ALGO_LM, n_kf=10, n_fixed=0 if variant == abi.VARIANT_PRV_XYZ else 1,
                             n_pt=300, n_obs=1800, seed=63, outlier_frac=0.02)


def test_global_ba_extraction():
    p = _gmap(abi.VARIANT_PRV_XYZ)
    fm = facade.FacadeMap(p)
    assert fm.global_ba_prv(n_it=20, robust=False, extract_only=True) == 0
    e = facade.last_problem()
    assert (e.variant, e.algo, e.protocol, e.robust, e.its_stage1, e.its_stage2) == (1, abi.ALGO_LM, abi.PROTO_SINGLE, 0, 20, 0)
    assert e.n_kf_free == e.n_kf == p.n_kf and e.n_imu == p.n_kf - 1 and e.n_pt == p.n_pt and e.n_obs == p.n_obs
    assert list(e.kf_fix) == [5] + [0] * (p.n_kf - 1)       # mnId 0: PR and Bias fixed, V free
    assert e.huber_vis == float(np.float32(np.sqrt(5.99)))
    _mp, kf_ids = facade.last_ids()
    assert list(kf_ids) == list(range(p.n_kf))             # id order
    rows = [list(fm.tidx).index(t) for t in kf_ids]
    np.testing.assert_array_equal(e.kf_pose, p.kf_pose[rows])
    np.testing.assert_array_equal(e.kf_vel, p.kf_vel[rows])
    np.testing.assert_allclose(e.pt, p.pt.astype(np.float32), rtol=0, atol=0)   # MapPoint positions are float32 in the map
    fm.close()
    # vision-only: keyframe 0 goes to the fixed rows
    p = _gmap(abi.VARIANT_SE3_XYZ)
    fm = facade.FacadeMap(p)
    assert fm.global_ba_vision(n_it=10, extract_only=True) == 0
    e = facade.last_problem()
    _mp, kf_ids = facade.last_ids()
    assert (e.variant, e.protocol, e.robust, e.its_stage1) == (0, abi.PROTO_SINGLE, 1, 10)
    assert e.n_kf_free == p.n_kf - 1 and kf_ids[-1] == 0 and e.kf_fix is None
    fm.close()


@pytest.mark.gpu
@pytest.mark.parametrize("loop_kf", [0, 7])
def test_facade_global_ba_prv_equals_direct_solve(loop_kf):
    from mc_slam_amd import backend
    p = _gmap(abi.VARIANT_PRV_XYZ)
    fm = facade.FacadeMap(p)
    fm.global_ba_prv(n_it=12, robust=True, extract_only=True)
    e = facade.last_problem()
    ba = backend.LocalBA(0)
    q, r = ba.solve(e)
    ba.close()
    assert r.status == 0 and r.its_done[0] >= 2
    before = [fm.nav(t)[0].copy() for t in range(p.n_kf)]
    fm.global_ba_prv(n_it=12, loop_kf=loop_kf, robust=True)
    for i in range(p.n_kf):
        live, T = fm.nav(i)
        parked, Tg, n = fm.gba(i)
        got, Tgot = (live, T) if loop_kf == 0 else (parked, Tg)
        assert (got[:7] == q.kf_pose[i]).all() and (got[7:10] == q.kf_vel[i]).all() and (got[16:22] == q.kf_bias[i, 6:]).all()
        Rwb = synth.quat_to_rot(got[3:7]); R_bc, p_bc = p.truth["R_bc"], p.truth["p_bc"]
        Rcw = (Rwb @ R_bc).T
        np.testing.assert_allclose(Tgot[:3, :3], Rcw, atol=2e-6)
        np.testing.assert_allclose(Tgot[:3, 3], -Rcw @ (Rwb @ p_bc + got[:3]), atol=2e-5)
        if loop_kf:   # the live map is untouched, results parked with the loop keyframe id (src/Optimizer.cpp:873-891)
            assert (live == before[i]).all() and n == loop_kf
    mp_ids, _kf = facade.last_ids()
    for j in range(0, len(mp_ids), 13):
        Pw, _n, upd = fm.mappoint(int(mp_ids[j]))
        Pg, n = fm.mappoint_gba(int(mp_ids[j]))
        if loop_kf == 0:
            assert upd == 1 and (Pw == q.pt[j].astype(np.float32)).all()
        else:
            assert upd == 0 and n == loop_kf and (Pg == q.pt[j].astype(np.float32)).all()
    fm.close()


@pytest.mark.gpu
def test_facade_global_ba_vision_equals_direct_solve():
    from mc_slam_amd import backend
    p = _gmap(abi.VARIANT_SE3_XYZ)
    fm = facade.FacadeMap(p)
    fm.global_ba_vision(n_it=10, robust=True, extract_only=True)
    e = facade.last_problem()
    ba = backend.LocalBA(0)
    q, r = ba.solve(e)
    ba.close()
    fm.global_ba_vision(n_it=10, robust=True)
    res = facade.lib().fc_last_result().contents
    assert tuple(res.its_done) == r.its_done and r.its_done[1] == 0
    _mp, kf_ids = facade.last_ids()
    for row in range(e.n_kf_free):
        T = fm.pose_tcw(int(kf_ids[row]))
        np.testing.assert_array_equal(T[:3, 3], q.kf_pose[row, :3].astype(np.float32))
    fm.close()


# ---- per-frame pose optimisation through the facade (BASELINE configs[0]) ----
def _narrowed(f):
    """the FrameProblem the facade ends up handing to vba_pose_optimize: float32 map points (cv::Mat CV_32F)"""
    g = f.copy()
    g.obs_pw = np.float32(f.obs_pw).astype(np.float64)
    g.last_pw = np.float32(f.last_pw).astype(np.float64)
    return g


@pytest.mark.gpu
@pytest.mark.parametrize("kind", [2, 0, 1])
def test_facade_pose_optimization_equals_direct_call(kind):
    from mc_slam_amd import backend
    R_bc, p_bc, _ = synth.extrinsics()
    f = synth.make_frame_vision(seed=1, n_obs=200) if kind == 2 else synth.make_frame(seed=90 + kind, n_obs=180, last_is_frame=bool(kind))
    ba = backend.LocalBA(0)
    r = ba.pose_optimize([_narrowed(f)])[0]
    ba.close()
    sc = facade.FrameScene(f, R_bc, p_bc)
    n_in = sc.run()
    nav, T, marg, prior, outl = sc.get(sc.CUR)
    assert n_in == r.n_inliers > 100
    assert (outl[sc.cur_index] == r.outlier.astype(bool)).all()
    assert outl[[i for i in range(len(outl)) if i not in set(sc.cur_index)]].all()      # unmatched keypoints are not touched
    if kind == 2:     # SetPose(Converter::toCvMat(SE3quat)): float32; the facade re-derives the quaternion from the float32 mTcw
        np.testing.assert_allclose(T[:3, 3], r.nav[:3], atol=5e-7)
        np.testing.assert_allclose(T[:3, :3], synth.quat_to_rot(r.nav[3:7]), atol=1e-6)
    else:
        # (the facade's float32 pyramid table differs from the generator's weights in the last float32 bit)
        np.testing.assert_allclose(nav[:10], r.nav[:10], atol=2e-6)
        np.testing.assert_allclose(nav[16:22], r.nav[16:22], atol=1e-7)
        assert (nav[10:16] == f.nav[10:16]).all()
        np.testing.assert_allclose(marg, r.marg_cov_inv, rtol=1e-4, atol=1e-6 * np.abs(r.marg_cov_inv).max())
        assert (prior == nav).all()                                                     # pFrame->mNavStatePrior = ns_recov
        Rwb = synth.quat_to_rot(nav[3:7]); Rcw = (Rwb @ R_bc).T
        np.testing.assert_allclose(T[:3, 3], -Rcw @ (Rwb @ p_bc + nav[:3]), atol=2e-5)
        if kind == 1:
            _n, _T, _m, _p, ol = sc.get(sc.LAST)
            assert (ol[sc.last_index] == r.outlier_last.astype(bool)).all()
            nl, *_ = sc.get(sc.LAST)
            assert (nl == f.nav_last).all()                                             # the last frame's state is not written back
    sc.close()


def test_list_overload_of_local_bundle_adjustment_extraction():
    """LocalBundleAdjustment(pKF, lLocalKeyFrames, ...) (src/Optimizer.cpp:2975-3336): the window is the given list, the
    keyframe before it is fixed first, then every other observer of the window's map points."""
    p = synth.make_window(abi.VARIANT_SE3_XYZ, n_kf=9, n_fixed=1, n_pt=250, n_obs=1400, seed=64)
    fm = facade.FacadeMap(p)
    ids = sorted(fm.tidx[i] for i in range(p.n_kf))
    win = ids[3:]                                   # keyframes 3..8 are the window; 2 precedes it; 0, 1 are older observers
    assert fm.local_ba_vision_list(win, extract_only=True) == 0
    e = facade.last_problem()
    _mp, kf_ids = facade.last_ids()
    assert (e.variant, e.algo, e.n_kf_free) == (0, abi.ALGO_LM, len(win))
    assert list(kf_ids[:len(win)]) == win and kf_ids[len(win)] == win[0] - 1      # the predecessor is the first fixed camera
    assert set(kf_ids[len(win):]) <= set(ids[:3])
    assert (e.its_stage1, e.its_stage2) == (5, 10)
    fm.close()


@pytest.mark.gpu
def test_list_overload_of_local_bundle_adjustment_runs():
    from mc_slam_amd import backend
    p = synth.make_window(abi.VARIANT_SE3_XYZ, n_kf=9, n_fixed=1, n_pt=250, n_obs=1400, seed=64)
    fm = facade.FacadeMap(p)
    win = sorted(fm.tidx[i] for i in range(p.n_kf))[3:]
    fm.local_ba_vision_list(win, extract_only=True)
    e = facade.last_problem()
    ba = backend.LocalBA(0); q, r = ba.solve(e); ba.close()
    fm.close()
    fm = facade.FacadeMap(p)      # a fresh map: the mnBALocalForKF / mnBAFixedForKF marks of the first extraction persist, as upstream
    fm.local_ba_vision_list(win)
    res = facade.lib().fc_last_result().contents
    assert tuple(res.its_done) == r.its_done and res.n_outliers == r.n_outliers and fm.L.fc_map_updated(fm.m) == 1
    for row, t in enumerate(win):
        np.testing.assert_array_equal(fm.pose_tcw(t)[:3, 3], q.kf_pose[row, :3].astype(np.float32))
    fm.close()


@pytest.mark.gpu
def test_facade_prv_xyz_window_equals_direct_solve():
    """LocalBundleAdjustmentNavStatePRV (src/Optimizer.cpp:937-1388): the VI window with world-XYZ landmarks and LM."""
    from mc_slam_amd import backend
    p = synth.make_window(abi.VARIANT_PRV_XYZ, n_kf=9, n_fixed=1, n_pt=250, n_obs=1400, seed=65)
    fm = facade.FacadeMap(p)
    assert fm.local_ba_prv_xyz(extract_only=True) == 0
    e = facade.last_problem()
    assert (e.variant, e.algo, e.n_kf_free, e.n_imu, e.its_stage1, e.its_stage2, e.depth_min) == (1, abi.ALGO_LM, p.n_kf_free, p.n_imu, 5, 10, 0.0)
    assert e.n_pt == p.n_pt and e.n_obs == p.n_obs
    ba = backend.LocalBA(0); q, r = ba.solve(e); ba.close()
    fm.close()
    fm = facade.FacadeMap(p)
    fm.local_ba_prv_xyz()
    assert fm.L.fc_map_updated(fm.m) == 1
    for i, t in enumerate(fm.window_ids()):
        nav, T = fm.nav(int(t))
        assert (nav[:7] == q.kf_pose[i]).all() and (nav[7:10] == q.kf_vel[i]).all() and (nav[16:22] == q.kf_bias[i, 6:]).all()
    mp_ids, _ = facade.last_ids()
    for j in range(0, len(mp_ids), 11):
        Pw, _n, upd = fm.mappoint(int(mp_ids[j]))
        assert upd == 1 and (Pw == q.pt[j].astype(np.float32)).all()
    fm.close()
