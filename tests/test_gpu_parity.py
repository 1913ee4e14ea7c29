"""GPU parity tests proper: the HIP backend, called through its C-ABI, against the CPU oracle on the same
seeded windows.  Bars (BASELINE.json north_star): final chi2 <= 1e-4 relative, keyframe translations
<= 1e-6 m; on top: same iteration counts and the same outlier bitmap."""
import ctypes as C
import os

import numpy as np
import pytest

from mc_slam_amd import abi, synth, backend

pytestmark = pytest.mark.gpu

CHI2_RTOL = 1e-4
TRANS_ATOL = 1e-6


@pytest.fixture(scope="module")
def ba():
    b = backend.LocalBA(0, hooks=True)
    yield b
    b.close()


def _check(p, q, r, qo, ro, tight=True, trace_rtol=1e-7, state_scale=1.0):
    """state_scale loosens the bars on the STATES only, for windows whose system is numerically singular (see test_gpu_stress)"""
    assert r.status == ro.status
    assert r.its_done == ro.its_done, (r.its_done, ro.its_done, r.chi2_trace, ro.chi2_trace)
    assert abs(r.chi2_vis - ro.chi2_vis) <= CHI2_RTOL * max(ro.chi2_vis, 1e-12)
    assert abs(r.chi2_prv - ro.chi2_prv) <= CHI2_RTOL * max(ro.chi2_prv, 1e-9)
    assert abs(r.chi2_bias - ro.chi2_bias) <= CHI2_RTOL * max(ro.chi2_bias, 1e-9)
    assert np.abs(q.kf_pose[:, :3] - qo.kf_pose[:, :3]).max() <= TRANS_ATOL * state_scale
    assert np.abs(q.kf_pose[:, 3:] - qo.kf_pose[:, 3:]).max() <= 1e-6 * state_scale
    assert np.abs(q.kf_vel - qo.kf_vel).max() <= 1e-5 * state_scale
    assert np.abs(q.pt - qo.pt).max() <= 1e-6 * state_scale * max(1.0, np.abs(qo.pt).max())
    assert (r.obs_outlier == ro.obs_outlier).all()
    np.testing.assert_allclose(r.chi2_trace, ro.chi2_trace, rtol=trace_rtol)
    np.testing.assert_allclose(r.obs_chi2, ro.obs_chi2, rtol=max(1e-5, 100 * trace_rtol), atol=1e-7)
    # fixed keyframes never move
    assert (q.kf_pose[p.n_kf_free:] == p.kf_pose[p.n_kf_free:]).all()


@pytest.mark.parametrize("kw", [
    dict(n_kf=6, n_fixed=1, n_pt=60, n_obs=240, seed=31),
    dict(n_kf=10, n_fixed=1, n_pt=400, n_obs=2000, seed=7),
    dict(n_kf=12, n_fixed=3, n_pt=500, n_obs=2500, seed=8),     # co-observer fixed keyframes, fixed reference KFs
    dict(n_kf=23, n_fixed=1, n_pt=1500, n_obs=9000, seed=9),    # np = 330: pads of the 32-blocked factorisation
])
def test_idp_window_matches_oracle(ba, oracle, kw):
    p = synth.make_window(abi.VARIANT_PRV_IDP, **kw)
    q, r = ba.solve(p)
    qo, ro = oracle.solve(p)
    _check(p, q, r, qo, ro)


@pytest.mark.parametrize("variant,algo,kw", [
    (abi.VARIANT_SE3_XYZ, abi.ALGO_LM, dict(n_kf=6, n_fixed=2, n_pt=60, n_obs=300, seed=33)),
    (abi.VARIANT_SE3_XYZ, abi.ALGO_LM, dict(n_kf=12, n_fixed=2, n_pt=500, n_obs=3000, seed=34)),
    (abi.VARIANT_PRV_XYZ, abi.ALGO_LM, dict(n_kf=6, n_fixed=1, n_pt=60, n_obs=300, seed=32)),
    (abi.VARIANT_PRV_XYZ, abi.ALGO_LM, dict(n_kf=12, n_fixed=1, n_pt=500, n_obs=3000, seed=35)),
    (abi.VARIANT_SE3_XYZ, abi.ALGO_LM, dict(n_kf=10, n_fixed=2, n_pt=300, n_obs=1800, seed=36)),
    (abi.VARIANT_PRV_XYZ, abi.ALGO_LM, dict(n_kf=10, n_fixed=1, n_pt=300, n_obs=1800, seed=37)),
])
def test_xyz_variants_match_oracle(ba, oracle, variant, algo, kw):
    """EdgeSE3ProjectXYZ (vision-only LocalBundleAdjustment, LM) and EdgeNavStatePRPointXYZ (+ IMU chain)."""
    p = synth.make_window(variant, algo=algo, **kw)
    q, r = ba.solve(p)
    qo, ro = oracle.solve(p)
    _check(p, q, r, qo, ro)
    if algo == abi.ALGO_LM:
        assert abs(r.lambda_final - ro.lambda_final) <= 1e-6 * ro.lambda_final


def _shuffled(p, seed, drop_middle=False):
    """the same window with its landmarks in another order and the observations of every landmark in another order
    (the caller's order is arbitrary: the structure build sorts where it has to); drop_middle removes the middle
    observation of every long track, so that tracks are no longer runs of consecutive keyframes"""
    rng = np.random.default_rng(seed)
    q = p.copy()
    nb = p.pt_obs_begin
    order = rng.permutation(len(nb) - 1)
    idx, begin = [], [0]
    for i in order:
        o = np.arange(nb[i], nb[i + 1])
        if drop_middle and len(o) >= 5:
            o = np.delete(o, len(o) // 2)
        o = rng.permutation(o)
        idx.append(o)
        begin.append(begin[-1] + len(o))
    idx = np.concatenate(idx)
    q.pt = p.pt[order].copy()
    if p.pt_ref_kf is not None:
        q.pt_ref_kf = p.pt_ref_kf[order].copy()
    q.obs_kf = p.obs_kf[idx].copy(); q.obs_uv = p.obs_uv[idx].copy(); q.obs_w = p.obs_w[idx].copy()
    q.pt_obs_begin = np.asarray(begin, dtype=np.int32)
    return q


@pytest.mark.parametrize("variant,drop", [(abi.VARIANT_PRV_IDP, False), (abi.VARIANT_PRV_IDP, True), (abi.VARIANT_PRV_XYZ, True),
                                          (abi.VARIANT_SE3_XYZ, False)])
def test_arbitrary_landmark_and_observation_order(ba, oracle, variant, drop):
    """records and item lists are ordered by track on the device; the caller's order must not matter"""
    algo = abi.ALGO_GN if variant == abi.VARIANT_PRV_IDP else abi.ALGO_LM
    p0 = synth.make_window(variant, algo=algo, n_kf=12, n_fixed=2 if variant == abi.VARIANT_SE3_XYZ else 1, n_pt=400, n_obs=2400, seed=41)
    p = _shuffled(p0, 5, drop_middle=drop)
    assert any((np.diff(p.obs_kf[p.pt_obs_begin[i]:p.pt_obs_begin[i + 1]]) < 0).any() for i in range(50))   # really unsorted
    q, r = ba.solve(p)
    qo, ro = oracle.solve(p)
    _check(p, q, r, qo, ro)
    # a batch of the shuffled window and its twins: every code path of the structure build (threaded upload)
    ba.upload([p] * 9); ba.run(); qs, rs = ba.download()
    for qq, rr in zip(qs, rs):
        assert rr.its_done == r.its_done and np.abs(qq.kf_pose - q.kf_pose).max() < 1e-9


def test_empty_window_is_refused_with_a_message(ba):
    p = synth.make_window(abi.VARIANT_PRV_IDP, n_kf=6, n_fixed=1, n_pt=60, n_obs=240, seed=31)
    q = abi.Problem(variant=p.variant, n_kf_free=p.n_kf_free, kf_pose=p.kf_pose, pt=np.zeros((0, 3)), pt_obs_begin=[0], obs_kf=[],
                    obs_uv=np.zeros((0, 2)), obs_w=[], K=p.K, kf_vel=p.kf_vel, kf_bias=p.kf_bias, T_cb=p.T_cb, g_w=p.g_w,
                    imu_kf_i=p.imu_kf_i, imu_kf_j=p.imu_kf_j, imu_meas=p.imu_meas, imu_info_prv=p.imu_info_prv, algo=p.algo)
    with pytest.raises(RuntimeError, match="nothing to optimise"):
        ba.solve(q)


def test_malformed_imu_arguments_are_refused(ba):
    """a negative n_imu, or n_imu > 0 with a NULL IMU array, is an error message, not a memcpy with a huge size"""
    p = synth.make_window(abi.VARIANT_PRV_IDP, n_kf=6, n_fixed=1, n_pt=60, n_obs=240, seed=31)
    q = p.copy()
    s = q.as_struct()
    rb = abi.ResultBuf(q.n_obs)
    s.n_imu = -3
    assert ba.lib.vba_solve(ba.h, C.byref(s), C.byref(rb.s), None) != 0
    assert b"bad sizes" in ba.lib.vba_last_error(ba.h)
    s = q.as_struct()
    s.imu_meas = None
    assert ba.lib.vba_solve(ba.h, C.byref(s), C.byref(rb.s), None) != 0
    assert b"IMU array is NULL" in ba.lib.vba_last_error(ba.h)
    q2, r2 = ba.solve(p)          # the handle is still usable
    assert r2.status == 0


def test_unsupported_combinations_fail_loudly(ba):
    # the reference never runs GN on XYZ landmarks nor LM on inverse-depth ones; the backend says so instead of guessing
    p = synth.make_window(abi.VARIANT_SE3_XYZ, algo=abi.ALGO_GN, n_kf=6, n_fixed=2, n_pt=60, n_obs=300, seed=33)
    with pytest.raises(RuntimeError, match="Levenberg-Marquardt only"):
        ba.solve(p)
    p = synth.make_window(abi.VARIANT_PRV_IDP, algo=abi.ALGO_LM, n_kf=6, n_fixed=1, n_pt=60, n_obs=240, seed=31)
    with pytest.raises(RuntimeError, match="Gauss-Newton only"):
        ba.solve(p)


def test_c2_full_size_matches_oracle(ba, oracle):
    """BASELINE configs[1]: vision-only LocalBundleAdjustment, 20 KF / 2k MapPoints / 12k EdgeSE3ProjectXYZ, LM."""
    p = synth.config_c2()
    q, r = ba.solve(p)
    qo, ro = oracle.solve(p)
    _check(p, q, r, qo, ro)


def test_c3_full_size_matches_oracle(ba, oracle):
    p = synth.config_c3()
    q, r = ba.solve(p)
    qo, ro = oracle.solve(p)
    _check(p, q, r, qo, ro)


@pytest.mark.parametrize("make", [
    lambda: synth.make_window(abi.VARIANT_PRV_IDP, n_kf=16, n_fixed=4, n_pt=600, n_obs=3600, seed=90, tracks="random"),
    lambda: synth.make_window(abi.VARIANT_PRV_XYZ, algo=abi.ALGO_LM, n_kf=14, n_fixed=3, n_pt=500, n_obs=3000, seed=91, tracks="random"),
    lambda: synth.config_c3s(),
    lambda: synth.config_c2s(),
])
def test_scattered_covisibility_windows_match_oracle(ba, oracle, make):
    """co-visibility windows (LocalBundleAdjustment(KeyFrame*, bool*, Map*, LocalMapping*), src/Optimizer.cpp:3861-3875): tracks are
    random subsets of the keyframes that see a landmark (gaps), several fixed co-observers, a fifth of the landmarks anchored in a
    fixed reference keyframe -- none of the layout assumptions of the sliding window hold; alone and inside a batch"""
    p = make()
    q, r = ba.solve(p)
    qo, ro = oracle.solve(p)
    _check(p, q, r, qo, ro)
    if p.n_pt <= 1000:
        ba.upload([p] * 9); ba.run(); qs, rs = ba.download()
        for qq, rr in zip(qs, rs):
            assert rr.its_done == r.its_done and np.abs(qq.kf_pose - q.kf_pose).max() < 1e-9


def test_noise_free_window_terminates_immediately(ba, oracle):
    p = synth.make_window(abi.VARIANT_PRV_IDP, n_kf=8, n_fixed=1, n_pt=200, n_obs=900, seed=41, noise=False)
    q, r = ba.solve(p)
    assert r.its_done == (1, 1) and r.n_outliers == 0 and r.status == 0
    assert np.abs(q.kf_pose - p.truth["pose"]).max() < 1e-6


def test_stop_flag_semantics(ba, oracle):
    p = synth.make_window(abi.VARIANT_PRV_IDP, n_kf=8, n_fixed=1, n_pt=200, n_obs=900, seed=40)
    q, r = ba.solve(p, stop=C.c_int(1))
    assert r.status == 2
    assert (q.kf_pose == p.kf_pose).all() and (q.pt == p.pt).all()
    # device-resident interface with the flag raised: also untouched
    ba.upload([p]); ba.run(stop=C.c_int(1)); qs, rs = ba.download()
    assert rs[0].status == 2 and (qs[0].kf_pose == p.kf_pose).all()


def test_batch_of_ragged_windows_equals_single_solves(ba, oracle):
    ps = [synth.make_window(abi.VARIANT_PRV_IDP, n_kf=8 + 2 * i, n_fixed=1, n_pt=200 + 50 * i, n_obs=900 + 300 * i, seed=50 + i)
          for i in range(4)]
    ba.upload(ps); ba.run(); qs, rs = ba.download()
    for p, q, r in zip(ps, qs, rs):
        q1, r1 = ba.solve(p)
        assert r.its_done == r1.its_done and (r.obs_outlier == r1.obs_outlier).all()
        assert (q.kf_pose == q1.kf_pose).all() and (q.pt == q1.pt).all()   # bit-reproducible
        qo, ro = oracle.solve(p)
        _check(p, q, r, qo, ro)


def test_lm_batch_of_ragged_windows_equals_single_solves(ba, oracle):
    ps = [synth.make_window(abi.VARIANT_SE3_XYZ, n_kf=7 + 2 * i, n_fixed=2, n_pt=150 + 60 * i, n_obs=800 + 400 * i, seed=70 + i)
          for i in range(3)]
    ba.upload(ps); ba.run(); qs, rs = ba.download()
    for p, q, r in zip(ps, qs, rs):
        q1, r1 = ba.solve(p)
        assert r.its_done == r1.its_done and (r.obs_outlier == r1.obs_outlier).all()
        assert (q.kf_pose == q1.kf_pose).all() and (q.pt == q1.pt).all()
        qo, ro = oracle.solve(p)
        _check(p, q, r, qo, ro)
    # nine windows: the >= 8 windows code path (XCD-aware mapping, four pairs per wave)
    ps9 = [ps[i % 3] for i in range(9)]
    ba.upload(ps9); ba.run(); q9, r9 = ba.download()
    for i in range(9):
        assert (q9[i].kf_pose == qs[i % 3].kf_pose).all() or np.abs(q9[i].kf_pose - qs[i % 3].kf_pose).max() < 1e-9
        assert r9[i].its_done == rs[i % 3].its_done


@pytest.mark.parametrize("variant", [abi.VARIANT_PRV_IDP, abi.VARIANT_PRV_XYZ])
def test_window_groups_on_several_streams_give_identical_results(ba, variant):
    """the multi-stream schedule (groups of windows on their own streams, interleaved enqueue; Gauss-Newton and the
    Levenberg-Marquardt state machine) must not change any window: one group vs the default policy vs 3 and 4 groups"""
    ps = [synth.make_window(variant, n_kf=8 + (i % 3), n_fixed=1, n_pt=150 + 10 * (i % 5), n_obs=700 + 40 * (i % 5), seed=80 + i % 5)
          for i in range(37)]
    ba.upload(ps)
    runs = []
    try:
        for streams in (1, 0, 3, 4):   # 0: default policy (two groups for 37 windows)
            ba.lib.vba_debug_set_streams(ba.h, streams)
            ba.run()
            q, r = ba.download()
            runs.append(([x.copy() for x in q], r))   # download() returns the wrapper's own problem objects
    finally:
        ba.lib.vba_debug_set_streams(ba.h, 0)
    q1, r1 = runs[0]
    assert all(r.status == 0 for r in r1)
    for qn, rn in runs[1:]:
        for a, b, ra, rb in zip(q1, qn, r1, rn):
            assert ra.its_done == rb.its_done and ra.status == rb.status == 0
            assert (a.kf_pose == b.kf_pose).all() and (a.pt == b.pt).all() and ra.chi2_vis == rb.chi2_vis


@pytest.mark.parametrize("variant", [abi.VARIANT_PRV_IDP, abi.VARIANT_SE3_XYZ])
def test_batch_solve_streams_fresh_windows_like_upload_run_download(ba, oracle, variant):
    """vba_batch_solve (fresh host arrays in, solved ones out, chunks of the batch on concurrent lanes) gives every window
    bit for bit what upload + run + download of the whole batch gives, whatever the chunking; and the oracle's result"""
    kw = dict(n_fixed=1) if variant == abi.VARIANT_PRV_IDP else dict(n_fixed=2)
    ps = [synth.make_window(variant, n_kf=7 + (i % 4), n_pt=120 + 15 * (i % 6), n_obs=600 + 70 * (i % 6), seed=300 + i % 6, **kw)
          for i in range(29)]
    ba.upload(ps); ba.run(); q0, r0 = ba.download()
    q0 = [x.copy() for x in q0]
    try:
        # chunks of >= 8 windows run the same kernels as the 29-window batch (the kernel choice depends on the number of windows
        # in a call: thresholds 8 / 64 / 256): bit-identical.  Smaller chunks take the few-window kernels, whose sums run in
        # another (fixed) order: equal to rounding.
        for chunk, lanes, exact in ((1000, 1, True), (10, 3, True), (15, 2, True), (4, 2, False), (7, 1, False)):
            ba.lib.vba_debug_set_chunking(ba.h, chunk, lanes)
            q, r = ba.solve_batch(ps)
            for a, b, ra, rb in zip(q0, q, r0, r):
                assert ra.status == rb.status == 0 and ra.its_done == rb.its_done and (ra.obs_outlier == rb.obs_outlier).all()
                if exact:
                    assert ra.chi2_vis == rb.chi2_vis and (ra.obs_chi2 == rb.obs_chi2).all()
                    assert (a.kf_pose == b.kf_pose).all() and (a.pt == b.pt).all() and (a.kf_vel == b.kf_vel).all()
                else:
                    assert abs(ra.chi2_vis - rb.chi2_vis) <= 1e-10 * ra.chi2_vis
                    assert np.abs(a.kf_pose - b.kf_pose).max() < 1e-9 and np.abs(a.pt - b.pt).max() < 1e-9
    finally:
        ba.lib.vba_debug_set_chunking(ba.h, 0, 0)
    for i in range(6):
        qo, ro = oracle.solve(ps[i])
        _check(ps[i], q[i], r[i], qo, ro)
    # the inputs were not touched (solve_batch works on copies), and a bad window fails the call with a message
    bad = ps[3].copy(); bad.obs_kf = bad.obs_kf.copy(); bad.obs_kf[5] = 99
    ba.lib.vba_debug_set_chunking(ba.h, 4, 2)
    try:
        with pytest.raises(RuntimeError, match="vba_batch_solve"):
            ba.solve_batch(ps[:9] + [bad] + ps[:3])
    finally:
        ba.lib.vba_debug_set_chunking(ba.h, 0, 0)


def test_fused_imu_factor_hessian_equals_edge_navstate(ba, oracle):
    """A6 on the GPU: the 30x30 local Hessian + rhs the linearisation kernel builds for one keyframe pair
    (EdgeNavStatePRV + EdgeNavStateBias fused, Huber weights applied) equals J^T (rho' Omega) J / -J^T (rho' Omega) e of the
    oracle's 15-D EdgeNavState (g2otypes.cpp:989-1168) after the P,Phi,V -> P,V,Phi permutation"""
    from test_oracle_units import split_factor, A6_FROM_SPLIT_COLS, A6_FROM_SPLIT_ROWS, _nav
    p = synth.make_window(abi.VARIANT_PRV_IDP, n_kf=7, n_fixed=1, n_pt=120, n_obs=600, seed=61)
    p.kf_bias = p.kf_bias.copy()
    p.kf_bias[:, 6:] = np.random.default_rng(5).normal(0, 1e-3, (p.n_kf, 6))   # non-zero delta biases
    p.its_stage1, p.its_stage2 = 1, 0          # one linearisation at the uploaded state; IMUH then holds its products
    ba.upload([p]); ba.run()
    buf = ba.lib.vba_debug_buf_id(b"IMUH")
    H = np.zeros((p.n_imu, 960))
    assert ba.lib.vba_debug_copy(ba.h, buf, C.c_uint64(0), H.ctypes.data_as(C.c_void_p), C.c_uint64(H.nbytes)) == 0
    pr, pc = np.array(A6_FROM_SPLIT_ROWS), np.array(A6_FROM_SPLIT_COLS)
    pc2 = np.concatenate([pc, 15 + pc])
    checked = 0
    for k in range(p.n_imu):
        i, j = p.imu_kf_i[k], p.imu_kf_j[k]
        # the A6 factor with the split edges' robust weights: rho'(chi2_prv) on the PVR rows, rho'(chi2_bias) on the bias rows
        e_s, Ji_s, Jj_s, Om_s = split_factor(oracle, p, k)
        w_prv = oracle.huber(e_s[:9] @ Om_s[:9, :9] @ e_s[:9], p.huber_prv)[1]
        w_b = oracle.huber(e_s[9:] @ Om_s[9:, 9:] @ e_s[9:], p.huber_bias)[1]
        W6 = (np.diag([w_prv] * 9 + [w_b] * 6) @ Om_s)[pr][:, pr]
        ni, nj = _nav(p, i), _nav(p, j)
        e6 = oracle.edge_navstate_error(ni, nj, p.imu_meas[k], p.g_w)
        J6 = np.hstack(oracle.edge_navstate_jac(ni, nj, p.imu_meas[k], p.g_w, e6))
        H6, b6 = J6.T @ W6 @ J6, -J6.T @ W6 @ e6
        Hg, bg = H[k, :900].reshape(30, 30), H[k, 900:930]
        # GPU local order per keyframe: P Phi V dbg dba (the split order) -> A6 order through the column permutation
        np.testing.assert_allclose(Hg[pc2][:, pc2], H6, rtol=1e-9, atol=1e-9 * np.abs(H6).max())
        np.testing.assert_allclose(bg[pc2], b6, rtol=1e-9, atol=1e-9 * np.abs(b6).max())
        checked += 1
    assert checked == p.n_imu >= 6


@pytest.mark.parametrize("variant", [abi.VARIANT_SE3_XYZ, abi.VARIANT_PRV_XYZ])
def test_xyz_linearisation_fallback_equals_edge_parallel(ba, oracle, variant):
    """XYZ windows are linearised edge-parallel (k_lin_xyz_e); a window with a track longer than 256 observations falls back to a
    thread per landmark (k_lin_xyz).  Both must give the oracle's result, and a batch may mix them."""
    p = synth.make_window(variant, algo=abi.ALGO_LM, n_kf=10, n_fixed=2 if variant == abi.VARIANT_SE3_XYZ else 1, n_pt=300, n_obs=1800, seed=36)
    q1, r1 = ba.solve(p)
    try:
        ba.lib.vba_debug_set_lin_fallback(ba.h, 1)
        q0, r0 = ba.solve(p)
    finally:
        ba.lib.vba_debug_set_lin_fallback(ba.h, 0)
    qo, ro = oracle.solve(p)
    _check(p, q1, r1, qo, ro)
    _check(p, q0, r0, qo, ro)
    assert r0.its_done == r1.its_done and np.abs(q0.kf_pose - q1.kf_pose).max() < 1e-10 and abs(r0.chi2_vis - r1.chi2_vis) <= 1e-10 * r1.chi2_vis


def test_first_form_of_the_factorisation_step_agrees_with_the_dpp_form(ba, oracle):
    """k_chol_step4 eliminates with hand-written v_fmac_f64_dpp row_newbcast instructions and carries the panel rows along; the first
    form of the step (k_chol_step: v_readlane broadcasts, panel solves behind the diagonal tile) stays as its cross-check: both
    against the oracle and against each other (same arithmetic up to the association of l = a / d)."""
    ba.lib.vba_debug_set_chol_step.argtypes = [C.c_void_p, C.c_int32]
    ps = [synth.config_c3_ragged(100 + s) for s in (3, 4)] + [synth.make_window(abi.VARIANT_PRV_XYZ, algo=abi.ALGO_LM, n_kf=12, n_fixed=1, n_pt=500, n_obs=3000, seed=35)]
    new = [ba.solve(p) for p in ps]
    try:
        ba.lib.vba_debug_set_chol_step(ba.h, 1)
        old = [ba.solve(p) for p in ps]
    finally:
        ba.lib.vba_debug_set_chol_step(ba.h, 0)
    for p, (q, r), (q1, r1) in zip(ps, new, old):
        qo, ro = oracle.solve(p)
        _check(p, q, r, qo, ro)
        _check(p, q1, r1, qo, ro)
        assert r.its_done == r1.its_done and (r.obs_outlier == r1.obs_outlier).all()
        np.testing.assert_allclose(r.chi2_trace, r1.chi2_trace, rtol=1e-9)
        assert np.abs(q.kf_pose - q1.kf_pose).max() < 1e-9


def test_rerun_is_bit_reproducible(ba):
    p = synth.make_window(abi.VARIANT_PRV_IDP, n_kf=10, n_fixed=1, n_pt=400, n_obs=2000, seed=7)
    ba.upload([p]); ba.run(); q1, r1 = ba.download()
    a = q1[0].kf_pose.copy(), q1[0].pt.copy(), r1[0].chi2_vis
    ba.run(); q2, r2 = ba.download()
    assert (a[0] == q2[0].kf_pose).all() and (a[1] == q2[0].pt).all() and a[2] == r2[0].chi2_vis


def test_empty_and_degenerate_inputs(ba):
    # a landmark with no edges, and a window without landmarks at all (IMU chain only)
    p = synth.make_window(abi.VARIANT_PRV_IDP, n_kf=6, n_fixed=1, n_pt=60, n_obs=240, seed=31)
    import copy
    q = p.copy()
    q.pt_obs_begin = p.pt_obs_begin.copy()
    # drop all edges of landmark 0 by making its CSR row empty (shift is not needed: rows may leave gaps? no:
    # rebuild compactly)
    keep = np.ones(p.n_obs, dtype=bool); keep[p.pt_obs_begin[0]:p.pt_obs_begin[1]] = False
    cnt = np.diff(p.pt_obs_begin); cnt[0] = 0
    q.pt_obs_begin = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    q.obs_kf = p.obs_kf[keep].copy(); q.obs_uv = p.obs_uv[keep].copy(); q.obs_w = p.obs_w[keep].copy()
    qq, r = ba.solve(q)
    assert r.status == 0 and qq.pt[0, 0] == p.pt[0, 0]
    # error path: duplicate observation from one keyframe is rejected with a message, not a crash
    bad = p.copy(); bad.obs_kf = p.obs_kf.copy()
    o0 = p.pt_obs_begin[0]
    bad.obs_kf[o0 + 1] = bad.obs_kf[o0]
    with pytest.raises(RuntimeError, match="observed twice"):
        ba.solve(bad)


def test_c4_full_size_properties(ba, oracle):
    """BASELINE configs[3]: 200 KF / 50k landmarks / 500k EdgePRIDP + IMU chain.  (The comparison with the oracle's own solve of
    this window is tests/test_gpu_protocol_edges.py::test_c4_full_size_matches_oracle.)  Size-independent properties, each
    against the oracle's residual functions:
      * the first traced value is the robust chi2 of the uploaded state;
      * every edge the backend keeps as an inlier carries the chi2 the oracle computes at the returned state;
      * the IMU chi2 sums match the oracle's at the returned state;
      * Gauss-Newton made progress: robust chi2 decreases over stage 1, relative keyframe geometry improves."""
    p = synth.config_c4()
    q, r = ba.solve(p)
    assert r.status == 0 and r.its_done[0] >= 1
    rob0 = oracle.evaluate(p, robust_vis=True)[0]
    assert abs(r.chi2_trace[0] - rob0) <= 1e-9 * rob0
    _, _, prv, bias, ch, dep = oracle.evaluate(q, robust_vis=False)
    assert abs(r.chi2_prv - prv) <= 1e-6 * prv and abs(r.chi2_bias - bias) <= 1e-6 * max(bias, 1e-9)
    inl = r.obs_outlier == 0
    np.testing.assert_allclose(r.obs_chi2[inl], ch[inl], rtol=1e-6, atol=1e-9)
    assert (ch[inl] <= p.chi2_th).all() and (dep[inl] > p.depth_min).all()
    tr = r.chi2_trace[: r.its_done[0] + 1]
    assert (np.diff(tr) < 1e-6 * tr[0]).all()
    # one fixed keyframe anchors a 50 s chain, so absolute error at the far end is drift; the local geometry
    # (relative translation of consecutive keyframes) is what local BA improves
    gt = p.truth["pose"][: p.n_kf_free, :3]
    rel = lambda a: np.diff(a, axis=0)
    e0 = np.abs(rel(p.kf_pose[: p.n_kf_free, :3]) - rel(gt)).mean()
    e1 = np.abs(rel(q.kf_pose[: p.n_kf_free, :3]) - rel(gt)).mean()
    assert e1 < 0.5 * e0, (e0, e1)
    planted = p.truth["is_outlier"]
    assert (r.obs_outlier.astype(bool) & planted).sum() >= 0.9 * planted.sum()


def test_c4_noise_free_terminates_at_truth(ba):
    p = synth.make_window(abi.VARIANT_PRV_IDP, n_kf=200, n_fixed=1, n_pt=20000, n_obs=200000, seed=4, noise=False)
    q, r = ba.solve(p)
    assert r.its_done == (1, 1) and r.n_outliers == 0 and r.status == 0
    assert np.abs(q.kf_pose - p.truth["pose"]).max() < 1e-6


def test_on_device_preintegration_matches_oracle_and_numpy(ba, oracle):
    """A15: IMUPreintegrator::update on the GPU vs the C oracle (sample by sample) and the numpy twin."""
    rng = np.random.default_rng(9)
    lens = [50, 51, 1, 37, 0, 200]
    sb = np.concatenate([[0], np.cumsum(lens)])
    n = int(sb[-1])
    gyr = rng.normal(0, 0.3, (n, 3)); acc = rng.normal(0, 1.0, (n, 3)) + [0, 0, 9.8]
    dt = np.full(n, 0.005); dt[0] = 0.0   # KeyFrame::ComputePreInt's leading (sample, t_imu0 - t_prevKF) call
    meas, cov, info = ba.preintegrate(sb, gyr, acc, dt)
    for e, L in enumerate(lens):
        s0, s1 = sb[e], sb[e + 1]
        m_c, c_c = oracle.preint(gyr[s0:s1], acc[s0:s1], dt[s0:s1])
        np.testing.assert_allclose(meas[e], m_c, rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(cov[e], c_c, rtol=1e-10, atol=1e-24)
        if L > 1:
            np.testing.assert_allclose(info[e], oracle.prv_information(c_c), rtol=1e-6)
    m_np, c_np = synth.preintegrate(gyr[None, :50], acc[None, :50], dt[None, :50])
    np.testing.assert_allclose(meas[0], m_np[0], rtol=1e-11, atol=1e-13)
    # the information it returns is what vba_problem.imu_info_prv expects
    p = synth.make_window(abi.VARIANT_PRV_IDP, n_kf=6, n_fixed=1, n_pt=60, n_obs=240, seed=31)
    assert info.shape[1:] == (9, 9) and p.imu_info_prv.shape[1] == 81


# ---- global bundle adjustment: one optimize(n), optional kernels, per-vertex fixed flags ----
def _gba(variant, robust, n_kf=12, n_pt=500, n_obs=3000, seed=50, its=20, outlier_frac=0.02):
    p = synth.make_window(variant, algo=abi.ALGO_LM, n_kf=n_kf, n_fixed=0 if variant != abi.VARIANT_SE3_XYZ else 1,
                          n_pt=n_pt, n_obs=n_obs, seed=seed, outlier_frac=outlier_frac)
    p.protocol, p.robust, p.its_stage1, p.its_stage2 = abi.PROTO_SINGLE, robust, its, 0
    p.huber_vis = float(np.float32(np.sqrt(5.99)))     # thHuber2D of BundleAdjustment, src/Optimizer.cpp:3414
    if variant != abi.VARIANT_SE3_XYZ:                 # keyframe 0: PR and Bias fixed, V free (:667-685)
        p.kf_fix = np.zeros(p.n_kf, np.uint8); p.kf_fix[0] = 0b101
    return p


@pytest.mark.parametrize("variant,robust,seed", [
    (abi.VARIANT_SE3_XYZ, 1, 50), (abi.VARIANT_SE3_XYZ, 0, 51),
    (abi.VARIANT_PRV_XYZ, 1, 52), (abi.VARIANT_PRV_XYZ, 0, 53),
])
def test_global_ba_protocol_matches_oracle(ba, oracle, variant, robust, seed):
    """BundleAdjustment (src/Optimizer.cpp:3377-3607) and GlobalBundleAdjustmentNavStatePRV (:629-933)."""
    p = _gba(variant, robust, seed=seed)
    q, r = ba.solve(p)
    qo, ro = oracle.solve(p)
    _check(p, q, r, qo, ro)
    assert r.its_done[1] == 0 and r.n_outliers == 0
    assert abs(r.lambda_final - ro.lambda_final) <= 1e-6 * ro.lambda_final
    if variant == abi.VARIANT_PRV_XYZ:
        assert (q.kf_pose[0] == p.kf_pose[0]).all() and (q.kf_bias[0] == p.kf_bias[0]).all()
        assert (q.kf_vel[0] != p.kf_vel[0]).any()
        np.testing.assert_allclose(q.kf_bias, qo.kf_bias, atol=1e-7)


def test_global_ba_map_scale_properties(ba, oracle):
    """A 150-keyframe map (n_p = 2250; against the oracle's own solve: test_gpu_protocol_edges.py): size-independent properties --
    the robust cost the solver reports equals the oracle's residual evaluation at the returned state, it went down,
    the gauge keyframe did not move, reruns are bit-identical."""
    p = _gba(abi.VARIANT_PRV_XYZ, 1, n_kf=150, n_pt=12000, n_obs=80000, seed=60, its=10)
    q, r = ba.solve(p)
    q2, r2 = ba.solve(p)
    assert r.status == 0 and r.its_done[0] >= 2 and r.its_done[1] == 0
    assert (q.kf_pose == q2.kf_pose).all() and r.chi2_vis == r2.chi2_vis
    out = oracle.evaluate(q, robust_vis=1)
    assert abs(out[1] - r.chi2_vis) <= 1e-9 * r.chi2_vis and abs(out[2] - r.chi2_prv) <= 1e-7 * max(r.chi2_prv, 1e-9)
    assert abs(out[0] - r.chi2_trace[-1]) <= 1e-9 * out[0]
    assert r.chi2_trace[-1] < 0.5 * r.chi2_trace[0]
    assert (q.kf_pose[0] == p.kf_pose[0]).all()


@pytest.mark.parametrize("variant,nwin", [(abi.VARIANT_PRV_IDP, 66), (abi.VARIANT_PRV_IDP, 388), (abi.VARIANT_PRV_XYZ, 388),
                                          (abi.VARIANT_SE3_XYZ, 66), (abi.VARIANT_SE3_XYZ, 388)])
def test_large_batches_switch_factorisation_kernels(ba, oracle, variant, nwin):
    """>= 64 windows: the many-window launch organisation (device mirror of the stop word, IMU factors in launches of their own);
    >= 256 windows: left-looking tile kernels (k_chol_diag_ll2 / k_chol_panel_ll) instead of the fused right-looking step
    (k_chol_step4).  Same results as the single-window path up to rounding, and the oracle's bars."""
    kw = [dict(n_kf=8, n_pt=200, n_obs=1000), dict(n_kf=13, n_pt=400, n_obs=2200), dict(n_kf=23, n_pt=900, n_obs=5200),
          dict(n_kf=6, n_pt=60, n_obs=300)]
    algo = abi.ALGO_GN if variant == abi.VARIANT_PRV_IDP else abi.ALGO_LM
    ps = [synth.make_window(variant, algo=algo, n_fixed=2 if variant == abi.VARIANT_SE3_XYZ else 1, seed=80 + i, **k) for i, k in enumerate(kw)]
    batch = [ps[i % 4] for i in range(nwin)]
    ba.upload(batch); ba.run(); qs, rs = ba.download()
    for i, p in enumerate(ps):
        q1, r1 = ba.solve(p)
        q, r = qs[i], rs[i]
        assert r.status == 0 and r.its_done == r1.its_done and (r.obs_outlier == r1.obs_outlier).all()
        np.testing.assert_allclose(r.chi2_trace, r1.chi2_trace, rtol=1e-9)
        assert np.abs(q.kf_pose - q1.kf_pose).max() < 1e-9 and np.abs(q.pt - q1.pt).max() < 1e-8
        qo, ro = oracle.solve(p)
        _check(p, q, r, qo, ro)
    for i in range(4, nwin):   # twins inside the batch agree bit for bit
        assert (qs[i].kf_pose == qs[i % 4].kf_pose).all() and rs[i].chi2_vis == rs[i % 4].chi2_vis


def test_chain_columns_in_one_launch_agree_with_one_launch_per_column(ba, oracle):
    """The V/Bias block columns of the IMU chain ("chain columns": row J of the factor has no tile left of (J, J-1)) are factored by
    one launch per factorisation instead of one per column (csrc/vba_chain.h): k_chol_chain_rows + k_chol_chain_upd for a handful
    of windows, k_chol_chain_diag + k_chol_chain_panel in the left-looking regime.  Same elimination order, sums in another fixed
    order: both forms against the oracle and against each other (to rounding), full-size ragged C3 windows (40..60 keyframes, 11..16
    chain columns), alone, five in one call (ragged: different chain lengths in one launch), and a window whose IMU chain is broken
    in the middle (two keyframes without an IMU factor between them: a chain column without its sub-diagonal tile)."""
    ps = [synth.config_c3_ragged(100 + s) for s in (1, 2, 5, 7, 9)]
    broken = synth.config_c3(seed=11, n_kf=30, n_pt=1500, n_obs=9000)
    keep = np.array([k for k in range(broken.n_imu) if k != broken.n_imu // 2])
    broken = broken.copy()
    broken.imu_kf_i, broken.imu_kf_j = broken.imu_kf_i[keep].copy(), broken.imu_kf_j[keep].copy()
    broken.imu_meas, broken.imu_info_prv = broken.imu_meas[keep].copy(), broken.imu_info_prv[keep].copy()
    ps.append(broken)
    ba.lib.vba_debug_set_chain.argtypes = [C.c_void_p, C.c_int32]
    res = {}
    try:
        for ll in (0, 1):
            ba.lib.vba_debug_set_ll_min(ba.h, 1 if ll else 0)
            for chain in (1, 0):
                ba.lib.vba_debug_set_chain(ba.h, chain)
                one, launches = [], []
                for p in ps:
                    one.append(ba.solve(p))
                    launches.append(ba.get_profile()["kernel_launches"])
                ba.upload(ps[:5]); ba.run(); qs, rs = ba.download()
                res[ll, chain] = (one, list(zip(qs, rs)))
                res[ll, chain, "launches"] = launches
    finally:
        ba.lib.vba_debug_set_chain(ba.h, 1)
        ba.lib.vba_debug_set_ll_min(ba.h, 0)
    for ll in (0, 1):   # the chain kernels really ran: 2 launches instead of nc (right-looking) / 2 nc (left-looking) per factorisation
        saved = [a - b for a, b in zip(res[ll, 0, "launches"], res[ll, 1, "launches"])]   # (a window in keyframe order has no chain columns: 0)
        assert sum(1 for x in saved if x >= 50) >= 3 and min(saved) >= 0, saved
    for i, p in enumerate(ps):
        qo, ro = oracle.solve(p)
        for ll in (0, 1):
            (q, r), (q0, r0) = res[ll, 1][0][i], res[ll, 0][0][i]
            _check(p, q, r, qo, ro)
            _check(p, q0, r0, qo, ro)
            assert r.its_done == r0.its_done and (r.obs_outlier == r0.obs_outlier).all()
            np.testing.assert_allclose(r.chi2_trace, r0.chi2_trace, rtol=1e-8)
            assert np.abs(q.kf_pose - q0.kf_pose).max() < 1e-8
            if i < 5:   # inside the five-window call
                qb, rb = res[ll, 1][1][i]
                _check(p, qb, rb, qo, ro)
                if ll:  # left-looking: the kernels do not depend on the window count -- bit for bit
                    assert (qb.kf_pose == q.kf_pose).all() and rb.chi2_vis == r.chi2_vis


def test_chain_kernels_in_a_batch_of_a_dozen_windows(ba, oracle):
    """Up to 64 windows a right-looking batch walks its chain columns in one launch (VBA_CHAIN_RL_MAX): twelve ragged C3 windows in one
    call -- chains of different lengths and splits, the four-pairs-per-wave Schur gather and the XCD-aware mapping of >= 8 windows --
    against the oracle and against the same call with one launch per block column."""
    ps = [synth.config_c3_ragged(300 + s) for s in range(12)]
    ba.lib.vba_debug_set_chain.argtypes = [C.c_void_p, C.c_int32]
    got = {}
    try:
        for chain in (1, 0):
            ba.lib.vba_debug_set_chain(ba.h, chain)
            ba.upload(ps); ba.run()
            got[chain] = (ba.download(), ba.get_profile()["kernel_launches"])
    finally:
        ba.lib.vba_debug_set_chain(ba.h, 1)
    assert got[0][1] - got[1][1] >= 50, (got[0][1], got[1][1])     # the chain kernels ran
    for i, p in enumerate(ps):
        qo, ro = oracle.solve(p)
        q, r = got[1][0][0][i], got[1][0][1][i]
        q0, r0 = got[0][0][0][i], got[0][0][1][i]
        _check(p, q, r, qo, ro)
        assert r.its_done == r0.its_done and (r.obs_outlier == r0.obs_outlier).all()
        assert np.abs(q.kf_pose - q0.kf_pose).max() < 1e-8


def test_two_sided_vbias_order_agrees_with_the_one_sided_order(ba, oracle, monkeypatch):
    """Order 2 (vba_host_structure.h): the V/Bias blocks as two chains that meet in the middle, pads between the parts of the reduced
    system.  Fewer tile products than order 0 on every full-size window, and k_chol_chain_rows walks the two chains side by side (512
    threads).  Against the oracle and against order 0 / 1 (VBA_ONE_CHAIN, read at every upload) to rounding, in both regimes: full-size
    ragged C3 windows, a window of 13 keyframes (the smallest the order is offered to have 12 free keyframes), a window whose IMU chain is
    broken next to the split, and a ragged five-window call (chains of different lengths and splits in one launch)."""
    ps = [synth.config_c3_ragged(100 + s) for s in (1, 2, 5, 7, 9)]
    ps.append(synth.config_c3(seed=12, n_kf=13, n_pt=600, n_obs=3600))
    broken = synth.config_c3(seed=11, n_kf=30, n_pt=1500, n_obs=9000).copy()
    keep = np.array([k for k in range(broken.n_imu) if k != broken.n_imu // 2 - 1])
    broken.imu_kf_i, broken.imu_kf_j = broken.imu_kf_i[keep].copy(), broken.imu_kf_j[keep].copy()
    broken.imu_meas, broken.imu_info_prv = broken.imu_meas[keep].copy(), broken.imu_info_prv[keep].copy()
    ps.append(broken)
    res, orders = {}, {}
    try:
        for ll in (0, 1):
            ba.lib.vba_debug_set_ll_min(ba.h, 1 if ll else 0)
            for one in (0, 1):
                if one: monkeypatch.setenv("VBA_ONE_CHAIN", "1")
                else: monkeypatch.delenv("VBA_ONE_CHAIN", raising=False)
                single = [ba.solve(p) for p in ps]
                ba.upload(ps[:5])
                tp = np.zeros((5, 5), dtype=np.int64)
                for i in range(5):
                    assert ba.lib.vba_debug_tile_products(ba.h, i, tp[i].ctypes.data_as(C.c_void_p)) == 0
                orders[ll, one] = tp
                ba.run(); qs, rs = ba.download()
                res[ll, one] = (single, list(zip(qs, rs)))
    finally:
        monkeypatch.delenv("VBA_ONE_CHAIN", raising=False)
        ba.lib.vba_debug_set_ll_min(ba.h, 0)
    for ll in (0, 1):
        tp2, tp1 = orders[ll, 0], orders[ll, 1]
        assert (tp1[:, 2] != 2).all() and (tp1[:, 4] == -1).all()          # the switch keeps order 2 out
        assert (tp2[:, 2] == 2).sum() >= 3 and (tp2[:, 4] > 0).all()       # offered to every window, taken by most
        assert (tp2[tp2[:, 2] == 2, 4] < tp2[tp2[:, 2] == 2, 0]).all()     # ... where it has fewer tile products than order 0
    for i, p in enumerate(ps):
        qo, ro = oracle.solve(p)
        for ll in (0, 1):
            (q, r), (q1, r1) = res[ll, 0][0][i], res[ll, 1][0][i]
            _check(p, q, r, qo, ro)
            _check(p, q1, r1, qo, ro)
            assert r.its_done == r1.its_done and (r.obs_outlier == r1.obs_outlier).all()
            np.testing.assert_allclose(r.chi2_trace, r1.chi2_trace, rtol=1e-8)
            assert np.abs(q.kf_pose - q1.kf_pose).max() < 1e-8
            if i < 5:
                qb, rb = res[ll, 0][1][i]
                _check(p, qb, rb, qo, ro)


def test_left_looking_kernels_at_full_window_size(ba, oracle):
    """The left-looking factorisation (tile-packed factor, operands loaded straight into the MFMA registers) is what batches of
    >= 256 windows run; here it is forced on full-size ragged C3 windows (40..60 keyframes: 19..28 block columns) and on a
    full-size C2 window, one window at a time, against the oracle and against the right-looking kernels."""
    ps = [synth.config_c3_ragged(100 + s) for s in (1, 2, 5)] + [synth.config_c2()]
    ref = [ba.solve(p) for p in ps]
    try:
        ba.lib.vba_debug_set_ll_min(ba.h, 1)
        got = [ba.solve(p) for p in ps]
        ba.upload(ps[:3]); ba.run(); qs, rs = ba.download()    # ragged windows in one batch
    finally:
        ba.lib.vba_debug_set_ll_min(ba.h, 0)
    for i, p in enumerate(ps):
        (q, r), (q1, r1) = got[i], ref[i]
        qo, ro = oracle.solve(p)
        _check(p, q, r, qo, ro)
        assert r.its_done == r1.its_done and (r.obs_outlier == r1.obs_outlier).all()
        np.testing.assert_allclose(r.chi2_trace, r1.chi2_trace, rtol=1e-8)
        if i < 3:
            assert (qs[i].kf_pose == q.kf_pose).all() and rs[i].chi2_vis == r.chi2_vis   # batch == single solve, bit for bit


def test_structure_row_walk_in_lanes_equals_the_lds_walk(ba):
    """Windows of at most 64 keyframes build their per-pair item lists with the counts of a keyframe row held in lanes
    (st_row_body1); VBA_ST_ROW_LDS keeps them in LDS as windows of more keyframes do (st_row_body).  Same ballots, same ranks: the
    lists -- and with them every summation order of the solve -- must be identical, so the results are equal bit for bit."""
    ps = [synth.config_c3_ragged(300 + s) for s in range(3)] + [synth.config_c3s(5),
          synth.make_window(abi.VARIANT_SE3_XYZ, algo=abi.ALGO_LM, n_kf=14, n_fixed=2, n_pt=700, n_obs=4200, seed=77)]
    fast = [ba.solve(p) for p in ps[:3]] + [ba.solve(p) for p in ps[3:]]
    ba.upload(ps[:3]); ba.run(); qb, rb = ba.download()
    qb = [q.copy() for q in qb]
    os.environ["VBA_ST_ROW_LDS"] = "1"
    try:
        slow = [ba.solve(p) for p in ps]
        ba.upload(ps[:3]); ba.run(); qc, rc = ba.download()
    finally:
        del os.environ["VBA_ST_ROW_LDS"]
    for (q, r), (q1, r1) in zip(fast, slow):
        assert r.its_done == r1.its_done and r.chi2_vis == r1.chi2_vis and r.n_outliers == r1.n_outliers
        assert np.array_equal(q.kf_pose, q1.kf_pose) and np.array_equal(q.pt, q1.pt)
    for q, q1, r, r1 in zip(qb, qc, rb, rc):
        assert r.chi2_vis == r1.chi2_vis and np.array_equal(q.kf_pose, q1.kf_pose)


@pytest.mark.parametrize("make", [
    lambda: synth.config_c3_ragged(411, landmark_order="caller"),
    lambda: synth.config_c3s(412, landmark_order="caller"),                       # fixed reference keyframes inside the runs
    lambda: synth.make_window(abi.VARIANT_PRV_IDP, n_kf=9, n_fixed=2, n_pt=300, n_obs=1500, seed=413, landmark_order="caller"),
    lambda: synth.make_window(abi.VARIANT_PRV_IDP, n_kf=70, n_fixed=1, n_pt=900, n_obs=5400, seed=414, landmark_order="caller"),   # two mask words
], ids=["c3_ragged", "c3s", "small", "70kf"])
def test_landmarks_in_the_callers_order_match_oracle(ba, oracle, make):
    """The reference's caller hands landmarks over grouped by the first local keyframe that observes them (lLocalMapPoints is filled
    keyframe by keyframe, src/Optimizer.cpp:59-78).  k_lin2 then sums the reference-keyframe terms of a workgroup over RUNS of
    landmarks with one reference keyframe (one record per run; the other windows of this suite come in random order: runs of one):
    alone and inside a batch, against the oracle (which does not care about the order)."""
    p = make()
    runs = 1 + int((np.diff(p.pt_ref_kf) != 0).sum())
    assert runs < 0.8 * p.n_pt                  # runs of several landmarks (long ones in the sliding windows, short ones with scattered tracks)
    q, r = ba.solve(p)
    qo, ro = oracle.solve(p)
    _check(p, q, r, qo, ro)
    if p.n_pt <= 1000:
        ps = [p, synth.make_window(abi.VARIANT_PRV_IDP, n_kf=9, n_fixed=2, n_pt=300, n_obs=1500, seed=415)] * 5   # mixed with random-order windows
        ba.upload(ps); ba.run(); qs, rs = ba.download()
        for qq, rr in zip(qs[0::2], rs[0::2]):
            assert rr.its_done == r.its_done and np.abs(qq.kf_pose - q.kf_pose).max() < 1e-9
