"""The protocol paths a live system hits that a clean solve never does, GPU backend against the CPU oracle through the C-ABI:

* full-size BASELINE configs[3] (C4, 94 block columns) and the 150-keyframe global BA against the oracle -- direct solver and PCG;
* LocalMapping::InterruptBA in the MIDDLE of a solve: g2o polls forceStopFlag before every iteration
  (/root/reference/Thirdparty/g2o/g2o/core/sparse_optimizer.cpp:376, optimization_algorithm_levenberg.cpp:149) and LocalBAPRVIDP
  checks it once more between optimize(5) and optimize(10) (/root/reference/src/Optimizer.cpp:462-470: stage 2 is skipped, the
  stage-1 state is still written back, the erase list still built, :496-517).  Both sides count a window's polls (backend:
  poll_stop / vba_debug_set_stop_after, oracle: stop_now / vba_oracle_solve_ex) so that the flag can be raised at a chosen poll;
  one more test raises the REAL flag from another host thread while a batch runs;
* solver failure: a reduced system with an exactly zero pivot (linear_solver_eigen.h:105-111 -> Fail) under Gauss-Newton ->
  VBA_SOLVER_FAILED, step dropped; under Levenberg-Marquardt a failed trial is a rejected trial (levenberg.cpp:126-127).
"""
import ctypes as C
import threading
import time

import numpy as np
import pytest

from mc_slam_amd import abi, synth, backend
from test_gpu_parity import _check, _gba

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ba():
    b = backend.LocalBA(0, hooks=True)
    b.lib.vba_debug_set_stop_after.argtypes = [C.c_void_p, C.c_int32]
    yield b
    b.lib.vba_debug_set_stop_after(b.h, -1)
    b.close()


def _with_pcg(p):
    q = p.copy()
    q.solver = abi.SOLVER_PCG
    return q


# ---- (1) the big configurations against the oracle itself ----
def test_c4_full_size_matches_oracle(ba, oracle):
    """BASELINE configs[3]: 200 KF / 50k landmarks / 500k EdgePRIDP + IMU chain, n_p = 2985 (94 block columns).  The oracle's
    V/Bias-first LDL^T solves it in ~10 s.  Iteration counts, outlier bitmap, chi2 1e-4, translations 1e-6 m -- LDL^T and PCG."""
    p = synth.config_c4()
    qo, ro = oracle.solve(p, solver_mode=1)
    q, r = ba.solve(p)
    _check(p, q, r, qo, ro)
    qp, rp = ba.solve(_with_pcg(p))
    assert rp.lin_iterations > sum(rp.its_done)
    _check(p, qp, rp, qo, ro, trace_rtol=1e-6)


def test_global_ba_150kf_matches_oracle(ba, oracle):
    """GlobalBundleAdjustmentNavStatePRV at map scale (150 KF, n_p = 2250, LM optimize(10)) against the oracle, LDL^T and PCG"""
    p = _gba(abi.VARIANT_PRV_XYZ, 1, n_kf=150, n_pt=12000, n_obs=80000, seed=60, its=10)
    qo, ro = oracle.solve(p)
    q, r = ba.solve(p)
    _check(p, q, r, qo, ro)
    assert abs(r.lambda_final - ro.lambda_final) <= 1e-6 * ro.lambda_final
    qp, rp = ba.solve(_with_pcg(p))
    _check(p, qp, rp, qo, ro, trace_rtol=1e-6)


# ---- (2) abort in the middle of a solve ----
def _solve_with_stop_after(ba, p, n):
    ba.lib.vba_debug_set_stop_after(ba.h, n)
    try:
        return ba.solve(p)
    finally:
        ba.lib.vba_debug_set_stop_after(ba.h, -1)


def _check_abort(p, q, r, qo, ro):
    _check(p, q, r, qo, ro)
    assert r.n_outliers == ro.n_outliers
    if ro.status == 1:   # VBA_ABORTED_AFTER_STAGE1
        assert r.its_done[1] == 0


@pytest.mark.parametrize("n", [0, 1, 2, 3, 4, 5, 6, 7, 8])
def test_gauss_newton_abort_at_every_poll_matches_oracle(ba, oracle, n):
    """LocalBAPRVIDP, the flag raised at the window's n-th terminate() poll: n <= 4 stops optimize(5) early (status 1, stage 2
    skipped, state = the stage-1 state, erase list from that state), n = 5 is the bDoMore check after a complete stage 1, n >= 6
    stops optimize(10) (status 0: the reference does not tell that case apart)"""
    p = synth.make_window(abi.VARIANT_PRV_IDP, n_kf=8, n_fixed=1, n_pt=200, n_obs=900, seed=40)
    qo, ro = oracle.solve(p, stop_after=n)
    q, r = _solve_with_stop_after(ba, p, n)
    _check_abort(p, q, r, qo, ro)
    full = oracle.solve(p)[1]
    assert full.its_done == (5, 3)           # the window really runs 5 + 3 when left alone: the hook is what cut it short
    if n <= 5:
        assert r.status == 1 and r.its_done == (min(n, 5), 0)
        assert (q.kf_pose[:p.n_kf_free] != p.kf_pose[:p.n_kf_free]).any() == (n > 0)   # stage-1 progress is written back
    else:
        assert r.status == 0 and r.its_done == (5, min(n - 6, 3))


@pytest.mark.parametrize("variant,kw", [
    (abi.VARIANT_SE3_XYZ, dict(n_kf=10, n_fixed=2, n_pt=300, n_obs=1800, seed=36)),
    (abi.VARIANT_PRV_XYZ, dict(n_kf=10, n_fixed=1, n_pt=300, n_obs=1800, seed=37)),
])
def test_levenberg_abort_at_every_poll_matches_oracle(ba, oracle, variant, kw):
    """LocalBundleAdjustment (LM): polls before every outer iteration and inside the trial loop after a rejected step"""
    p = synth.make_window(variant, algo=abi.ALGO_LM, **kw)
    seen = set()
    for n in range(0, 14):
        qo, ro = oracle.solve(p, stop_after=n)
        q, r = _solve_with_stop_after(ba, p, n)
        _check_abort(p, q, r, qo, ro)
        if ro.lambda_final > 0:
            assert abs(r.lambda_final - ro.lambda_final) <= 1e-6 * ro.lambda_final
        else:
            assert r.lambda_final == 0.0
        seen.add((r.status, r.its_done))
    assert any(s == 1 for s, _ in seen) and any(s == 0 and it[1] > 0 for s, it in seen)


@pytest.mark.parametrize("nwin", [11, 70])
def test_abort_in_a_batch_where_only_some_windows_are_past_stage_one(ba, oracle, nwin):
    """Windows of one batch reach their polls at different launches: with the flag raised at poll 3 a noise-free window (its
    5-iteration budget ends after one iteration each) has finished BOTH stages before its third poll, a noisy one is cut inside
    stage 1.  11 windows: each reads the pinned word itself; 70: through the per-launch device mirror (k_poll_stop)."""
    noisy = [synth.make_window(abi.VARIANT_PRV_IDP, n_kf=8 + (i % 3), n_fixed=1, n_pt=180 + 20 * i, n_obs=800 + 100 * i, seed=140 + i) for i in range(3)]
    clean = synth.make_window(abi.VARIANT_PRV_IDP, n_kf=8, n_fixed=1, n_pt=200, n_obs=900, seed=41, noise=False)
    distinct = noisy + [clean]
    ps = [distinct[i % 4] for i in range(nwin)]
    ba.upload(ps)
    ba.lib.vba_debug_set_stop_after(ba.h, 3)
    try:
        ba.run()
    finally:
        ba.lib.vba_debug_set_stop_after(ba.h, -1)
    qs, rs = ba.download()
    qs = [x.copy() for x in qs]
    ref = [oracle.solve(p, stop_after=3) for p in distinct]
    for i in range(nwin):
        _check_abort(ps[i], qs[i], rs[i], *ref[i % 4])
    assert [r.status for r in rs[:4]] == [1, 1, 1, 0] and rs[3].its_done == (1, 1) and rs[0].its_done == (3, 0)
    # the same batch without the hook afterwards: nothing of the aborted run lingers
    ba.run()
    q2, r2 = ba.download()
    for i in range(4):
        _check(ps[i], q2[i], r2[i], *oracle.solve(distinct[i]))


def test_real_flag_raised_from_another_thread_mid_run(ba, oracle):
    """The asynchronous path itself: InterruptBA sets *pbStopFlag from the Tracking thread while LocalMapping is inside the solve
    (/root/reference/src/LocalMapping.cpp:1769-1772).  The host side of vba_batch_run forwards the caller's flag into the pinned
    word while it enqueues and while it waits; every window must come back as the oracle's result for SOME poll count."""
    distinct = [synth.make_window(abi.VARIANT_PRV_IDP, n_kf=10 + i, n_fixed=1, n_pt=400 + 50 * i, n_obs=2000 + 300 * i, seed=150 + i) for i in range(3)]
    table = []
    for p in distinct:
        row = {}
        for n in list(range(0, 17)) + [-1]:
            qo, ro = oracle.solve(p, stop_after=n)
            row.setdefault((ro.status, ro.its_done), (qo, ro))
        table.append(row)
    ps = [distinct[i % 3] for i in range(600)]
    ba.upload(ps)
    ba.run()
    _, r_full = ba.download()
    assert all(r.status == 0 and r.its_done[1] >= 1 for r in r_full[:3])
    cut = False
    for delay in (0.004, 0.010, 0.020, 0.002, 0.040):
        flag = C.c_int(0)
        th = threading.Thread(target=lambda: (time.sleep(delay), setattr(flag, "value", 1)))
        th.start()
        ba.run(stop=flag)
        th.join()
        qs, rs = ba.download()
        for i, (q, r) in enumerate(zip(qs, rs)):
            key = (r.status, r.its_done)
            assert key in table[i % 3], (i, key, sorted(table[i % 3]))
            qo, ro = table[i % 3][key]
            _check(ps[i], q, r, qo, ro)
        if any(r.its_done != r_full[i % 3].its_done for i, r in enumerate(rs)):
            cut = True
            break
    assert cut, "the flag never arrived while the batch was running"


# ---- (3) solver failure ----
def _zero_pivot_window(p, a):
    """keyframe a keeps its observations but with zero information, is no landmark's reference keyframe and no IMU edge touches it:
    its PR vertex is in the active set (level-0 edges) with an exactly zero row in H -> the reduced system has a zero pivot"""
    assert not (p.pt_ref_kf == a).any()
    q = p.copy()
    keep = ~((p.imu_kf_i == a) | (p.imu_kf_j == a))
    q.imu_kf_i, q.imu_kf_j = p.imu_kf_i[keep].copy(), p.imu_kf_j[keep].copy()
    q.imu_meas, q.imu_info_prv = p.imu_meas[keep].copy(), p.imu_info_prv[keep].copy()
    q.obs_w = p.obs_w.copy()
    q.obs_w[p.obs_kf == a] = 0.0
    return q


def test_gauss_newton_zero_pivot_reports_solver_failed_and_drops_the_step(ba, oracle):
    p = synth.make_window(abi.VARIANT_PRV_IDP, n_kf=8, n_fixed=1, n_pt=200, n_obs=900, seed=40)
    bad = _zero_pivot_window(p, 6)
    qo, ro = oracle.solve(bad)
    assert ro.status == -2 and ro.its_done == (1, 1)
    q, r = ba.solve(bad)
    _check(bad, q, r, qo, ro)
    assert r.status == -2
    # the step of a failed solve is never applied (documented deviation: g2o applies whatever x held before)
    assert (q.kf_pose == bad.kf_pose).all() and (q.pt == bad.pt).all() and (q.kf_vel == bad.kf_vel).all()
    assert r.chi2_trace[0] == r.chi2_trace[1]
    # one bad window inside a batch does not disturb its neighbours: they come out bit for bit as in a batch without it
    good = [synth.make_window(abi.VARIANT_PRV_IDP, n_kf=8 + (i % 2), n_fixed=1, n_pt=200, n_obs=900, seed=160 + i) for i in range(3)]
    for nwin in (9, 66, 300):    # fused right-looking / split right-looking / left-looking factorisation kernels
        mixed = [good[i % 3] for i in range(nwin)]
        plain = list(mixed)
        mixed[4] = bad
        ba.upload(plain); ba.run(); q0, r0 = ba.download()
        q0 = [x.copy() for x in q0]
        ba.upload(mixed); ba.run(); q1, r1 = ba.download()
        for i in range(nwin):
            if i == 4:
                assert r1[i].status == -2 and r1[i].its_done == (1, 1) and (q1[i].kf_pose == bad.kf_pose).all()
                assert abs(r1[i].chi2_vis - ro.chi2_vis) <= 1e-9 * ro.chi2_vis and (r1[i].obs_outlier == ro.obs_outlier).all()
            else:
                assert r1[i].status == 0 and r1[i].its_done == r0[i].its_done and r1[i].chi2_vis == r0[i].chi2_vis
                assert (q1[i].kf_pose == q0[i].kf_pose).all() and (q1[i].pt == q0[i].pt).all()


def test_levenberg_failed_trial_is_a_rejected_trial(ba, oracle):
    """Levenberg-Marquardt adds lambda to every active diagonal, so a zero pivot needs lambda = 0: a vision-only window whose
    edges all carry zero information (lambda_0 = 1e-5 max diag H = 0).  Every trial then fails to factor (the landmark inverses
    are not finite, nor is any pivot) -> tempChi = max double, rho < 0 -> rejected, state popped, lambda *= ni (still 0), ten
    trials, and the optimize() ends after one iteration (levenberg.cpp:120-147, :155); the status stays VBA_OK, as in g2o, whose
    LM never returns Fail for this.  The result is the untouched window, in both stages."""
    p = synth.make_window(abi.VARIANT_SE3_XYZ, algo=abi.ALGO_LM, n_kf=8, n_fixed=2, n_pt=150, n_obs=800, seed=170)
    p.obs_w = np.zeros_like(p.obs_w)
    qo, ro = oracle.solve(p)
    assert ro.status == 0 and ro.its_done == (1, 1) and (qo.kf_pose == p.kf_pose).all()
    q, r = ba.solve(p)
    assert r.status == 0 and r.its_done == (1, 1) and r.lambda_final == ro.lambda_final == 0.0
    assert (q.kf_pose == p.kf_pose).all() and (q.pt == p.pt).all()
    assert r.chi2_vis == ro.chi2_vis == 0.0 and (r.obs_outlier == ro.obs_outlier).all()
    np.testing.assert_array_equal(r.chi2_trace, ro.chi2_trace)
    # inside a batch: the neighbours are not disturbed
    good = synth.make_window(abi.VARIANT_SE3_XYZ, algo=abi.ALGO_LM, n_kf=8, n_fixed=2, n_pt=150, n_obs=800, seed=172)
    ba.upload([good] * 9); ba.run(); q0, r0 = ba.download()
    q0 = [x.copy() for x in q0]
    ba.upload([good] * 4 + [p] + [good] * 4); ba.run(); q1, r1 = ba.download()
    assert r1[4].its_done == (1, 1) and (q1[4].kf_pose == p.kf_pose).all()
    for i in (0, 3, 5, 8):
        assert r1[i].its_done == r0[i].its_done and (q1[i].kf_pose == q0[i].kf_pose).all() and r1[i].chi2_vis == r0[i].chi2_vis
    _check(good, q1[0], r1[0], *oracle.solve(good))


def test_zero_vision_information_leaves_the_imu_chain(ba, oracle):
    """the same degenerate input with an IMU chain: lambda_0 comes from the IMU blocks, nothing is singular, the window is solved on
    its preintegration factors alone and the landmarks do not move"""
    p = synth.make_window(abi.VARIANT_PRV_XYZ, algo=abi.ALGO_LM, n_kf=8, n_fixed=1, n_pt=150, n_obs=800, seed=171)
    p.obs_w = np.zeros_like(p.obs_w)
    qo, ro = oracle.solve(p)
    q, r = ba.solve(p)
    assert r.status == ro.status == 0 and r.its_done == ro.its_done
    assert (q.pt == p.pt).all() and (qo.pt == p.pt).all()
    np.testing.assert_allclose(q.kf_pose, qo.kf_pose, atol=1e-7)
    np.testing.assert_allclose(r.chi2_prv, ro.chi2_prv, rtol=1e-4, atol=1e-9)
    assert r.chi2_vis == ro.chi2_vis == 0.0
