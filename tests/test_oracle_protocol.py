"""The two-stage LocalBAPRVIDP protocol driven from Python, written from the reference text, against the oracle's own driver.

The oracle (oracle/vba_oracle.c) is one author's reading of the reference; its Jacobians and its assembly are pinned by finite
differences and by numpy solves elsewhere in this suite.  What is left is the CONTROL FLOW: the Gauss-Newton loop with the
|pre - after| < 1e-3 stop (Thirdparty/g2o/g2o/core/optimization_algorithm_gauss_newton.cpp:50-105, the block "ADD BY wangjing"),
how SparseOptimizer::optimize counts iterations (sparse_optimizer.cpp:354-419), the outlier pass between the two optimize()
calls and the final erase rule (src/Optimizer.cpp:458-517).  Here that flow is restated a second time, in Python, on top of
the oracle's linearisation primitive only (one dense H, b per call; the update goes through numpy.linalg.solve on the UNREDUCED
system and the retraction hooks), and must land where vba_oracle_solve lands: same iteration counts, same levels, same erase
bitmap, states equal to solver accuracy."""
import numpy as np
import pytest

from mc_slam_amd import abi, synth


def _chi(oracle, p, robust, lvl):
    return oracle.linearize_ex(p, robust, lvl)[2]


def _gn_step(oracle, p, robust, lvl):
    """one OptimizationAlgorithmGaussNewton::solve: returns (preChi2, afterChi2); p is updated in place"""
    H, b, pre = oracle.linearize_ex(p, robust, lvl)
    act = np.flatnonzero(np.diag(H) != 0.0)                # vertices outside the active set have no row
    x = np.zeros(len(b))
    x[act] = np.linalg.solve(H[np.ix_(act, act)], b[act])   # Hx = b on the unreduced system (g2o solves it through the Schur complement)
    nf = p.n_kf_free
    for a in range(nf):                                     # SparseOptimizer::update -> oplusImpl of every active vertex
        dx = x[15 * a:15 * a + 15]
        if H[15 * a, 15 * a] != 0.0:
            p.kf_pose[a] = oracle.oplus_pr(p.kf_pose[a], dx[:6])          # NavState::IncSmallPR, src/IMU/NavState.cpp:63-70
        if H[15 * a + 6, 15 * a + 6] != 0.0:
            p.kf_vel[a] += dx[6:9]                                         # IncSmallV :74-77
        if H[15 * a + 9, 15 * a + 9] != 0.0:
            p.kf_bias[a, 6:12] += dx[9:15]                                 # IncSmallBias :100-109
    dl = x[15 * nf:]
    on = np.diag(H)[15 * nf:] != 0.0
    rho = p.pt[:, 0] + np.where(on, dl, 0.0)
    p.pt[:, 0] = np.where(on, np.maximum(rho, 1e-6), p.pt[:, 0])          # VertexIDP::oplusImpl, src/IMU/g2otypes.h:50-55
    return pre, _chi(oracle, p, robust, lvl)


def _optimize(oracle, p, iterations, robust, lvl):
    """SparseOptimizer::optimize(iterations): cjIterations, counting the terminating iteration"""
    done = 0
    for _ in range(iterations):
        pre, after = _gn_step(oracle, p, robust, lvl)
        done += 1
        if abs(pre - after) < 1e-3:                         # Terminate, gauss_newton.cpp:97
            break
    return done


def protocol_twin(oracle, p0):
    p = p0.copy()
    lvl = np.zeros(p.n_obs, np.uint8)
    its1 = _optimize(oracle, p, p.its_stage1, True, lvl)                  # optimize(5), Huber on every vision edge
    _, _, _, _, chi_e, depth = oracle.evaluate(p, robust_vis=False)       # e->chi2(), isDepthPositive() at the current estimates
    rho_e = np.repeat(p.pt[:, 0], np.diff(p.pt_obs_begin))
    lvl = ((chi_e > p.chi2_th) | ~(depth > p.depth_min) | (rho_e < p.rho_min)).astype(np.uint8)   # src/Optimizer.cpp:483-487
    its2 = _optimize(oracle, p, p.its_stage2, False, lvl)                 # setRobustKernel(0); initializeOptimization(0); optimize(10)
    _, _, _, _, chi_e, depth = oracle.evaluate(p, robust_vis=False)
    rho_e = np.repeat(p.pt[:, 0], np.diff(p.pt_obs_begin))
    erase = (chi_e > p.chi2_th) | ~(depth > p.depth_min) | (rho_e < p.rho_min) | (lvl != 0)      # :503-508
    return p, (its1, its2), lvl, erase


@pytest.mark.parametrize("kw", [
    dict(n_kf=6, n_fixed=1, n_pt=60, n_obs=240, seed=31),
    dict(n_kf=8, n_fixed=1, n_pt=120, n_obs=600, seed=7),
    dict(n_kf=9, n_fixed=3, n_pt=100, n_obs=450, seed=8),      # fixed co-observers and fixed reference keyframes
])
def test_two_stage_protocol_restated_in_python_lands_where_the_oracle_lands(oracle, kw):
    p0 = synth.make_window(abi.VARIANT_PRV_IDP, **kw)
    q, its, lvl, erase = protocol_twin(oracle, p0)
    qo, ro = oracle.solve(p0)
    assert its == ro.its_done, (its, ro.its_done)
    assert (erase == ro.obs_outlier.astype(bool)).all()
    assert erase.sum() > 0 and 0 < lvl.sum() <= erase.sum()              # the outlier pass did something; final erase includes the level-1 edges
    np.testing.assert_allclose(q.kf_pose, qo.kf_pose, rtol=0, atol=1e-8)
    np.testing.assert_allclose(q.kf_vel, qo.kf_vel, rtol=0, atol=1e-7)
    np.testing.assert_allclose(q.pt[:, 0], qo.pt[:, 0], rtol=1e-7, atol=1e-9)
    out = oracle.evaluate(q, robust_vis=False)
    chi_vis_l0 = out[4][lvl == 0].sum()
    assert abs(chi_vis_l0 - ro.chi2_vis) <= 1e-7 * ro.chi2_vis          # N3: sum over level-0 vision edges at the final estimates


def test_noise_free_window_stops_after_one_iteration_per_stage(oracle):
    p0 = synth.make_window(abi.VARIANT_PRV_IDP, n_kf=6, n_fixed=1, n_pt=60, n_obs=240, seed=12, noise=False)
    q, its, lvl, erase = protocol_twin(oracle, p0)
    qo, ro = oracle.solve(p0)
    assert its == ro.its_done == (1, 1) and not erase.any() and not ro.obs_outlier.any()


# ---- Levenberg-Marquardt (XYZ landmarks): optimization_algorithm_levenberg.cpp:61-164 restated ----
class _LM:
    """the state OptimizationAlgorithmLevenberg keeps between solve() calls"""
    lam = 0.0; ni = 2.0; nbad = 0


def _apply_xyz(oracle, p, x, H):
    nf, P = p.n_kf_free, (6 if p.variant == abi.VARIANT_SE3_XYZ else 15)
    for a in range(nf):
        dx = x[P * a:P * a + P]
        if H[P * a, P * a] != 0.0:
            p.kf_pose[a] = oracle.oplus_se3(p.kf_pose[a], dx[:6]) if P == 6 else oracle.oplus_pr(p.kf_pose[a], dx[:6])
        if P == 15:
            if H[P * a + 6, P * a + 6] != 0.0:
                p.kf_vel[a] += dx[6:9]
            if H[P * a + 9, P * a + 9] != 0.0:
                p.kf_bias[a, 6:12] += dx[9:15]
    dl = x[P * nf:].reshape(-1, 3)
    on = np.diag(H)[P * nf:].reshape(-1, 3)[:, 0] != 0.0
    p.pt[on] += dl[on]                                                    # VertexSBAPointXYZ::oplusImpl, types_sba.h:52-56


def _lm_solve(oracle, p, iteration, st, robust, lvl):
    """one OptimizationAlgorithmLevenberg::solve; returns 'OK' or 'Terminate'"""
    H, b, current = oracle.linearize_ex(p, robust, lvl)
    ini = current
    act = np.flatnonzero(np.diag(H) != 0.0)
    if iteration == 0:
        st.lam = 1e-5 * np.abs(np.diag(H)[act]).max()                     # computeLambdaInit, _tau = 1e-5
        st.ni, st.nbad = 2.0, 0
    qmax, rho = 0, 0.0
    while True:
        saved = (p.kf_pose.copy(), p.kf_vel.copy(), p.kf_bias.copy(), p.pt.copy())      # _optimizer->push()
        x = np.zeros(len(b))
        x[act] = np.linalg.solve(H[np.ix_(act, act)] + st.lam * np.eye(len(act)), b[act])
        _apply_xyz(oracle, p, x, H)
        temp = _chi(oracle, p, robust, lvl)
        scale = float((x * (st.lam * x + b)).sum()) + 1e-3                # computeScale
        rho = (current - temp) / scale
        if rho > 0 and np.isfinite(temp):
            alpha = min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0)
            st.lam *= max(1.0 / 3.0, alpha)
            st.ni = 2.0
            current = temp                                                # discardTop
        else:
            st.lam *= st.ni
            st.ni *= 2
            p.kf_pose[...], p.kf_vel[...], p.kf_bias[...], p.pt[...] = saved              # pop
        qmax += 1
        if not (rho < 0 and qmax < 10):
            break
    if qmax == 10 or rho == 0:
        return "Terminate", current
    st.nbad = st.nbad + 1 if (ini - current) * 1e3 < ini else 0
    return ("Terminate" if st.nbad >= 3 else "OK"), current


def _optimize_lm(oracle, p, iterations, robust, lvl, st):
    done = 0
    for i in range(iterations):
        res, _ = _lm_solve(oracle, p, i, st, robust, lvl)
        done += 1
        if res != "OK":
            break
    return done


@pytest.mark.parametrize("variant,kw", [
    (abi.VARIANT_SE3_XYZ, dict(n_kf=6, n_fixed=2, n_pt=60, n_obs=300, seed=33)),
    (abi.VARIANT_SE3_XYZ, dict(n_kf=8, n_fixed=2, n_pt=120, n_obs=700, seed=34)),
    (abi.VARIANT_PRV_XYZ, dict(n_kf=6, n_fixed=1, n_pt=60, n_obs=300, seed=32)),
])
def test_levenberg_protocol_restated_in_python_lands_where_the_oracle_lands(oracle, variant, kw):
    """LocalBundleAdjustment (src/Optimizer.cpp:4093-4143) / LocalBundleAdjustmentNavStatePRV (:1259-1317): optimize(5) with Huber,
    the chi2 > 5.991 / depth pass, optimize(10) without -- Levenberg-Marquardt with g2o's lambda / rho schedule"""
    p0 = synth.make_window(variant, algo=abi.ALGO_LM, **kw)
    p = p0.copy()
    st = _LM()
    lvl = np.zeros(p.n_obs, np.uint8)
    its1 = _optimize_lm(oracle, p, p.its_stage1, True, lvl, st)
    _, _, _, _, chi1, depth = oracle.evaluate(p, robust_vis=False)
    lvl = ((chi1 > p.chi2_th) | ~(depth > p.depth_min)).astype(np.uint8)
    its2 = _optimize_lm(oracle, p, p.its_stage2, False, lvl, st)
    _, _, _, _, chi2, depth = oracle.evaluate(p, robust_vis=False)
    chi_stored = np.where(lvl == 0, chi2, chi1)          # a level-1 edge keeps the _error of its last evaluation; depth is read fresh
    erase = (chi_stored > p.chi2_th) | ~(depth > p.depth_min)
    qo, ro = oracle.solve(p0)
    assert (its1, its2) == ro.its_done, ((its1, its2), ro.its_done)
    assert (erase == ro.obs_outlier.astype(bool)).all()
    assert abs(st.lam - ro.lambda_final) <= 1e-6 * ro.lambda_final
    np.testing.assert_allclose(p.kf_pose, qo.kf_pose, rtol=0, atol=1e-8)
    np.testing.assert_allclose(p.pt, qo.pt, rtol=0, atol=1e-7)
