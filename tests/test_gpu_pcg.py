"""VBA_SOLVER_PCG (block-Jacobi preconditioned conjugate gradients on the reduced system, BASELINE north_star / configs[3]) against
the LDL^T path of the same backend and against the CPU oracle (which solves directly, as the reference does).
Bars: same iteration counts, same outlier bitmap, final chi2 <= 1e-4 relative."""
import numpy as np
import pytest

from mc_slam_amd import abi, synth, backend
from test_gpu_parity import _check, _gba

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ba():
    b = backend.LocalBA(0)
    yield b
    b.close()


def _with_pcg(p):
    q = p.copy()
    q.solver = abi.SOLVER_PCG
    return q


@pytest.mark.parametrize("variant,algo,kw", [
    (abi.VARIANT_PRV_IDP, abi.ALGO_GN, dict(n_kf=10, n_fixed=1, n_pt=400, n_obs=2000, seed=7)),
    (abi.VARIANT_PRV_IDP, abi.ALGO_GN, dict(n_kf=12, n_fixed=3, n_pt=500, n_obs=2500, seed=8)),
    (abi.VARIANT_SE3_XYZ, abi.ALGO_LM, dict(n_kf=12, n_fixed=2, n_pt=500, n_obs=3000, seed=34)),
    (abi.VARIANT_PRV_XYZ, abi.ALGO_LM, dict(n_kf=12, n_fixed=1, n_pt=500, n_obs=3000, seed=35)),
])
def test_pcg_window_matches_oracle_and_ldlt(ba, oracle, variant, algo, kw):
    p = synth.make_window(variant, algo=algo, **kw)
    q, r = ba.solve(_with_pcg(p))
    q1, r1 = ba.solve(p)
    qo, ro = oracle.solve(p)
    assert r.lin_iterations > 0 and r1.lin_iterations == 0
    _check(p, q, r, qo, ro, trace_rtol=1e-6)
    assert r.its_done == r1.its_done and (r.obs_outlier == r1.obs_outlier).all()
    assert abs(r.chi2_vis - r1.chi2_vis) <= 1e-8 * r1.chi2_vis
    assert np.abs(q.kf_pose - q1.kf_pose).max() <= 1e-7
    # a batch: PCG windows iterate independently, one workgroup each
    ps = [_with_pcg(p)] * 9
    ba.upload(ps); ba.run(); qs, rs = ba.download()
    for qq, rr in zip(qs, rs):
        assert rr.its_done == r.its_done and np.abs(qq.kf_pose - q.kf_pose).max() < 1e-9
    with pytest.raises(RuntimeError, match="mixed batch"):
        ba.upload([p, _with_pcg(p)])


def test_c4_full_size_pcg_equals_ldlt(ba):
    """BASELINE configs[3] (200 KF / 50k landmarks / 500k edges + IMU chain, "Schur + PCG"): the PCG path lands where the direct
    path lands -- same iteration counts, same outlier bitmap, chi2 <= 1e-4 relative (measured ~1e-9), translations <= 1e-6 m"""
    p = synth.config_c4()
    q1, r1 = ba.solve(p)
    q, r = ba.solve(_with_pcg(p))
    assert r.status == r1.status == 0 and r.its_done == r1.its_done
    assert (r.obs_outlier == r1.obs_outlier).all()
    assert abs(r.chi2_vis - r1.chi2_vis) <= 1e-4 * r1.chi2_vis and abs(r.chi2_prv - r1.chi2_prv) <= 1e-4 * r1.chi2_prv
    assert np.abs(q.kf_pose[:, :3] - q1.kf_pose[:, :3]).max() <= 1e-6
    assert r.lin_iterations > sum(r.its_done)          # really iterative


def test_global_ba_150kf_pcg_equals_ldlt(ba):
    p = _gba(abi.VARIANT_PRV_XYZ, 1, n_kf=150, n_pt=12000, n_obs=80000, seed=60, its=10)
    q1, r1 = ba.solve(p)
    q, r = ba.solve(_with_pcg(p))
    assert r.status == r1.status == 0 and r.its_done == r1.its_done
    assert abs(r.chi2_vis - r1.chi2_vis) <= 1e-4 * r1.chi2_vis
    assert abs(r.lambda_final - r1.lambda_final) <= 1e-6 * r1.lambda_final
    assert np.abs(q.kf_pose[:, :3] - q1.kf_pose[:, :3]).max() <= 1e-6
    assert (q.kf_pose[0] == p.kf_pose[0]).all()        # the gauge keyframe did not move


def test_pcg_window_with_many_keyframes_and_few_landmarks(ba, oracle):
    """k_pcg_matvec leaves one partial per 64 rows of the reduced system in the window's slice of the partial-sum array; that slice
    used to be sized by landmarks and observations only, so a window with many keyframes but few landmarks (here 600 rows = 10
    partials against 3 * ceil(64 / 64) + 2 = 5 doubles) ran into its neighbour's slice.  A batch of such windows must
    solve like each of them alone, and like the direct solver."""
    ps = [_with_pcg(synth.make_window(abi.VARIANT_PRV_IDP, n_kf=41, n_fixed=1, n_pt=64, n_obs=400, seed=180 + i)) for i in range(3)]
    assert 15 * ps[0].n_kf_free // 64 > 3 * ((ps[0].n_pt + 63) // 64) + 2
    ba.upload(ps * 3); ba.run(); qs, rs = ba.download()
    qs = [x.copy() for x in qs]
    for i, p in enumerate(ps):
        q1, r1 = ba.solve(p)
        qd, rd = ba.solve(synth.make_window(abi.VARIANT_PRV_IDP, n_kf=41, n_fixed=1, n_pt=64, n_obs=400, seed=180 + i))
        for j in (i, i + 3, i + 6):
            assert rs[j].status == r1.status == 0 and rs[j].its_done == r1.its_done == rd.its_done
            assert np.abs(qs[j].kf_pose - q1.kf_pose).max() < 1e-9
        assert abs(r1.chi2_vis - rd.chi2_vis) <= 1e-6 * rd.chi2_vis and np.abs(q1.kf_pose[:, :3] - qd.kf_pose[:, :3]).max() <= 1e-6
