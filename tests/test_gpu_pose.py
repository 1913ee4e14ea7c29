"""GPU parity of the IMU-aided per-frame PoseOptimization against the CPU oracle, through the C-ABI."""
import numpy as np
import pytest

from mc_slam_amd import abi, synth, backend

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ba():
    b = backend.LocalBA(0)
    yield b
    b.close()


def _check(f, r, ro):
    assert r.status == ro.status == 0
    assert r.its_done == ro.its_done, (r.its_done, ro.its_done, r.chi2_round, ro.chi2_round)
    assert (r.outlier == ro.outlier).all() and (r.outlier_last == ro.outlier_last).all()
    assert r.n_inliers == ro.n_inliers
    np.testing.assert_allclose(r.chi2_round, ro.chi2_round, rtol=1e-7)
    assert np.abs(r.nav[:3] - ro.nav[:3]).max() <= 1e-6            # translations: the north-star bar
    assert np.abs(r.nav[3:7] - ro.nav[3:7]).max() <= 1e-7
    assert np.abs(r.nav[7:10] - ro.nav[7:10]).max() <= 1e-6
    np.testing.assert_allclose(r.nav[16:22], ro.nav[16:22], atol=1e-8)
    assert (r.nav[10:16] == f.nav[10:16]).all()                    # bg, ba are never touched
    if f.compute_marg:
        np.testing.assert_allclose(r.marg_cov_inv, ro.marg_cov_inv, rtol=1e-5, atol=1e-7 * np.abs(ro.marg_cov_inv).max())
    else:
        assert (r.marg_cov_inv == 0).all()


@pytest.mark.parametrize("lif,seed,n_obs", [(False, 21, 300), (False, 22, 120), (True, 23, 300), (True, 24, 90)])
def test_pose_optimization_matches_oracle(ba, oracle, lif, seed, n_obs):
    f = synth.make_frame(seed=seed, n_obs=n_obs, last_is_frame=lif, outlier_frac=0.1)
    r = ba.pose_optimize([f])[0]
    ro = oracle.pose_optimize(f)
    _check(f, r, ro)
    assert r.outlier.sum() > 0


def test_batch_of_frames_equals_single_calls(ba, oracle):
    frames = [synth.make_frame(seed=30 + i, n_obs=60 + 37 * i, last_is_frame=bool(i % 2), compute_marg=bool(i % 3)) for i in range(7)]
    tiny = synth.make_frame(seed=40, n_obs=60)
    frames.append(abi.FrameProblem(nav=tiny.nav, nav_last=tiny.nav_last, obs_pw=tiny.obs_pw[:2], obs_uv=tiny.obs_uv[:2],
                                   obs_w=tiny.obs_w[:2], K=tiny.K, T_cb=tiny.T_cb, g_w=tiny.g_w, imu_meas=tiny.imu_meas,
                                   imu_cov_pvphi=tiny.imu_cov_pvphi))
    rs = ba.pose_optimize(frames)
    for f, r in zip(frames[:-1], rs[:-1]):
        _check(f, r, oracle.pose_optimize(f))
        r1 = ba.pose_optimize([f])[0]
        assert (r1.nav == r.nav).all() and (r1.outlier == r.outlier).all() and (r1.marg_cov_inv == r.marg_cov_inv).all()
    assert rs[-1].n_inliers == 0 and (rs[-1].nav == tiny.nav).all()     # fewer than 3 correspondences: untouched


def test_noise_free_frame_recovers_the_truth(ba):
    f = synth.make_frame(seed=41, n_obs=200, noise=False)
    r = ba.pose_optimize([f])[0]
    assert r.outlier.sum() == 0 and r.n_inliers == 200
    assert np.abs(r.nav[:3] - f.truth["nav"][:3]).max() < 1e-3


def test_c1_vision_only_pose_optimization_matches_oracle(ba, oracle):
    """BASELINE configs[0]: Optimizer::PoseOptimization(Frame*) on one frame with ~200 monocular correspondences
    (src/Optimizer.cpp:3610-3835), seed 1 -- and a batch that mixes all three frame kinds."""
    f = synth.make_frame_vision(seed=1, n_obs=200)
    r = ba.pose_optimize([f])[0]
    ro = oracle.pose_optimize(f)
    assert r.its_done == ro.its_done and (r.outlier == ro.outlier).all() and r.n_inliers == ro.n_inliers
    np.testing.assert_allclose(r.chi2_round, ro.chi2_round, rtol=1e-7)
    assert np.abs(r.nav[:3] - ro.nav[:3]).max() <= 1e-6 and np.abs(r.nav[3:7] - ro.nav[3:7]).max() <= 1e-7
    assert (r.nav[7:] == 0).all() and r.outlier.sum() > 0
    mix = [synth.make_frame_vision(seed=2, n_obs=150), synth.make_frame(seed=3, n_obs=100), synth.make_frame(seed=4, n_obs=100, last_is_frame=True), f]
    rs = ba.pose_optimize(mix)
    for g, rr in zip(mix, rs):
        rq = oracle.pose_optimize(g)
        assert rr.its_done == rq.its_done and (rr.outlier == rq.outlier).all()
        assert np.abs(rr.nav[:7] - rq.nav[:7]).max() <= 1e-6
    assert (rs[-1].nav == r.nav).all()


def test_bad_frames_are_refused_with_a_message(ba):
    """negative n_obs_last and a singular / non-finite preintegration covariance fail the call instead of corrupting memory or
    returning NaN information with VBA_OK"""
    import ctypes as C
    from mc_slam_amd import abi
    f = synth.make_frame(seed=45, n_obs=60, last_is_frame=True)
    packed = ba.pose_pack([f])
    packed[1][0].n_obs_last = -5
    assert ba.lib.vba_pose_optimize(ba.h, 1, packed[3], packed[4]) != 0
    assert b"n_obs_last" in ba.lib.vba_last_error(ba.h)
    for bad in (np.zeros((9, 9)), np.full((9, 9), np.nan)):
        g = synth.make_frame(seed=46, n_obs=60)
        g.imu_cov_pvphi = bad
        with pytest.raises(RuntimeError, match="singular or not finite"):
            ba.pose_optimize([g])
    r = ba.pose_optimize([synth.make_frame(seed=46, n_obs=60)])[0]       # the handle is still usable
    assert r.status == 0 and r.n_inliers > 0
