"""Randomised parity sweep: windows of random shape (keyframes, fixed keyframes, track length, outlier share, caller-side
order) through the C-ABI against the CPU oracle, all variants.  The bars are those of test_gpu_parity."""
import os

import numpy as np
import pytest

from mc_slam_amd import abi, synth, backend
from test_gpu_parity import _check, _shuffled

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ba():
    b = backend.LocalBA(0)
    yield b
    b.close()


def _random_window(i):
    rng = np.random.default_rng(5000 + i)
    variant = [abi.VARIANT_PRV_IDP, abi.VARIANT_PRV_IDP, abi.VARIANT_PRV_XYZ, abi.VARIANT_SE3_XYZ][i % 4]
    algo = abi.ALGO_GN if variant == abi.VARIANT_PRV_IDP else abi.ALGO_LM
    n_kf = int(rng.integers(6, 28))
    n_fixed = int(rng.integers(1, 4)) if n_kf > 8 else 1
    if variant == abi.VARIANT_SE3_XYZ:
        n_fixed = max(n_fixed, 2)
    per = int(rng.integers(3, max(4, min(8, n_kf - 3))))
    n_pt = int(rng.integers(40, 700))
    n_obs = n_pt * per + int(rng.integers(0, n_pt))
    p = synth.make_window(variant, algo=algo, n_kf=n_kf, n_fixed=n_fixed, n_pt=n_pt, n_obs=n_obs, seed=1000 + i,
                          outlier_frac=float(rng.choice([0.0, 0.05, 0.15])),
                          landmark_order="caller" if (i // 4) % 2 else "random")   # half of the windows in the reference caller's landmark order
    if i % 3 == 0:
        p = _shuffled(p, i, drop_middle=(i % 2 == 0))
    return p


N_WINDOWS = int(os.environ.get("VBA_STRESS_WINDOWS", "24"))   # a longer hunt: VBA_STRESS_WINDOWS=300
N_FRAMES = int(os.environ.get("VBA_STRESS_FRAMES", "18"))


@pytest.mark.parametrize("i", range(N_WINDOWS))
def test_random_window_matches_oracle(ba, oracle, i):
    p = _random_window(i)
    q, r = ba.solve(p)
    qo, ro = oracle.solve(p)
    # the north-star bars as in test_gpu_parity; the chi2 TRACE only to 1e-5: on a slowly converging (ill-conditioned) LM
    # window of the long hunt (i = 63, SE3, 14 iterations) the summation order shows at 1.9e-7 relative
    # Windows with fewer than six landmarks per keyframe are under-constrained: at i = 63 (26 keyframes, 94 landmarks) the final
    # system has cond(H) = 2e20 and only the LM damping makes it solvable; a float64 LAPACK twin of the same protocol
    # (tests/test_oracle_protocol.py) then differs from the oracle by 6e-7 m in the translations, 2e-6 in the landmarks and 1e-5
    # relative in lambda -- the arithmetic's noise floor.  Iteration counts, the outlier bitmap and chi2 keep their bars there;
    # the states get ten times the bar.
    _check(p, q, r, qo, ro, trace_rtol=1e-5, state_scale=10.0 if p.n_pt < 6 * p.n_kf else 1.0)


def test_random_windows_in_one_ragged_batch(ba):
    """the same windows, variant by variant, as ragged batches: every window equals its single solve"""
    ps = [_random_window(i) for i in range(N_WINDOWS)]
    for variant in (abi.VARIANT_PRV_IDP, abi.VARIANT_PRV_XYZ, abi.VARIANT_SE3_XYZ):
        sel = [p for p in ps if p.variant == variant]
        singles = [ba.solve(p) for p in sel]
        ba.upload(sel); ba.run(); qs, rs = ba.download()
        for (q1, r1), q, r in zip(singles, qs, rs):
            assert r.its_done == r1.its_done and r.status == r1.status and (r.obs_outlier == r1.obs_outlier).all()
            # batch and single solve take different factorisation kernels (left- / right-looking): same arithmetic up to
            # rounding, which an ill-conditioned LM window of the long hunt amplifies to 5e-7 -- the bar is the north star's 1e-6 m
            assert np.abs(q.kf_pose - q1.kf_pose).max() < 1e-6 and np.abs(q.pt - q1.pt).max() < 1e-5 * max(1.0, np.abs(q1.pt).max())


def _its_agree(a, b):
    """LM iteration counts per round.  At a converged point the last iteration's gain is zero up to rounding, so whether one
    more (rejected or empty) iteration is counted depends on the summation order: a difference of one is a tie, not an error,
    as long as the chi2 of the round agrees (checked by the callers to 1e-7 relative)."""
    return all(abs(x - y) <= 1 for x, y in zip(a, b))


@pytest.mark.parametrize("i", range(N_FRAMES))
def test_random_frame_pose_optimization_matches_oracle(ba, oracle, i):
    """PoseOptimization: frames of random size, kind (last keyframe / last frame / vision only), outlier share
    and marginalisation switch against the oracle; the bars are those of test_gpu_pose, except that the per-round iteration
    counts may differ by one at a converged tie (seed 517: (5,5,5,4) vs (5,5,5,5) with chi2 equal to 1e-14 relative)"""
    rng = np.random.default_rng(7000 + i)
    n_obs = int(rng.integers(12, 420))
    kind = i % 3
    if kind == 2:
        f = synth.make_frame_vision(seed=500 + i, n_obs=n_obs)
    else:
        f = synth.make_frame(seed=500 + i, n_obs=n_obs, last_is_frame=bool(kind), compute_marg=bool(i % 2),
                             outlier_frac=float(rng.choice([0.0, 0.1, 0.25])))
    r = ba.pose_optimize([f])[0]
    ro = oracle.pose_optimize(f)
    assert r.status == ro.status == 0
    assert _its_agree(r.its_done, ro.its_done), (r.its_done, ro.its_done)
    assert (r.outlier == ro.outlier).all() and (r.outlier_last == ro.outlier_last).all() and r.n_inliers == ro.n_inliers
    np.testing.assert_allclose(r.chi2_round, ro.chi2_round, rtol=1e-7)
    assert np.abs(r.nav[:3] - ro.nav[:3]).max() <= 1e-6 and np.abs(r.nav[3:7] - ro.nav[3:7]).max() <= 1e-7
    if kind != 2:
        assert np.abs(r.nav[7:10] - ro.nav[7:10]).max() <= 1e-6
        np.testing.assert_allclose(r.nav[16:22], ro.nav[16:22], atol=1e-8)
        if f.compute_marg:
            np.testing.assert_allclose(r.marg_cov_inv, ro.marg_cov_inv, rtol=1e-5, atol=1e-7 * np.abs(ro.marg_cov_inv).max())
