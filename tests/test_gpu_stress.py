"""Randomised parity sweep: windows of random shape (keyframes, fixed keyframes, track length, outlier share, caller-side
order) through the C-ABI against the CPU oracle, all variants.  The bars are those of test_gpu_parity."""
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
                          outlier_frac=float(rng.choice([0.0, 0.05, 0.15])))
    if i % 3 == 0:
        p = _shuffled(p, i, drop_middle=(i % 2 == 0))
    return p


@pytest.mark.parametrize("i", range(24))
def test_random_window_matches_oracle(ba, oracle, i):
    p = _random_window(i)
    q, r = ba.solve(p)
    qo, ro = oracle.solve(p)
    _check(p, q, r, qo, ro)


def test_random_windows_in_one_ragged_batch(ba):
    """the same windows, variant by variant, as ragged batches: every window equals its single solve"""
    ps = [_random_window(i) for i in range(24)]
    for variant in (abi.VARIANT_PRV_IDP, abi.VARIANT_PRV_XYZ, abi.VARIANT_SE3_XYZ):
        sel = [p for p in ps if p.variant == variant]
        singles = [ba.solve(p) for p in sel]
        ba.upload(sel); ba.run(); qs, rs = ba.download()
        for (q1, r1), q, r in zip(singles, qs, rs):
            assert r.its_done == r1.its_done and r.status == r1.status and (r.obs_outlier == r1.obs_outlier).all()
            assert np.abs(q.kf_pose - q1.kf_pose).max() < 1e-9 and np.abs(q.pt - q1.pt).max() < 1e-8
