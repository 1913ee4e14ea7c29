"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the CPU oracle; the reference has
none): the oracle must keep reproducing them, and the HIP backend must hit them through the C-ABI."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_golden import CASES, FRAME_CASES, GBA_CASES, problem_from_npz, frame_from_npz, gba_from_npz  # noqa: E402


def _load(name):
    z = np.load(os.path.join(HERE, "golden", name + ".npz"))
    return z, problem_from_npz(z)


def _compare(z, q, r, tol_t=1e-6, rtol_chi=1e-4):
    assert tuple(z["its_done"]) == r.its_done
    assert (z["obs_outlier"] == r.obs_outlier).all()
    chi = z["chi2"]
    assert abs(chi[0] - r.chi2_vis) <= rtol_chi * chi[0]
    assert abs(chi[1] - r.chi2_prv) <= rtol_chi * max(chi[1], 1e-9)
    assert np.abs(z["out_kf_pose"][:, :3] - q.kf_pose[:, :3]).max() <= tol_t
    assert np.abs(z["out_pt"] - q.pt).max() <= 1e-6 * max(1.0, np.abs(z["out_pt"]).max())
    np.testing.assert_allclose(z["chi2_trace"], r.chi2_trace, rtol=1e-7)


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_golden(oracle, name):
    z, p = _load(name)
    q, r = oracle.solve(p)
    _compare(z, q, r, tol_t=1e-9, rtol_chi=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_gpu_matches_golden(name):
    from mc_slam_amd import backend
    z, p = _load(name)
    ba = backend.LocalBA(0)
    q, r = ba.solve(p)
    ba.close()
    _compare(z, q, r)


# ---- global BA protocol and per-frame pose optimisation ----
@pytest.mark.parametrize("name", sorted(GBA_CASES))
def test_oracle_reproduces_golden_gba(oracle, name):
    z = np.load(os.path.join(HERE, "golden", name + ".npz"))
    q, r = oracle.solve(gba_from_npz(z))
    _compare(z, q, r, tol_t=1e-9, rtol_chi=1e-9)
    assert r.its_done[1] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(GBA_CASES))
def test_gpu_matches_golden_gba(name):
    from mc_slam_amd import backend
    z = np.load(os.path.join(HERE, "golden", name + ".npz"))
    ba = backend.LocalBA(0)
    q, r = ba.solve(gba_from_npz(z))
    ba.close()
    _compare(z, q, r)


def _compare_frame(z, r, tol=1e-6):
    assert tuple(z["its_done"]) == r.its_done and int(z["n_inliers"]) == r.n_inliers
    assert (z["outlier"] == r.outlier).all() and (z["outlier_last"] == r.outlier_last).all()
    assert np.abs(z["out_nav"][:10] - r.nav[:10]).max() <= tol
    np.testing.assert_allclose(z["chi2_round"], r.chi2_round, rtol=1e-7)
    np.testing.assert_allclose(z["marg_cov_inv"], r.marg_cov_inv, rtol=1e-5, atol=1e-7 * max(1.0, np.abs(z["marg_cov_inv"]).max()))


@pytest.mark.parametrize("name", sorted(FRAME_CASES))
def test_oracle_reproduces_golden_frames(oracle, name):
    z = np.load(os.path.join(HERE, "golden", name + ".npz"))
    _compare_frame(z, oracle.pose_optimize(frame_from_npz(z)), tol=1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(FRAME_CASES))
def test_gpu_matches_golden_frames(name):
    from mc_slam_amd import backend
    z = np.load(os.path.join(HERE, "golden", name + ".npz"))
    ba = backend.LocalBA(0)
    r = ba.pose_optimize([frame_from_npz(z)])[0]
    ba.close()
    _compare_frame(z, r)
