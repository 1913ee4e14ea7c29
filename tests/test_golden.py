"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the CPU oracle; the reference has
none): the oracle must keep reproducing them, and the HIP backend must hit them through the C-ABI."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_golden import CASES, problem_from_npz  # noqa: E402


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
