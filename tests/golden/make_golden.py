"""Generates the committed golden fixtures: seeded mini windows (inputs) + the CPU oracle's outputs.

The reference ships no golden vectors for this path and cannot run here, so these are
REGRESSION pins of our own oracle (oracle/vba_oracle.c), not reference outputs: they freeze the oracle's
behaviour once its known-answer tests (tests/test_oracle_*.py) are green, and give the GPU tests a target that
does not depend on the oracle library being loadable.   Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from mc_slam_amd import abi, synth  # noqa: E402
import oracle_lib  # noqa: E402

CASES = {
    "c3_mini": dict(variant=abi.VARIANT_PRV_IDP, n_kf=8, n_fixed=1, n_pt=150, n_obs=700, seed=3),     # LocalBAPRVIDP, GN
    "c2_mini": dict(variant=abi.VARIANT_SE3_XYZ, n_kf=8, n_fixed=2, n_pt=150, n_obs=800, seed=2),     # LocalBundleAdjustment, LM
    "c3x_mini": dict(variant=abi.VARIANT_PRV_XYZ, n_kf=8, n_fixed=1, n_pt=150, n_obs=800, seed=5),    # ...NavStatePRV, LM
}
IN_FIELDS = ["kf_pose", "kf_vel", "kf_bias", "pt", "pt_ref_kf", "pt_obs_begin", "obs_kf", "obs_uv", "obs_w", "K", "T_cb", "g_w",
             "imu_kf_i", "imu_kf_j", "imu_meas", "imu_info_prv"]


def problem_from_npz(z):
    kw = {f: z["in_" + f] for f in IN_FIELDS}
    return abi.Problem(variant=int(z["variant"]), n_kf_free=int(z["n_kf_free"]), algo=int(z["algo"]),
                       depth_min=float(z["depth_min"]), **kw)


def main():
    for name, kw in CASES.items():
        p = synth.make_window(**kw)
        q, r = oracle_lib.solve(p)
        out = {"variant": p.variant, "n_kf_free": p.n_kf_free, "algo": p.algo, "depth_min": p.depth_min}
        out.update({"in_" + f: getattr(p, f) for f in IN_FIELDS})
        out.update(out_kf_pose=q.kf_pose, out_kf_vel=q.kf_vel, out_kf_bias=q.kf_bias, out_pt=q.pt,
                   chi2=np.array([r.chi2_vis, r.chi2_prv, r.chi2_bias]), its_done=np.array(r.its_done),
                   obs_outlier=r.obs_outlier, obs_chi2=r.obs_chi2, chi2_trace=r.chi2_trace, lambda_final=r.lambda_final)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "its", r.its_done, "chi2", r.chi2_vis, "outliers", r.n_outliers)


if __name__ == "__main__":
    main()
