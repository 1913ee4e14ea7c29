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


# per-frame pose optimisation (BASELINE configs[0] = c1_pose): inputs + oracle outputs
FRAME_CASES = {
    "c1_pose": dict(kind=2, seed=1, n_obs=200),          # vision-only PoseOptimization(Frame*), configs[0]
    "pose_kf": dict(kind=0, seed=6, n_obs=200),          # PoseOptimization(Frame*, KeyFrame*, ...)
    "pose_frame": dict(kind=1, seed=7, n_obs=150),       # PoseOptimization(Frame*, Frame*, ...)
}
FRAME_FIELDS = ["nav", "nav_last", "obs_pw", "obs_uv", "obs_w", "last_pw", "last_uv", "last_w", "K", "T_cb", "g_w", "imu_meas",
                "imu_cov_pvphi", "prior_nav", "prior_info"]
# global bundle adjustment protocol
GBA_CASES = {
    "gba_prv": dict(variant=abi.VARIANT_PRV_XYZ, robust=1, seed=52),
    "gba_vision": dict(variant=abi.VARIANT_SE3_XYZ, robust=0, seed=51),
}


def make_frame_case(kw):
    if kw["kind"] == 2:
        return synth.make_frame_vision(seed=kw["seed"], n_obs=kw["n_obs"])
    return synth.make_frame(seed=kw["seed"], n_obs=kw["n_obs"], last_is_frame=bool(kw["kind"]))


def frame_from_npz(z):
    kw = {f: z["in_" + f] for f in FRAME_FIELDS}
    return abi.FrameProblem(last_is_frame=int(z["kind"]), compute_marg=int(z["compute_marg"]), **kw)


def make_gba_case(kw):
    v = kw["variant"]
    p = synth.make_window(v, algo=abi.ALGO_LM, n_kf=10, n_fixed=0 if v != abi.VARIANT_SE3_XYZ else 1, n_pt=250, n_obs=1500,
                          seed=kw["seed"], outlier_frac=0.02)
    p.protocol, p.robust, p.its_stage1, p.its_stage2 = abi.PROTO_SINGLE, kw["robust"], 20, 0
    p.huber_vis = float(np.float32(np.sqrt(5.99)))
    if v != abi.VARIANT_SE3_XYZ:
        p.kf_fix = np.zeros(p.n_kf, np.uint8); p.kf_fix[0] = 0b101
    return p


def gba_from_npz(z):
    p = problem_from_npz(z)
    p.protocol, p.robust, p.its_stage1, p.its_stage2 = abi.PROTO_SINGLE, int(z["robust"]), 20, 0
    p.huber_vis = float(z["huber_vis"])
    if "in_kf_fix" in z.files:
        p.kf_fix = z["in_kf_fix"]
    return p


def main():
    for name, kw in FRAME_CASES.items():
        f = make_frame_case(kw)
        r = oracle_lib.pose_optimize(f)
        out = {"kind": f.last_is_frame, "compute_marg": f.compute_marg}
        out.update({"in_" + k: getattr(f, k) for k in FRAME_FIELDS})
        out.update(out_nav=r.nav, its_done=np.array(r.its_done), outlier=r.outlier, outlier_last=r.outlier_last, n_inliers=r.n_inliers,
                   chi2_round=r.chi2_round, marg_cov_inv=r.marg_cov_inv)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "its", r.its_done, "inliers", r.n_inliers)
    for name, kw in GBA_CASES.items():
        p = make_gba_case(kw)
        q, r = oracle_lib.solve(p)
        out = {"variant": p.variant, "n_kf_free": p.n_kf_free, "algo": p.algo, "depth_min": p.depth_min, "robust": p.robust, "huber_vis": p.huber_vis}
        out.update({"in_" + f: getattr(p, f) for f in IN_FIELDS})
        if p.kf_fix is not None:
            out["in_kf_fix"] = p.kf_fix
        out.update(out_kf_pose=q.kf_pose, out_kf_vel=q.kf_vel, out_kf_bias=q.kf_bias, out_pt=q.pt,
                   chi2=np.array([r.chi2_vis, r.chi2_prv, r.chi2_bias]), its_done=np.array(r.its_done),
                   obs_outlier=r.obs_outlier, obs_chi2=r.obs_chi2, chi2_trace=r.chi2_trace, lambda_final=r.lambda_final)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "its", r.its_done, "chi2", r.chi2_vis)
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
