"""Synthetic EuRoC-calibrated local-BA windows.

No EuRoC data exists in the build environment, so "EuRoC-derived" means: EuRoC camera intrinsics and
camera-IMU extrinsic (config/euroc.yaml:40-44,54-57), 200 Hz IMU / 4 Hz keyframes, the reference's IMU
noise constants (src/IMU/imudata.cpp:25-31), ORB scale-pyramid weights (1/1.2^(2*octave)) and the
reference's own preintegration recursion (src/IMU/IMUPreintegrator.cpp:63-112, restated here in numpy so
that input synthesis never touches oracle/).

The ground-truth trajectory is DEFINED as the discrete integration of the noise-free IMU signal with the
reference's update rule, so noise-free preintegrated factors have exactly zero residual at ground truth.
"""
import numpy as np

from . import abi

EUROC_K = np.array([458.654, 457.296, 367.215, 248.375])
EUROC_WH = (752.0, 480.0)
EUROC_TBC = np.array([
    [0.0148655429818, -0.999880929698, 0.00414029679422, -0.0216401454975],
    [0.999557249008, 0.0149672133247, 0.025715529948, -0.064676986768],
    [-0.0257744366974, 0.00375618835797, 0.999660727178, 0.00981073058949],
    [0.0, 0.0, 0.0, 1.0]])
GRAVITY = 9.810  # src/IMU/configparam.cpp:6


# ---------------------------------------------------------------- small Lie helpers (batched numpy)
def hat(v):
    v = np.asarray(v, dtype=np.float64)
    M = np.zeros(v.shape[:-1] + (3, 3))
    M[..., 0, 1], M[..., 0, 2] = -v[..., 2], v[..., 1]
    M[..., 1, 0], M[..., 1, 2] = v[..., 2], -v[..., 0]
    M[..., 2, 0], M[..., 2, 1] = -v[..., 1], v[..., 0]
    return M


def so3_exp(w):
    """Rodrigues, batched: (...,3) -> (...,3,3)."""
    w = np.asarray(w, dtype=np.float64)
    th = np.linalg.norm(w, axis=-1)[..., None, None]
    W = hat(w)
    W2 = W @ W
    small = th < 1e-8
    ths = np.where(small, 1.0, th)
    a = np.where(small, 1.0 - th * th / 6.0, np.sin(ths) / ths)
    b = np.where(small, 0.5 - th * th / 24.0, (1 - np.cos(ths)) / (ths * ths))
    return np.eye(3) + a * W + b * W2


def so3_jr(w):
    """Right Jacobian, the reference's formula (src/IMU/so3.cpp:33-50), batched."""
    w = np.asarray(w, dtype=np.float64)
    th = np.linalg.norm(w, axis=-1)[..., None, None]
    small = th < 0.00001
    ths = np.where(small, 1.0, th)
    K = hat(w) / ths
    J = np.eye(3) - (1 - np.cos(ths)) / ths * K + (1 - np.sin(ths) / ths) * (K @ K)
    return np.where(small, np.eye(3), J)


def rot_to_quat(R):
    """3x3 -> (x,y,z,w), w >= 0, unit."""
    R = np.asarray(R, dtype=np.float64)
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[3] = (R[k, j] - R[j, k]) / s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


def quat_to_rot(q):
    x, y, z, w = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _orthonormalise(R):
    return quat_to_rot(rot_to_quat(R))  # as configparam.cpp:55-56


def extrinsics():
    """(R_bc, p_bc, T_cb[7]) from config/euroc.yaml, rotation re-normalised through a quaternion."""
    R_bc = _orthonormalise(EUROC_TBC[:3, :3])
    p_bc = EUROC_TBC[:3, 3].copy()
    R_cb = R_bc.T
    t_cb = -R_cb @ p_bc
    return R_bc, p_bc, np.concatenate([t_cb, rot_to_quat(R_cb)])


# ---------------------------------------------------------------- preintegration (numpy restatement of A15)
def preintegrate(omega, acc, dts):
    """IMUPreintegrator::update applied over samples, batched over E edges.

    omega, acc: [E,S,3] bias-corrected samples, dts: [E,S].  Returns (meas [E,61], cov_PVphi [E,9,9]).
    Follows src/IMU/IMUPreintegrator.cpp:63-112 line by line.
    """
    E, S, _ = omega.shape
    dP = np.zeros((E, 3)); dV = np.zeros((E, 3)); dR = np.tile(np.eye(3), (E, 1, 1))
    JPg = np.zeros((E, 3, 3)); JPa = np.zeros((E, 3, 3)); JVg = np.zeros((E, 3, 3)); JVa = np.zeros((E, 3, 3))
    JRg = np.zeros((E, 3, 3))
    cov = np.zeros((E, 9, 9)); T = np.zeros(E)
    I3 = np.eye(3)
    for s in range(S):
        dt = dts[:, s][:, None, None]
        dt2 = dt * dt
        w, a = omega[:, s], acc[:, s]
        dRk = so3_exp(w * dts[:, s][:, None])
        Jr = so3_jr(w * dts[:, s][:, None])
        RS = dR @ hat(a)
        A = np.tile(np.eye(9), (E, 1, 1))
        A[:, 6:9, 6:9] = np.swapaxes(dRk, 1, 2)
        A[:, 3:6, 6:9] = -RS * dt
        A[:, 0:3, 6:9] = -0.5 * RS * dt2
        A[:, 0:3, 3:6] = I3 * dt
        Bg = np.zeros((E, 9, 3)); Bg[:, 6:9] = Jr * dt
        Ca = np.zeros((E, 9, 3)); Ca[:, 3:6] = dR * dt; Ca[:, 0:3] = 0.5 * dR * dt2
        cov = A @ cov @ np.swapaxes(A, 1, 2) + abi.GYR_MEAS_COV * (Bg @ np.swapaxes(Bg, 1, 2)) \
            + abi.ACC_MEAS_COV * (Ca @ np.swapaxes(Ca, 1, 2))
        RSJ = RS @ JRg
        JPa = JPa + JVa * dt - 0.5 * dR * dt2
        JPg = JPg + JVg * dt - 0.5 * RSJ * dt2
        JVa = JVa - dR * dt
        JVg = JVg - RSJ * dt
        JRg = np.swapaxes(dRk, 1, 2) @ JRg - Jr * dt
        Ra = np.einsum("eij,ej->ei", dR, a)
        dP = dP + dV * dts[:, s][:, None] + 0.5 * Ra * (dts[:, s] ** 2)[:, None]
        dV = dV + Ra * dts[:, s][:, None]
        dR = dR @ dRk
        dR = np.stack([_orthonormalise(dR[e]) for e in range(E)])  # normalizeRotationM, IMUPreintegrator.h:163-174
        T = T + dts[:, s]
    meas = np.concatenate([T[:, None], dP, dV, dR.reshape(E, 9), JPg.reshape(E, 9), JPa.reshape(E, 9),
                           JVg.reshape(E, 9), JVa.reshape(E, 9), JRg.reshape(E, 9)], axis=1)
    return meas, cov


def prv_information(cov_pvphi):
    """inverse of the V/phi-swapped covariance (src/Optimizer.cpp:273-280)."""
    perm = [0, 1, 2, 6, 7, 8, 3, 4, 5]
    c = cov_pvphi[..., perm, :][..., :, perm]
    return np.linalg.inv(c)


# ---------------------------------------------------------------- the generator
def make_window(variant=abi.VARIANT_PRV_IDP, n_kf=50, n_fixed=1, n_pt=5000, n_obs=30000, seed=3,
                kf_dt=0.25, imu_dt=0.005, outlier_frac=0.05, algo=None, noise=True, init_scale=1.0, pix_noise=1.0,
                tracks="nearest", landmark_order="random"):
    """Build one synthetic local-BA window.

    n_kf keyframes in time order t0..; the FIRST n_fixed in time are fixed (the window's predecessor and,
    if n_fixed > 1, older co-observers).  In the returned Problem free keyframes come first (time order),
    fixed ones after.  n_obs counts reprojection EDGES (variant 2: the reference keyframe's own
    observation is not an edge, src/Optimizer.cpp:395-398).  init_scale multiplies the perturbation of the initial
    guess (poses, velocities, depths / points): < 1 = a window that starts close to its optimum.  pix_noise multiplies the
    keypoint noise (1 = one pixel at octave 0, ORB-SLAM's model); the information matrices are not changed, so chi2 scales with
    pix_noise^2 and the absolute |dchi2| < 1e-3 stop of Gauss-Newton is reached after fewer iterations for sharp features.
    tracks: which of the keyframes that see a landmark observe it -- "nearest" = the ones closest in time to the keyframe it was
    created from (tracks are runs of consecutive keyframes: a sliding window), "random" = a random subset of them (tracks with
    gaps, co-visibility scattered over the whole span in which the point is in view: a co-visibility window).
    """
    rng = np.random.default_rng(seed)
    R_bc, p_bc, T_cb = extrinsics()
    R_cb = R_bc.T
    S = int(round(kf_dt / imu_dt))
    n_steps = (n_kf - 1) * S
    t = np.arange(n_steps) * imu_dt

    # gravity: (0,0,-g) rotated by a fixed seed rotation
    g_w = so3_exp(rng.normal(0, 0.3, 3)) @ np.array([0.0, 0.0, -GRAVITY])
    # smooth body-rate and world-acceleration signals (MAV-like: ~1 m/s, ~0.3 rad/s)
    fw = rng.uniform(0.15, 0.6, 3); pw = rng.uniform(0, 2 * np.pi, 3); aw = rng.uniform(0.1, 0.3, 3)
    fa = rng.uniform(0.1, 0.4, 3); pa = rng.uniform(0, 2 * np.pi, 3); aa = np.array([1.0, 1.0, 0.3]) * rng.uniform(0.5, 1.2, 3)
    omega_true = aw * np.sin(2 * np.pi * fw * t[:, None] + pw)
    acc_world = aa * np.sin(2 * np.pi * fa * t[:, None] + pa)
    # discrete ground truth with the reference's integration rule
    R = np.eye(3); V = np.array([1.0, 0.3, 0.0]) * rng.uniform(0.6, 1.2); P = np.zeros(3)
    Rs, Vs, Ps = [R.copy()], [V.copy()], [P.copy()]
    acc_true = np.zeros((n_steps, 3))
    for k in range(n_steps):
        a_b = R.T @ (acc_world[k] - g_w)
        acc_true[k] = a_b
        P = P + V * imu_dt + 0.5 * g_w * imu_dt ** 2 + 0.5 * (R @ a_b) * imu_dt ** 2
        V = V + g_w * imu_dt + (R @ a_b) * imu_dt
        R = _orthonormalise(R @ so3_exp(omega_true[k] * imu_dt))
        if (k + 1) % S == 0:
            Rs.append(R.copy()); Vs.append(V.copy()); Ps.append(P.copy())
    Rs, Vs, Ps = np.array(Rs), np.array(Vs), np.array(Ps)   # time order, n_kf entries

    bg_true = rng.normal(0, 1e-3, 3) if noise else np.zeros(3)
    ba_true = rng.normal(0, 2e-2, 3) if noise else np.zeros(3)
    bg_est = bg_true + (rng.normal(0, 2e-4, 3) if noise else 0)
    ba_est = ba_true + (rng.normal(0, 5e-3, 3) if noise else 0)
    sg = 1.7e-4 / np.sqrt(imu_dt) if noise else 0.0
    sa = 2.0e-3 / np.sqrt(imu_dt) if noise else 0.0
    gyr_meas = omega_true + bg_true + rng.normal(0, 1, omega_true.shape) * sg
    acc_meas = acc_true + ba_true + rng.normal(0, 1, acc_true.shape) * sa

    # camera poses (ground truth): R_cw = (R_wb R_bc)^T, p_wc = R_wb p_bc + p_wb  (src/KeyFrame.cpp:101-106)
    R_wc = Rs @ R_bc
    p_wc = np.einsum("kij,j->ki", Rs, p_bc) + Ps
    fx, fy, cx, cy = EUROC_K
    Wd, Hd = EUROC_WH

    def project(Pw, k):
        Pc = R_wc[k].T @ (Pw - p_wc[k])
        return np.array([fx * Pc[0] / Pc[2] + cx, fy * Pc[1] / Pc[2] + cy]), Pc[2]

    # how many edges each point gets (sum exactly n_obs, each >= 1 for IDP / >= 2 otherwise)
    extra = 1 if variant == abi.VARIANT_PRV_IDP else 0   # the reference KF's own observation
    kmin = 1 if variant == abi.VARIANT_PRV_IDP else 2
    base = n_obs // n_pt
    ks = np.full(n_pt, base, dtype=np.int64)
    ks[: n_obs - base * n_pt] += 1
    jmax = max(0, min(2, base - kmin, n_kf - extra - base - 1))
    if base + (1 if n_obs % n_pt else 0) + extra > n_kf:
        raise ValueError("n_obs/n_pt too large for n_kf keyframes")
    jit = rng.integers(-jmax, jmax + 1, n_pt // 2)
    ks[: 2 * (n_pt // 2): 2] += jit
    ks[1: 2 * (n_pt // 2): 2] -= jit
    rng.shuffle(ks)
    assert ks.sum() == n_obs and ks.min() >= kmin

    R_cw_all = np.swapaxes(R_wc, 1, 2)

    def project_all(Pw):
        """pixel and depth of world point Pw in every keyframe."""
        Pc = np.einsum("kij,kj->ki", R_cw_all, Pw[None, :] - p_wc)
        z = Pc[:, 2]
        zs = np.where(np.abs(z) < 1e-9, 1e-9, z)
        return np.stack([fx * Pc[:, 0] / zs + cx, fy * Pc[:, 1] / zs + cy], axis=1), z

    pts_w = np.zeros((n_pt, 3)); obs_lists = []
    kf_ids = np.arange(n_kf)
    for p in range(n_pt):
        need = int(ks[p]) + extra
        for _try in range(500):
            c = int(rng.integers(0, n_kf))
            u, v = rng.uniform(20, Wd - 20), rng.uniform(20, Hd - 20)
            d = rng.uniform(2.0, 10.0)
            Pw = R_wc[c] @ (np.array([(u - cx) / fx, (v - cy) / fy, 1.0]) * d) + p_wc[c]
            uv, z = project_all(Pw)
            ok = (z > 0.5) & (uv[:, 0] > 5) & (uv[:, 0] < Wd - 5) & (uv[:, 1] > 5) & (uv[:, 1] < Hd - 5)
            cand = kf_ids[ok]
            if cand.size < need:
                continue
            if tracks == "random":
                vis = rng.choice(cand, need, replace=False)
            else:
                vis = cand[np.argsort(np.abs(cand - c), kind="stable")[:need]]
            if (vis >= n_fixed).any():
                break
        else:
            raise RuntimeError("could not place point")
        pts_w[p] = Pw
        obs_lists.append(np.sort(vis))

    # index map: time index -> problem index (free first)
    n_free = n_kf - n_fixed
    tidx = list(range(n_fixed, n_kf)) + list(range(n_fixed))
    pidx = np.zeros(n_kf, dtype=np.int64)
    pidx[tidx] = np.arange(n_kf)

    if landmark_order == "caller":
        # the order the reference's caller hands landmarks over in: lLocalMapPoints is filled keyframe by keyframe over
        # lLocalKeyFrames (oldest first, the current keyframe last), every keyframe appending the map points no earlier one has
        # listed (src/Optimizer.cpp:59-78) -- landmarks come grouped by the FIRST local keyframe that observes them; inside a
        # group the order is the keyframe's feature order (arbitrary: the generation order here)
        key = np.array([min(int(pidx[k]) for k in l if pidx[k] < n_free) for l in obs_lists])
        perm = np.argsort(key, kind="stable")
        pts_w = pts_w[perm]
        obs_lists = [obs_lists[i] for i in perm]
    elif landmark_order != "random":
        raise ValueError("landmark_order: 'random' or 'caller'")

    # observations (vectorised): edge list in point order, ascending keyframe time inside a point
    first = 1 if variant == abi.VARIANT_PRV_IDP else 0
    e_pt = np.concatenate([np.full(len(l) - first, p) for p, l in enumerate(obs_lists)])
    e_tk = np.concatenate([l[first:] for l in obs_lists])
    pt_obs_begin = np.concatenate([[0], np.cumsum([len(l) - first for l in obs_lists])])

    def measure(pt_idx, tk_idx):
        Pc = np.einsum("eij,ej->ei", R_cw_all[tk_idx], pts_w[pt_idx] - p_wc[tk_idx])
        uv = np.stack([fx * Pc[:, 0] / Pc[:, 2] + cx, fy * Pc[:, 1] / Pc[:, 2] + cy], axis=1)
        n = len(pt_idx)
        octave = np.minimum(rng.geometric(0.45, n) - 1, 7)
        w = np.float32(1.0 / (1.2 ** (2 * octave))).astype(np.float64)
        out = np.zeros(n, dtype=bool)
        if noise:
            uv = uv + rng.normal(0, 1, (n, 2)) * (1.2 ** octave)[:, None] * pix_noise
            out = rng.uniform(size=n) < outlier_frac
            uv = uv + out[:, None] * rng.choice([-1.0, 1.0], (n, 2)) * rng.uniform(15, 25, (n, 2))
        return np.float32(uv).astype(np.float64), w, out, Pc[:, 2]

    obs_uv, obs_w, is_outlier, _ = measure(e_pt, e_tk)
    obs_kf = pidx[e_tk]
    if variant == abi.VARIANT_PRV_IDP:
        ref_tk = np.array([l[0] for l in obs_lists])
        _of = outlier_frac
        outlier_frac = 0.0   # the reference observation defines the bearing: noisy but never a gross outlier
        uvr, _w, _o, depth = measure(np.arange(n_pt), ref_tk)
        outlier_frac = _of
        depth0 = depth * (1 + (init_scale * rng.normal(0, 0.03, n_pt) if noise else 0.0))
        pt_arr = np.stack([1.0 / depth0, (uvr[:, 0] - cx) / fx, (uvr[:, 1] - cy) / fy], axis=1)
        pt_ref = pidx[ref_tk]
    else:
        pt_arr = pts_w + (init_scale * rng.normal(0, 0.05, (n_pt, 3)) if noise else 0)
        if variant == abi.VARIANT_SE3_XYZ:
            pt_arr = np.float32(pt_arr).astype(np.float64)
        pt_ref = np.zeros(n_pt, dtype=np.int64)

    # keyframe states: ground truth + initial-guess perturbation (free ones only)
    pose = np.zeros((n_kf, 7)); vel = np.zeros((n_kf, 3)); bias = np.zeros((n_kf, 12))
    pose_gt = np.zeros((n_kf, 7))
    for tk in range(n_kf):
        i = pidx[tk]
        Rk, Pk, Vk = Rs[tk], Ps[tk], Vs[tk]
        if variant == abi.VARIANT_SE3_XYZ:
            Rcw = R_wc[tk].T
            pose_gt[i] = np.concatenate([-Rcw @ p_wc[tk], rot_to_quat(Rcw)])
        else:
            pose_gt[i] = np.concatenate([Pk, rot_to_quat(Rk)])
        if i < n_free and noise:
            Rn = Rk @ so3_exp(init_scale * rng.normal(0, np.deg2rad(0.5), 3))
            Pn = Pk + init_scale * rng.normal(0, 0.02, 3)
            Vn = Vk + init_scale * rng.normal(0, 0.05, 3)
        else:
            Rn, Pn, Vn = Rk, Pk, Vk
        if variant == abi.VARIANT_SE3_XYZ:
            Rcw = (Rn @ R_bc).T
            tcw = -Rcw @ (Rn @ p_bc + Pn)
            # the vision path stores poses as float32 cv::Mat (include/Converter.h:56-60)
            T = np.eye(4); T[:3, :3] = Rcw; T[:3, 3] = tcw
            T = np.float32(T).astype(np.float64)
            pose[i] = np.concatenate([T[:3, 3], rot_to_quat(T[:3, :3])])
        else:
            pose[i] = np.concatenate([Pn, rot_to_quat(Rn)])
        vel[i] = Vn
        bias[i, 0:3] = bg_est; bias[i, 3:6] = ba_est

    prob_kw = dict(variant=variant, n_kf_free=n_free, kf_pose=pose, pt=np.array(pt_arr),
                   pt_obs_begin=pt_obs_begin, obs_kf=obs_kf, obs_uv=np.array(obs_uv), obs_w=obs_w, K=EUROC_K)
    if variant != abi.VARIANT_SE3_XYZ:
        # one PRV + one bias edge per consecutive keyframe pair whose later KF is free
        ei, ej = [], []
        om, ac, dts = [], [], []
        for tk in range(1, n_kf):
            if pidx[tk] >= n_free:
                continue   # edges between two fixed keyframes are never built (src/Optimizer.cpp:251-312)
            ei.append(pidx[tk - 1]); ej.append(pidx[tk])
            sl = slice((tk - 1) * S, tk * S)
            g = gyr_meas[sl] - bg_est
            a = acc_meas[sl] - ba_est
            # KeyFrame::ComputePreInt (src/KeyFrame.cpp:195-252): first sample integrated over t_imu0 - t_prevKF
            # (= 0 here) and then again inside the loop
            om.append(np.concatenate([g[:1], g])); ac.append(np.concatenate([a[:1], a]))
            dts.append(np.concatenate([[0.0], np.full(S, imu_dt)]))
        meas, cov = preintegrate(np.array(om), np.array(ac), np.array(dts))
        prob_kw.update(kf_vel=vel, kf_bias=bias, T_cb=T_cb, g_w=g_w, imu_kf_i=ei, imu_kf_j=ej, imu_meas=meas,
                       imu_info_prv=prv_information(cov).reshape(-1, 81))
    if variant == abi.VARIANT_PRV_IDP:
        prob_kw.update(pt_ref_kf=pt_ref, depth_min=0.01, algo=abi.ALGO_GN)
    else:
        prob_kw.update(depth_min=0.0, algo=abi.ALGO_LM)
    if algo is not None:
        prob_kw["algo"] = algo
    prob = abi.Problem(**prob_kw)
    prob.truth = dict(pose=pose_gt, vel=np.array([Vs[tk] for tk in tidx]), pts_w=pts_w, bg=bg_true, ba=ba_true,
                      bg_est=bg_est, ba_est=ba_est, is_outlier=np.array(is_outlier, dtype=bool), time_index=tidx,
                      R_bc=R_bc, p_bc=p_bc)
    return prob


# the BASELINE.json configurations
def config_c2(seed=2, landmark_order="random"):
    """Vision-only LocalBundleAdjustment: 20 KF / 2k MapPoints / ~12k EdgeSE3ProjectXYZ, LM."""
    return make_window(abi.VARIANT_SE3_XYZ, n_kf=20, n_fixed=2, n_pt=2000, n_obs=12000, seed=seed, landmark_order=landmark_order)


def config_c3(seed=3, n_kf=50, n_pt=5000, n_obs=30000, landmark_order="random"):
    """LocalBAPRVIDP: 50 KF (49 free + fixed predecessor) / 5k pts / 30k EdgePRIDP + 49 PRV + 49 bias, GN."""
    return make_window(abi.VARIANT_PRV_IDP, n_kf=n_kf, n_fixed=1, n_pt=n_pt, n_obs=n_obs, seed=seed, landmark_order=landmark_order)


def config_c3_ragged(seed=3, landmark_order="random", kinds=True):
    """A LocalBAPRVIDP window whose size is drawn around BASELINE configs[2]: 40..60 keyframes (mean 50), 100 landmarks per
    keyframe, 6 edges per landmark (mean 5 000 / 30 000) -- so that the windows of a batch differ in size, co-visibility and
    ITERATION COUNTS the way the windows of a real session do (everything depends on the seed only).  Three kinds of window:
    60 %: one-pixel keypoint noise and 2..8 % gross outliers (the window still holds the mismatches of its newest keyframes):
          Gauss-Newton runs 5 + 3 iterations;
    20 %: no gross outliers left (earlier passes erased them), keypoint noise 0.3 px: 4..5 + 2;
    20 %: no gross outliers, 0.1 px: 3..4 + 1 (the |dchi2| < 1e-3 stop is absolute, so it comes earlier where chi2 is small).
    kinds=False: the iteration mix of round 2 -- every window of the first kind (the same sizes, all 5 + 3), so that a rate can be
    compared like for like across rounds (bench.py --iteration-mix r2)."""
    r = np.random.default_rng(1000003 * seed + 17)
    n_kf = int(r.integers(40, 61))
    outlier_frac = float(r.uniform(0.02, 0.08))
    kind = float(r.uniform())
    pix = 1.0
    if kinds and kind >= 0.8:
        outlier_frac, pix = 0.0, 0.1
    elif kinds and kind >= 0.6:
        outlier_frac, pix = 0.0, 0.3
    return make_window(abi.VARIANT_PRV_IDP, n_kf=n_kf, n_fixed=1, n_pt=100 * n_kf, n_obs=600 * n_kf, seed=seed,
                       outlier_frac=outlier_frac, pix_noise=pix, landmark_order=landmark_order)


def config_c3s(seed=3, landmark_order="random"):
    """The LocalBAPRVIDP window of configs[2] with SCATTERED co-visibility: 50 free keyframes + 8 fixed older
    co-observers (src/Optimizer.cpp:199-232: every keyframe outside the window that sees a window landmark is added as a fixed
    vertex), 5 000 landmarks / 30 000 edges whose tracks are random subsets of the keyframes that see them (gaps instead of runs of
    consecutive keyframes), 10-20 % of the landmarks anchored in a fixed reference keyframe."""
    return make_window(abi.VARIANT_PRV_IDP, n_kf=58, n_fixed=8, n_pt=5000, n_obs=30000, seed=seed, tracks="random", landmark_order=landmark_order)


def config_c2s(seed=2, landmark_order="random"):
    """configs[1] with scattered co-visibility: the vision-only LocalBundleAdjustment(KeyFrame*, bool*, Map*, LocalMapping*) takes the
    co-visibility set of the current keyframe as its window (src/Optimizer.cpp:3861-3875), not a run of consecutive keyframes"""
    return make_window(abi.VARIANT_SE3_XYZ, n_kf=20, n_fixed=2, n_pt=2000, n_obs=12000, seed=seed, tracks="random", landmark_order=landmark_order)


def config_c4(seed=4, landmark_order="random"):
    """Synthetic VI graph: 200 KF / 50k pts / 500k obs + IMU chain."""
    return make_window(abi.VARIANT_PRV_IDP, n_kf=200, n_fixed=1, n_pt=50000, n_obs=500000, seed=seed, landmark_order=landmark_order)


def config_gba(seed=6, n_kf=300, n_pt=30000, n_obs=180000, its=10, robust=1):
    """GlobalBundleAdjustmentNavStatePRV at map scale: every keyframe free except PR/Bias of keyframe 0,
    XYZ landmarks, one optimize(its) with Levenberg-Marquardt."""
    p = make_window(abi.VARIANT_PRV_XYZ, algo=abi.ALGO_LM, n_kf=n_kf, n_fixed=0, n_pt=n_pt, n_obs=n_obs, seed=seed, outlier_frac=0.02)
    p.protocol, p.robust, p.its_stage1, p.its_stage2 = abi.PROTO_SINGLE, robust, its, 0
    p.huber_vis = float(np.float32(np.sqrt(5.99)))
    p.kf_fix = np.zeros(p.n_kf, np.uint8); p.kf_fix[0] = 0b101
    return p


def make_frame(seed=5, n_obs=300, last_is_frame=False, compute_marg=True, noise=True, outlier_frac=0.1, pt_noise=0.01):
    """One tracked frame for the IMU-aided PoseOptimization (src/Optimizer.cpp:1671-2317): the last keyframe / frame,
    the preintegration up to the current frame, n_obs map points seen by both, a perturbed initial state."""
    w = make_window(abi.VARIANT_PRV_XYZ, n_kf=3, n_fixed=1, n_pt=n_obs, n_obs=3 * n_obs, seed=seed, noise=noise,
                    outlier_frac=outlier_frac)
    rng = np.random.default_rng(seed + 7919)
    LAST, CUR = 0, 1                     # problem rows of time 1 and time 2
    k = [i for i in range(w.n_imu) if w.imu_kf_i[i] == LAST and w.imu_kf_j[i] == CUR][0]
    perm = [0, 1, 2, 6, 7, 8, 3, 4, 5]
    cov = np.linalg.inv(w.imu_info_prv[k].reshape(9, 9))[np.ix_(perm, perm)]
    pts = w.truth["pts_w"] + (rng.normal(0, pt_noise, (n_obs, 3)) if noise else 0.0)
    e_pt = np.repeat(np.arange(w.n_pt), np.diff(w.pt_obs_begin))
    sel_c, sel_l = w.obs_kf == CUR, w.obs_kf == LAST
    nav = lambda i: np.concatenate([w.kf_pose[i], w.kf_vel[i], w.kf_bias[i]])
    gt = w.truth
    nav_last = nav(LAST)
    if not last_is_frame:   # the last keyframe is trusted: ground truth pose
        nav_last[:7] = gt["pose"][LAST]; nav_last[7:10] = gt["vel"][LAST]
    kw = dict(nav=nav(CUR), nav_last=nav_last, obs_pw=pts[e_pt[sel_c]], obs_uv=w.obs_uv[sel_c], obs_w=w.obs_w[sel_c],
              K=w.K, T_cb=w.T_cb, g_w=w.g_w, imu_meas=w.imu_meas[k], imu_cov_pvphi=cov,
              last_is_frame=int(last_is_frame), compute_marg=int(compute_marg))
    if last_is_frame:
        sig = np.array([0.02] * 3 + [0.05] * 3 + [0.01] * 3 + [1e-3] * 3 + [1e-2] * 3)
        L = np.diag(1.0 / sig) + np.tril(rng.normal(0, 0.05, (15, 15)), -1) / sig[None, :]
        prior = nav_last.copy()
        prior[:3] = gt["pose"][LAST][:3] + rng.normal(0, 0.01, 3)
        kw.update(last_pw=pts[e_pt[sel_l]], last_uv=w.obs_uv[sel_l], last_w=w.obs_w[sel_l], prior_nav=prior, prior_info=L.T @ L)
    f = abi.FrameProblem(**kw)
    f.truth = dict(nav=np.concatenate([gt["pose"][CUR], gt["vel"][CUR]]), is_outlier=gt["is_outlier"][sel_c],
                   is_outlier_last=gt["is_outlier"][sel_l])
    return f


def make_frame_vision(seed=1, n_obs=200, noise=True, outlier_frac=0.1, pt_noise=0.01):
    """BASELINE configs[0] (C1): one frame, ~200 monocular correspondences, vision-only Optimizer::PoseOptimization(Frame*)
    (src/Optimizer.cpp:3610-3835).  nav[0..6] = T_cw as SE3Quat, float32-narrowed like pFrame->mTcw; points float32."""
    f = make_frame(seed=seed, n_obs=n_obs, noise=noise, outlier_frac=outlier_frac, pt_noise=pt_noise)
    R_bc, p_bc, _ = extrinsics()

    def tcw(nav):
        Rwb = quat_to_rot(nav[3:7])
        Rcw = (Rwb @ R_bc).T
        t = -Rcw @ (Rwb @ p_bc + nav[:3])
        T = np.float32(np.concatenate([Rcw.reshape(-1), t])).astype(np.float64)
        return np.concatenate([T[9:], rot_to_quat(_orthonormalise(T[:9].reshape(3, 3)))])
    nav = np.zeros(abi.NAV_STRIDE); nav[:7] = tcw(f.nav)
    g = abi.FrameProblem(nav=nav, nav_last=np.zeros(abi.NAV_STRIDE), obs_pw=np.float32(f.obs_pw).astype(np.float64), obs_uv=f.obs_uv,
                         obs_w=f.obs_w, K=f.K, T_cb=f.T_cb, g_w=f.g_w, imu_meas=f.imu_meas, imu_cov_pvphi=f.imu_cov_pvphi,
                         last_is_frame=2, compute_marg=0)
    gt = np.concatenate([f.truth["nav"][:7], np.zeros(3)])
    g.truth = dict(T_cw=tcw(gt), is_outlier=f.truth["is_outlier"])
    return g
