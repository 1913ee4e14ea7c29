"""The host half of the structure build and the problem file reader under AddressSanitizer + UBSan (CPU only): the index
arithmetic that runs before every upload (validation walk, landmark masks, keyframe-pair occupancy, IMU lists, symbolic tile
factorisation, k_lin2 work split) on valid, shuffled, many-keyframe and malformed windows."""
import os
import subprocess

import numpy as np
import pytest

from mc_slam_amd import abi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("hs") / "host_structure_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "host_structure_check.cpp"), "-o", exe])
    return exe


def _run(checker, tmp_path, probs):
    files = []
    for i, p in enumerate(probs):
        f = str(tmp_path / ("w%d.vbap" % i))
        abi.save_problem(f, p)
        files.append(f)
    r = subprocess.run([checker] + files, capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"), timeout=300)
    assert r.returncode == 0 and "ERROR" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == len(probs)
    return lines


def _shuffle(p, seed):
    from test_gpu_parity import _shuffled
    return _shuffled(p, seed, drop_middle=True)


def test_valid_windows_of_every_variant(checker, tmp_path):
    ps = [synth.make_window(abi.VARIANT_PRV_IDP, n_kf=12, n_fixed=1, n_pt=400, n_obs=2400, seed=41),
          synth.make_window(abi.VARIANT_PRV_IDP, n_kf=9, n_fixed=3, n_pt=150, n_obs=700, seed=42),
          synth.make_window(abi.VARIANT_PRV_IDP, n_kf=70, n_fixed=1, n_pt=900, n_obs=5400, seed=44),     # two-word landmark masks
          synth.make_window(abi.VARIANT_SE3_XYZ, algo=abi.ALGO_LM, n_kf=10, n_fixed=2, n_pt=300, n_obs=1800, seed=43),
          synth.make_window(abi.VARIANT_PRV_XYZ, algo=abi.ALGO_LM, n_kf=10, n_fixed=1, n_pt=300, n_obs=1800, seed=45),
          synth.config_c3(seed=3)]
    ps += [_shuffle(p, 7) for p in ps[:4]]
    q = ps[0].copy(); q.solver = abi.SOLVER_PCG      # (the solver is a run-time option: not in the file, the default path is checked)
    for p, line in zip(ps, _run(checker, tmp_path, ps)):
        assert line.startswith("ok "), line
        f = dict(zip(line.split()[1::2], line.split()[2::2]))
        m = np.diff(p.pt_obs_begin).astype(np.int64)
        cap = int((m * (m + 1) // 2).sum()) if p.variant == abi.VARIANT_PRV_IDP else int((m * (m - 1) // 2).sum())
        assert int(f["item_cap"]) == cap and int(f["mask_bits"]) == p.n_obs and int(f["mwords"]) == (p.n_kf + 63) // 64
        n_imu_entries = 0 if p.variant == abi.VARIANT_SE3_XYZ else sum(
            int(i < p.n_kf_free) + int(j < p.n_kf_free) + int(i < p.n_kf_free and j < p.n_kf_free) for i, j in zip(p.imu_kf_i, p.imu_kf_j))
        assert int(f["pimu"]) == n_imu_entries
        assert int(f["tiles"]) == int(f["klist"]) > 0     # one tile product per k-list entry of the left-looking form
    # the caller's order of landmarks / observations changes neither the tile structure nor the masks' content
    lines = _run(checker, tmp_path, [ps[0], ps[6]])
    a, b = (dict(zip(l.split()[1::2], l.split()[2::2])) for l in lines)
    assert a["h_tiles"] != "" and int(a["tiles"]) >= int(b["tiles"]) > 0      # (drop_middle removed observations: not more structure)


def test_malformed_windows_are_refused_with_a_message(checker, tmp_path):
    p = synth.make_window(abi.VARIANT_PRV_IDP, n_kf=8, n_fixed=1, n_pt=100, n_obs=500, seed=46)
    bad = []
    q = p.copy(); q.obs_kf = p.obs_kf.copy(); q.obs_kf[3] = 99; bad.append((q, "obs_kf out of range"))
    q = p.copy(); q.obs_kf = p.obs_kf.copy(); q.obs_kf[3] = -1; bad.append((q, "obs_kf out of range"))
    q = p.copy(); q.pt_ref_kf = p.pt_ref_kf.copy(); q.pt_ref_kf[5] = 8; bad.append((q, "pt_ref_kf out of range"))
    q = p.copy(); q.obs_kf = p.obs_kf.copy(); o = p.pt_obs_begin[7]; q.obs_kf[o + 1] = q.obs_kf[o]; bad.append((q, "observed twice"))
    q = p.copy(); q.obs_kf = p.obs_kf.copy(); o = p.pt_obs_begin[9]; q.obs_kf[o] = p.pt_ref_kf[9]; bad.append((q, "reference keyframe is not an edge"))
    q = p.copy(); q.pt_obs_begin = p.pt_obs_begin.copy(); q.pt_obs_begin[4] = -3; bad.append((q, "not a valid CSR"))
    q = p.copy(); q.imu_kf_i = p.imu_kf_i.copy(); q.imu_kf_i[2] = 50; bad.append((q, "imu keyframe index out of range"))
    q = p.copy(); q.imu_kf_j = p.imu_kf_j.copy(); q.imu_kf_j[2] = q.imu_kf_i[2]; bad.append((q, "imu keyframe index out of range"))
    for (q, msg), line in zip(bad, _run(checker, tmp_path, [b[0] for b in bad])):
        assert line.startswith("error ") and msg in line, (msg, line)
    # a truncated / corrupt file is refused by the reader, not read out of bounds
    f = str(tmp_path / "t.vbap")
    abi.save_problem(f, p)
    raw = open(f, "rb").read()
    for cut in (10, 200, len(raw) - 9):
        open(f, "wb").write(raw[:cut])
        r = subprocess.run([checker, f], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0 and r.stdout.startswith("error load") and "ERROR" not in r.stderr, (cut, r.stdout, r.stderr[-500:])
    hdr = bytearray(raw); hdr[8 + 4 * 3:8 + 4 * 4] = (2 ** 31 - 1).to_bytes(4, "little")      # n_pt = INT_MAX in the header
    open(f, "wb").write(bytes(hdr))
    r = subprocess.run([checker, f], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.startswith("error load") and "ERROR" not in r.stderr, (r.stdout, r.stderr[-500:])
