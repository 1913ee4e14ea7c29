"""Throughput of vba_pose_optimize (IMU-aided PoseOptimization) vs the CPU oracle on the same frames."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from mc_slam_amd import synth, backend
import oracle_lib

n_batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
distinct = [synth.make_frame(seed=100 + i, n_obs=300, last_is_frame=bool(i % 2)) for i in range(16)]
frames = [distinct[i % 16] for i in range(n_batch)]
ba = backend.LocalBA(0)
ba.pose_optimize(frames[:16])
for nb in (1, 16, 256, n_batch):
    ts = []
    packed = ba.pose_pack(frames[:nb])
    for _ in range(3):
        for s_, f_ in zip(packed[1], packed[5]):
            s_.nav[:] = f_.nav.tolist()
        t0 = time.perf_counter()
        rc = ba.lib.vba_pose_optimize(ba.h, packed[0], packed[3], packed[4])
        ts.append(time.perf_counter() - t0)
        assert rc == 0
    rs = [b.get(s_) for b, s_ in zip(packed[2], packed[1])]
    print("GPU  batch %5d: %.3f ms per vba_pose_optimize call (gather + H2D + kernel + D2H), %.0f frames/s" % (nb, min(ts) * 1e3, nb / min(ts)), flush=True)
t0 = time.perf_counter()
for f in distinct:
    ro = oracle_lib.pose_optimize(f)
tc = (time.perf_counter() - t0) / len(distinct)
print("CPU oracle: %.3f ms per frame, %.0f frames/s (1 core)" % (tc * 1e3, 1 / tc))
r0 = rs[0]; ro = oracle_lib.pose_optimize(distinct[0])
print("check frame 0: its", r0.its_done, ro.its_done, "dP", np.abs(r0.nav[:3] - ro.nav[:3]).max())
