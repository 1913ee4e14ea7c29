"""Quick A/B on the GPU box: one fresh window (vba_solve wall time), and a resident batch (vba_batch_run) with the class profile.
usage: python scripts/quick_ab.py [n_windows] [reps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mc_slam_amd import synth, backend

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
order = sys.argv[3] if len(sys.argv) > 3 else "caller"
p = synth.config_c3(seed=3, landmark_order=order)
ba = backend.LocalBA(0)
ts = []
for i in range(25):
    q = p.copy()
    st = q.as_struct()
    rb = backend.abi.ResultBuf(q.n_obs)
    import ctypes as C
    t0 = time.perf_counter()
    ba.lib.vba_solve(ba.h, C.byref(st), C.byref(rb.s), None)
    ts.append(time.perf_counter() - t0)
ts = np.array(ts[4:]) * 1e3
print("single window: median %.3f ms  min %.3f ms" % (np.median(ts), ts.min()))
wins = [synth.config_c3(seed=100 + i, landmark_order=order) for i in range(16)]
batch = [wins[i % 16] for i in range(n)]
ba.upload(batch)
ba.run()
t0 = time.perf_counter()
for _ in range(reps):
    ba.run()
dt = (time.perf_counter() - t0) / reps
print("%d windows resident: %.2f ms per run, %.0f windows/s" % (n, dt * 1e3, n / dt))
ba.set_profile(True)
ba.run()
pr = ba.get_profile()
print("classes (ms, launches):", {k: (round(v["ms"], 2), int(v["launches"])) for k, v in pr.items() if isinstance(v, dict)}, "total", round(pr["total_ms"], 2))
