"""single-window latency A/B: median wall time of vba_solve on the C3 window (seed 3) + parity against the oracle"""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from mc_slam_amd import abi, synth, backend
import oracle_lib
p = synth.config_c3(seed=3)
ba = backend.LocalBA(0)
ts = []
for k in range(24):
    q = p.copy(); s = q.as_struct(); rb = abi.ResultBuf(q.n_obs)
    t1 = time.perf_counter()
    assert ba.lib.vba_solve(ba.h, C.byref(s), C.byref(rb.s), None) == 0
    ts.append(time.perf_counter() - t1)
ts = sorted(ts[3:])
r = rb.get()
qo, ro = oracle_lib.solve(p)
ok = (ro.its_done == r.its_done and abs(ro.chi2_vis - r.chi2_vis) <= 1e-9 * ro.chi2_vis and (ro.obs_outlier == r.obs_outlier).all()
      and np.abs(qo.kf_pose[:, :3] - q.kf_pose[:, :3]).max() <= 1e-9)
print("single window: median %.3f ms  min %.3f ms  its %s  parity %s  (env %s)" % (ts[len(ts) // 2] * 1e3, ts[0] * 1e3, r.its_done, ok,
      {k: v for k, v in os.environ.items() if k.startswith("VBA_")}))
