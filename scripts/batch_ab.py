"""many-window A/B: windows/s of vba_batch_run on N replicated ragged C3 windows + per-class profile"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from mc_slam_amd import synth, backend
import oracle_lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
nd = int(sys.argv[2]) if len(sys.argv) > 2 else 8
order = os.environ.get("AB_LM_ORDER", "random")   # "caller": landmarks grouped by first local keyframe (src/Optimizer.cpp:59-78)
wins = [synth.config_c3_ragged(100 + i, landmark_order=order) for i in range(nd)]
ba = backend.LocalBA(0)
ba.upload([wins[i % nd] for i in range(n)])
ba.run(); ba.run()
t0 = time.perf_counter()
for _ in range(4): ba.run()
dt = (time.perf_counter() - t0) / 4
sol, res = ba.download()
ok = True
for i in range(nd):
    qo, ro = oracle_lib.solve(wins[i])
    ok = ok and ro.its_done == res[i].its_done and abs(ro.chi2_vis - res[i].chi2_vis) <= 1e-9 * ro.chi2_vis and (ro.obs_outlier == res[i].obs_outlier).all() and np.abs(qo.kf_pose[:, :3] - sol[i].kf_pose[:, :3]).max() < 1e-8
ba.set_profile(True); ba.run(); pf = ba.get_profile(); ba.set_profile(False)
print("%d windows: %.1f ms/step = %.0f windows/s  parity %s  its %s  classes %s  (env %s)" % (n, dt * 1e3, n / dt, ok, [r.its_done for r in res[:nd]],
      {k: round(v["ms"], 1) for k, v in pf.items() if k != "total_ms" and v["launches"]}, {k: v for k, v in os.environ.items() if k.startswith("VBA_")}))
