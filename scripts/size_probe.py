"""Exploration: per-class time of C3-shaped windows of several sizes (does the Schur pass speed up once a window's records fit the L2?)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mc_slam_amd import synth, backend

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ba = backend.LocalBA(0)
for n_pt, n_obs in ((1250, 7500), (2500, 15000), (3750, 22500), (5000, 30000), (7500, 45000)):
    wins = [synth.config_c3(seed=100 + i, n_pt=n_pt, n_obs=n_obs) for i in range(8)]
    ba.upload([wins[i % 8] for i in range(B)])
    ba.set_profile(False); ba.run()
    ba.set_profile(True); ba.run(); pf = ba.get_profile(); ba.set_profile(False)
    print("n_obs %6d: " % n_obs + "  ".join("%s %.2f/%d" % (k, v["ms"], v["launches"]) for k, v in pf.items() if k != "total_ms" and v["launches"])
          + "   schur us per launch per 1k obs: %.3f" % (pf["schur"]["ms"] * 1e3 / pf["schur"]["launches"] / (n_obs / 1e3)), flush=True)
