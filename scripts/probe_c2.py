"""run time of a batch of C2 (vision-only, LM) windows"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mc_slam_amd import synth, backend
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
wins = [synth.config_c2(seed=100 + i) for i in range(8)]
ba = backend.LocalBA(0)
ba.upload([wins[i % 8] for i in range(nb)])
ba.run()
ts = []
for _ in range(3):
    t0 = time.time(); ba.run(); ts.append(time.time() - t0)
q, r = ba.download()
ba.set_profile(True); ba.run(); pf = ba.get_profile(); ba.set_profile(False)
print("C2 B=%d run min %.2f ms -> %.0f windows/s its %s" % (nb, min(ts) * 1e3, nb / min(ts), r[0].its_done))
print("   " + "  ".join("%s %.2f ms/%d" % (k, v["ms"], v["launches"]) for k, v in pf.items() if k != "total_ms" and v["launches"]))
