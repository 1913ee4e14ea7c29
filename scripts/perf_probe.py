"""Exploration: run-time and per-class breakdown of the HIP backend at several batch sizes (GPU box)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mc_slam_amd import synth, backend

nwin = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sizes = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["1", "4"])]
t0 = time.time()
wins = [synth.config_c3(seed=100 + i) for i in range(nwin)]
print("generated %d windows in %.1fs" % (nwin, time.time() - t0), flush=True)
ba = backend.LocalBA(0)
for B in sizes:
    batch = [wins[i % nwin] for i in range(B)]
    t0 = time.time(); ba.upload(batch); tu = time.time() - t0
    ba.set_profile(False)
    ba.run()
    ts = []
    for _ in range(5):
        t0 = time.time(); ba.run(); ts.append(time.time() - t0)
    q, r = ba.download()
    ba.set_profile(True); ba.run(); pf = ba.get_profile(); ba.set_profile(False)
    print("B=%d upload %.1f ms  run min %.2f ms median %.2f ms -> %.1f windows/s   its %s" % (
        B, tu * 1e3, min(ts) * 1e3, np.median(ts) * 1e3, B / min(ts), [x.its_done for x in r[:4]]), flush=True)
    print("   profile total %.2f ms: " % pf["total_ms"] + "  ".join("%s %.2f ms/%d" % (k, v["ms"], v["launches"]) for k, v in pf.items() if k != "total_ms" and v["launches"]), flush=True)
