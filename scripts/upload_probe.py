"""Exploration: cost of vba_batch_upload (first call allocates, later calls reuse the buffers) and of vba_solve on one window."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mc_slam_amd import synth, backend

sizes = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["1", "64", "1024"])]
wins = [synth.config_c3(seed=100 + i) for i in range(8)]
ba = backend.LocalBA(0)
for B in sizes:
    batch = [wins[i % len(wins)] for i in range(B)]
    for rep in range(3):
        t0 = time.time(); ba.upload(batch); tu = time.time() - t0
        print("B=%d upload #%d %.2f ms (%.3f ms per window)" % (B, rep, tu * 1e3, tu * 1e3 / B), flush=True)
    ba.run()
    for rep in range(2):
        t0 = time.time(); ba.download(); td = time.time() - t0
        print("B=%d download #%d %.2f ms (%.3f ms per window, python wrapper included)" % (B, rep, td * 1e3, td * 1e3 / B), flush=True)
for rep in range(4):
    t0 = time.time(); ba.solve(wins[rep]); print("vba_solve one window: %.2f ms" % ((time.time() - t0) * 1e3), flush=True)
