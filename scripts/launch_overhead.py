import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mc_slam_amd import synth, backend
p = synth.config_c3(seed=100)
ba = backend.LocalBA(0); ba.upload([p])
for stop in (0, 1):
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); ba.run(stop=C.c_int(stop)); ts.append(time.perf_counter() - t0)
    print("stop=%d run min %.3f ms (448 launches -> %.2f us per launch when all exit at once)" % (stop, min(ts) * 1e3, min(ts) * 1e6 / 448))
