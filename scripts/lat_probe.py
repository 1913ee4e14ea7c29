"""single-window end-to-end latency of vba_solve (upload + run + download) through the C-ABI"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes as C
from mc_slam_amd import synth, backend, abi
p = synth.config_c3(seed=100)
ba = backend.LocalBA(0)
for _ in range(3):
    ba.solve(p)
ts = []
for _ in range(10):
    q = p.copy(); s = q.as_struct(); rb = abi.ResultBuf(q.n_obs)
    t0 = time.perf_counter()
    ba.lib.vba_solve(ba.h, C.byref(s), C.byref(rb.s), None)
    ts.append(time.perf_counter() - t0)
print("vba_solve C3 end-to-end: min %.2f ms median %.2f ms" % (min(ts) * 1e3, np.median(ts) * 1e3))
