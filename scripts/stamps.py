import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from mc_slam_amd import synth, backend
p = synth.config_c3(seed=3)
ba = backend.LocalBA(0, hooks=True)
for _ in range(3): ba.solve(p)
bid = ba.lib.vba_debug_buf_id(b"DBG")
a = np.zeros(32)
assert ba.lib.vba_debug_copy(ba.h, bid, C.c_uint64(0), a.ctypes.data_as(C.c_void_p), C.c_uint64(256)) == 0
t = a[:6] - a[0]
print("stamps (shader cycles from kernel start): loads landed %.0f, elimination done %.0f, factor stores issued %.0f, MFMA + stores issued %.0f, stores drained %.0f" % tuple(t[1:6]))
u = a[16:23] - a[16]
print("k_trsv column 10 (cycles from column start): gather + partials written %.0f, barrier passed %.0f, tile column in registers + division + partials summed %.0f, 32-step solve done %.0f, second barrier passed %.0f (prefetch issued %.0f)" % tuple(u[1:7]))
