"""Exploration (diagnostic build -DVBA_STAMPS only): shader-clock cycles per phase of the off-diagonal Schur workgroups."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mc_slam_amd import synth, backend

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
wins = [synth.config_c3(seed=100 + i) for i in range(8)]
ba = backend.LocalBA(0)
ba.upload([wins[i % 8] for i in range(B)])
lib = ba.lib
lib.vba_debug_buf_id.argtypes = [C.c_char_p]
lib.vba_debug_copy.argtypes = [C.c_void_p, C.c_int32, C.c_uint64, C.c_void_p, C.c_uint64]
bid = lib.vba_debug_buf_id(b"DBG")
def rd():
    a = np.zeros(16, dtype=np.uint64)
    assert lib.vba_debug_copy(ba.h, bid, 0, a.ctypes.data_as(C.c_void_p), 128) == 0
    return a
ba.run(); a0 = rd(); ba.run(); a1 = rd()
dlt = (a1 - a0).astype(np.float64)
n = dlt[8]
names = ["pair look-ups", "first item indices", "item loop", "butterfly + LDS", "write block"]
print("off-diagonal workgroups per run: %d" % n)
for k, nm in enumerate(names):
    print("  %-20s %8.0f cycles per workgroup  (%.1f %%)" % (nm, dlt[k] / n, 100 * dlt[k] / dlt[:5].sum()))
