import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mc_slam_amd import synth, backend
ba = backend.LocalBA(0)
ba.upload([synth.config_c3(seed=100)])
L = ba.lib
L.vba_debug_buf_id.argtypes = [C.c_char_p]; L.vba_debug_copy.argtypes = [C.c_void_p, C.c_int32, C.c_uint64, C.c_void_p, C.c_uint64]
def get(name, n):
    a = np.zeros(n, dtype=np.int32)
    rc = L.vba_debug_copy(ba.h, L.vba_debug_buf_id(name.encode()), 0, a.ctypes.data_as(C.c_void_p), a.nbytes)
    assert rc == 0, name
    return a
nb = 23
pb = get("TLPANB", nb + 1)
kb = get("TLKB", int(pb[nb]) + nb + 1)
pan = get("TLPAN", int(pb[nb]))
print("npan per step:", list(np.diff(pb)))
print("K(J,J):", [int(kb[pb[J] + J + 1] - kb[pb[J] + J]) for J in range(nb)])
print("sum K over panel tiles per step:", [int(kb[pb[J + 1] + J + 1] - kb[pb[J] + J + 1]) for J in range(nb)])
print("total tile products", int(kb[-1]))
