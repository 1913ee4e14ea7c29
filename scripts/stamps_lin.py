import sys, os, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
from mc_slam_amd import synth, backend
wins = [synth.config_c3(seed=100 + i) for i in range(4)]
ba = backend.LocalBA(0)
for nb in (1, 512):
    ba.upload([wins[i % 4] for i in range(nb)]); ba.run(); ba.run()
    lib = ba.lib
    lib.vba_debug_buf_id.argtypes = [C.c_char_p]; lib.vba_debug_copy.argtypes = [C.c_void_p, C.c_int32, C.c_uint64, C.c_void_p, C.c_uint64]
    a = np.zeros(16)
    assert lib.vba_debug_copy(ba.h, lib.vba_debug_buf_id(b"DBG"), 0, a.ctypes.data_as(C.c_void_p), a.nbytes) == 0
    for r in range(2):
        t = a[8 * r: 8 * r + 7]
        print("batch %d window %s: phases A,B,C,D,E,sum in wall_clock64 ticks (100 MHz): " % (nb, "0" if r == 0 else "300"), np.diff(t).astype(int), " total", int(t[6] - t[0]))
