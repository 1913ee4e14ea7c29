import sys, os, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
from mc_slam_amd import synth, backend
p = synth.config_c3(seed=100); p.its_stage1, p.its_stage2 = 1, 0
ba = backend.LocalBA(0); ba.upload([p]); ba.run(); ba.run()
lib = ba.lib
lib.vba_debug_buf_id.argtypes = [C.c_char_p]; lib.vba_debug_copy.argtypes = [C.c_void_p, C.c_int32, C.c_uint64, C.c_void_p, C.c_uint64]
a = np.zeros(8 * 23)
assert lib.vba_debug_copy(ba.h, lib.vba_debug_buf_id(b"DBG"), 0, a.ctypes.data_as(C.c_void_p), a.nbytes) == 0
a = a.reshape(23, 8)[:, :4]
np.set_printoptions(linewidth=200, suppress=True)
print("cycles per phase [load, LDL, TRSM, MFMA+store] per step (s_memtime ticks = 100 MHz? or shader clock):")
print(a.astype(int))
print("mean", a.mean(axis=0))
