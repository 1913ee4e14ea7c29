"""In-kernel stamps of k_chol_chain (diagnostic build: scripts/stamps.sh, VBA_LIB=ab/stamps.so): per chain column, wave 0 and wave 2."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from mc_slam_amd import synth, backend
p = synth.config_c3(seed=3)
ba = backend.LocalBA(0, hooks=True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
if n == 1:
    for _ in range(3): ba.solve(p)
else:
    ba.upload([p] * n); ba.run(); ba.run()
bid = ba.lib.vba_debug_buf_id(b"DBG")
a = np.zeros(512)
assert ba.lib.vba_debug_copy(ba.h, bid, C.c_uint64(0), a.ctypes.data_as(C.c_void_p), C.c_uint64(4096)) == 0
t0 = a[60]
print("prologue done (cycles from kernel start): %.0f" % (a[61] - t0))
for wv, base in ((0, 64), (int(os.environ.get("STAMP_W2", "2")), 256)):
    print("wave %d: column: start | loaded | elim done | at A | past A | at B | past B | products done  (cycles from kernel start)" % wv)
    for J in range(14):
        v = a[base + 8 * J: base + 8 * J + 8]
        if v[0] == 0: break
        print("  J=%2d " % J + " ".join("%7.0f" % (x - t0) if x else "      -" for x in v))

print("phase F of wave 1 (cycles from past A): LDS operands in registers | S quadrants there | MFMAs done")
for J in range(13):
    v = a[448 + 4 * J: 448 + 4 * J + 3]
    if v[0] == 0: break
    print("  J=%2d " % J + " ".join("%7.0f" % (x - a[256 + 8 * J + 4]) for x in v))
