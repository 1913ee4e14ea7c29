import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from mc_slam_amd import abi, synth, backend
ba = backend.LocalBA(0, hooks=True)
p = synth.make_window(abi.VARIANT_SE3_XYZ, algo=abi.ALGO_LM, n_kf=8, n_fixed=2, n_pt=150, n_obs=800, seed=170)
p.obs_w = np.zeros_like(p.obs_w)
good = synth.make_window(abi.VARIANT_SE3_XYZ, algo=abi.ALGO_LM, n_kf=8, n_fixed=2, n_pt=150, n_obs=800, seed=172)
names = ["S", "LF", "YV", "VEC", "BPOSE", "DVEC", "WINV", "SLOT", "EREC", "PREC", "KFDIR", "PART", "CHI2E", "CHI2F", "DEPTH", "PT", "POSE", "PTBK", "POSEBK", "KFR", "OUTCHI"]
def scan(tag):
    for n in names:
        bid = ba.lib.vba_debug_buf_id(n.encode())
        # sizes: probe by halving
        sz = 1 << 26
        buf = None
        while sz >= 64:
            a = np.zeros(sz // 8)
            if ba.lib.vba_debug_copy(ba.h, bid, C.c_uint64(0), a.ctypes.data_as(C.c_void_p), C.c_uint64(a.nbytes)) == 0:
                buf = a; break
            sz //= 2
        if buf is None: continue
        bad = ~np.isfinite(buf)
        if bad.any():
            idx = np.nonzero(bad)[0]
            print(tag, n, "size>=", sz, "nonfinite", bad.sum(), "first", idx[:6], "last", idx[-1])
ba.upload([good] * 9); ba.run()
q, r = ba.solve(p); print("zero", r.its_done)
scan("after zero run")
ba.upload([good] * 9)
scan("after upload")
ba.run(); q0, r0 = ba.download(); print("batch after zero", [r.its_done for r in r0])
scan("after bad run")
ba.upload([good] * 9)
scan("after 2nd upload")
ba.run(); q0, r0 = ba.download(); print("then", [r.its_done for r in r0])
