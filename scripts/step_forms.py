"""cross-check of the factorisation step forms (vba_debug_set_chol_step): same problems, every form, results compared bit for bit
with the first form listed, and timed"""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from mc_slam_amd import abi, synth, backend
ba = backend.LocalBA(0, hooks=True)
ba.lib.vba_debug_set_chol_step.argtypes = [C.c_void_p, C.c_int32]
forms = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "4,1").split(",")]
cases = {"c3 x1": [synth.config_c3(seed=3)], "c3 x20": [synth.config_c3(seed=10 + i) for i in range(20)],
         "c3_ragged x100": [synth.config_c3_ragged(seed=500 + i) for i in range(100)], "c4 x2": [synth.config_c4(seed=1), synth.config_c4(seed=2)]}
for name, ps in cases.items():
    ref = None
    for f in forms:
        ba.lib.vba_debug_set_chol_step(ba.h, f)
        ts = []
        for rep in range(4):
            ba.upload(ps); t0 = time.perf_counter(); ba.run(); ts.append(time.perf_counter() - t0)
        qs, rs = ba.download()
        sig = (np.concatenate([q.kf_pose.ravel() for q in qs]), [r.its_done for r in rs], [r.chi2_vis for r in rs])
        if ref is None: ref = sig
        same = np.array_equal(sig[0], ref[0]) and sig[1] == ref[1] and sig[2] == ref[2]
        print("%-16s form %d: %.3f ms  identical to form %d: %s  max|dpose| %.3g" % (name, f, min(ts) * 1e3, forms[0], same, np.abs(sig[0] - ref[0]).max()))
ba.lib.vba_debug_set_chol_step(ba.h, 0)
