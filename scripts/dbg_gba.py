import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from mc_slam_amd import abi, synth, backend
import oracle_lib
from test_gpu_parity import _gba
np.set_printoptions(precision=15, linewidth=200)
p = _gba(abi.VARIANT_PRV_XYZ, 0, seed=53)
ba = backend.LocalBA(0)
q, r = ba.solve(p)
qo, ro = oracle_lib.solve(p)
print(r.its_done, ro.its_done, r.lambda_final, ro.lambda_final)
n = min(len(r.chi2_trace), len(ro.chi2_trace))
for i in range(max(len(r.chi2_trace), len(ro.chi2_trace))):
    a = r.chi2_trace[i] if i < len(r.chi2_trace) else None
    b = ro.chi2_trace[i] if i < len(ro.chi2_trace) else None
    print(i, a, b, (a - b) / b if a is not None and b is not None else "")
