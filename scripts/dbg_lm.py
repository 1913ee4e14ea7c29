import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from mc_slam_amd import synth, backend, abi
import oracle_lib
np.set_printoptions(precision=12, linewidth=200)
v = int(sys.argv[1]) if len(sys.argv) > 1 else 0
its = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (5, 10)
p = synth.make_window(v, algo=abi.ALGO_LM, n_kf=6, n_fixed=2 if v == 0 else 1, n_pt=60, n_obs=300, seed=33)
p.its_stage1, p.its_stage2 = its
ba = backend.LocalBA(0)
q, r = ba.solve(p)
qo, ro = oracle_lib.solve(p)
print("gpu its", r.its_done, "lam", r.lambda_final, "out", r.n_outliers); print(r.chi2_trace)
print("cpu its", ro.its_done, "lam", ro.lambda_final, "out", ro.n_outliers); print(ro.chi2_trace)
print("dpose", np.abs(q.kf_pose - qo.kf_pose).max(), "dpt", np.abs(q.pt - qo.pt).max())
