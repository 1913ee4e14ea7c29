import sys, os, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from mc_slam_amd import synth, backend, abi
import oracle_lib
np.set_printoptions(precision=6, linewidth=220, suppress=False)
v = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ALGO = int(os.environ.get('ALGO', '0'))
p = synth.make_window(v, algo=ALGO, n_kf=6, n_fixed=2 if v == 0 else 1, n_pt=60, n_obs=300, seed=33)
p.its_stage1, p.its_stage2 = int(sys.argv[2]) if len(sys.argv) > 2 else 1, int(sys.argv[3]) if len(sys.argv) > 3 else 0
ba = backend.LocalBA(0)
q, r = ba.solve(p)
lib = ba.lib
lib.vba_debug_buf_id.argtypes = [C.c_char_p]; lib.vba_debug_copy.argtypes = [C.c_void_p, C.c_int32, C.c_uint64, C.c_void_p, C.c_uint64]
def rd(name, n, dt=np.float64):
    a = np.zeros(n, dtype=dt)
    rc = lib.vba_debug_copy(ba.h, lib.vba_debug_buf_id(name.encode()), 0, a.ctypes.data_as(C.c_void_p), a.nbytes)
    assert rc == 0, name
    return a
pdim = 6 if v == 0 else 15
npp = pdim * p.n_kf_free; nS = (npp + 31) // 32 * 32
S = rd("S", nS * nS).reshape(nS, nS)
x = rd("VEC", nS); bp = rd("BPOSE", 2 * nS)
lam0 = 0.0
if ALGO == 1:
    n, H, b, xo, chi = oracle_lib.linearize(p, 0.0)
    lam0 = 1e-5 * np.abs(np.diag(H)).max()
    print('lambda0 expected', lam0, 'gpu lambda_final', r.lambda_final, 'ratio', r.lambda_final / lam0)
n, H, b, xo, chi = oracle_lib.linearize(p, lam0)
Hpp = H[:npp, :npp]; Hpl = H[:npp, npp:]; Hll = H[npp:, npp:]
Sref = Hpp + lam0 * np.eye(npp) - Hpl @ np.linalg.solve(Hll + lam0 * np.eye(Hll.shape[0]), Hpl.T)
if v == 0:
    perm = np.arange(npp)
else:
    nf = p.n_kf_free
    perm = np.array([ (9 * nf + 6 * a + rr) if rr < 6 else (9 * a + rr - 6) for a in range(nf) for rr in range(15)])
Sg = np.tril(S[:npp, :npp]); Sg = Sg + np.tril(Sg, -1).T
Sg = Sg[np.ix_(perm, perm)]
print("S max abs", np.abs(Sref).max(), "diff", np.abs(Sg - Sref).max())
d = np.abs(Sg - Sref); i, j = np.unravel_index(d.argmax(), d.shape); print("worst", i, j, Sg[i, j], Sref[i, j])
print("bpose diff", np.abs(bp[:nS][perm] - b[:npp]).max(), "hdiag diff", np.abs(bp[nS:][perm] - np.diag(Hpp)).max())
print("x diff", np.abs(x[perm] - xo[:npp]).max(), np.abs(xo[:npp]).max())
print("blocks:\n", (np.abs(Sg - Sref).reshape(p.n_kf_free, pdim, p.n_kf_free, pdim).max(axis=(1, 3))))
qo, ro = oracle_lib.solve(p)
print("after 1 it: dpose", np.abs(q.kf_pose - qo.kf_pose).max(axis=0), "dpt", np.abs(q.pt - qo.pt).max(), "dvel", np.abs(q.kf_vel - qo.kf_vel).max())
dl_g = q.pt - p.pt; dl_o = qo.pt - p.pt
print("pt step gpu", dl_g[:3], "\npt step cpu", dl_o[:3])
print("trace", r.chi2_trace, ro.chi2_trace)
