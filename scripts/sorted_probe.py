"""Exploration: does it pay to hand the windows over with landmarks already in track order (by first keyframe)?"""
import sys, time, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mc_slam_amd import synth, backend

def sort_landmarks(p):
    q = p.copy()
    nb = p.pt_obs_begin
    cnt = np.diff(nb)
    first = p.pt_ref_kf.copy()
    for i in range(len(cnt)):
        if cnt[i]:
            first[i] = min(first[i], p.obs_kf[nb[i]:nb[i + 1]].min())
    order = np.argsort(first, kind="stable")
    q.pt = p.pt[order].copy(); q.pt_ref_kf = p.pt_ref_kf[order].copy()
    idx = np.concatenate([np.arange(nb[i], nb[i + 1]) for i in order])
    q.obs_kf = p.obs_kf[idx].copy(); q.obs_uv = p.obs_uv[idx].copy(); q.obs_w = p.obs_w[idx].copy()
    q.pt_obs_begin = np.concatenate([[0], np.cumsum(cnt[order])]).astype(np.int32)
    return q

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
wins = [synth.config_c3(seed=100 + i) for i in range(8)]
ba = backend.LocalBA(0)
for name, ws in (("as generated", wins), ("landmarks in track order", [sort_landmarks(w) for w in wins]), ("as generated", wins)):
    ba.upload([ws[i % 8] for i in range(B)])
    ba.set_profile(False); ba.run()
    ts = []
    for _ in range(3):
        t0 = time.time(); ba.run(); ts.append(time.time() - t0)
    q, r = ba.download()
    ba.set_profile(True); ba.run(); pf = ba.get_profile(); ba.set_profile(False)
    print("%-26s run %.2f ms  its %s chi2 %.6f  " % (name, min(ts) * 1e3, r[0].its_done, r[0].chi2_vis) +
          "  ".join("%s %.2f" % (k, v["ms"]) for k, v in pf.items() if k != "total_ms" and v["launches"]), flush=True)
