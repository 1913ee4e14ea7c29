"""end-to-end (vba_batch_solve) throughput for several chunk sizes / lane counts: python scripts/e2e_sweep.py [n_windows] [distinct]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mc_slam_amd import backend
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nd = int(sys.argv[2]) if len(sys.argv) > 2 else 64
wins = bench.make_windows([("c3", 100 + i, False, "caller") for i in range(nd)], 16)
batch = [wins[i % nd] for i in range(n)]
ba = backend.LocalBA(0, hooks=True)
ba.upload(batch); ba.run()
t = time.perf_counter(); ba.run(); ba.run(); tr = (time.perf_counter() - t) / 2
print("resident %.0f windows/s" % (n / tr), flush=True)
packed = ba.pack(batch)
cfgs = [tuple(map(int, a.split("x"))) for a in sys.argv[3:]] or [(512, 3), (512, 4), (512, 6), (256, 4), (256, 6), (1024, 3), (1024, 4), (384, 5)]
for chunk, lanes in cfgs:
    ba.lib.vba_debug_set_chunking(ba.h, chunk, lanes)
    ba.solve_packed(packed)
    ts = []
    for _ in range(3):
        ba.pack_reset(packed)
        t = time.perf_counter(); ba.solve_packed(packed); ts.append(time.perf_counter() - t)
    print("chunk %4d lanes %d: %.0f windows/s (best %.0f)" % (chunk, lanes, n * len(ts) / sum(ts), n / min(ts)), flush=True)
