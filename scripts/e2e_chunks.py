"""end-to-end (vba_batch_solve) throughput with the chunking of the environment (VBA_CHUNKS / VBA_LANES / VBA_RUN_SLOTS ...):
python scripts/e2e_chunks.py [n_windows] [distinct] [warm-up calls]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mc_slam_amd import backend
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nd = int(sys.argv[2]) if len(sys.argv) > 2 else 64
wins = bench.make_windows([("c3", 100 + i, False, "caller") for i in range(nd)], 16)
batch = [wins[i % nd] for i in range(n)]
ba = backend.LocalBA(0)
ba.upload(batch); ba.run()
t = time.perf_counter(); ba.run(); ba.run(); tr = (time.perf_counter() - t) / 2
packed = ba.pack(batch)
for _ in range(int(sys.argv[3]) if len(sys.argv) > 3 else 1):
    ba.pack_reset(packed); ba.solve_packed(packed)
ts = []
for _ in range(4):
    ba.pack_reset(packed)
    t = time.perf_counter(); ba.solve_packed(packed); ts.append(time.perf_counter() - t)
e = n * len(ts) / sum(ts)
print("resident %.0f windows/s  end to end %.0f (best %.0f)  ratio %.3f   env %s" % (n / tr, e, n / min(ts), e / (n / tr), {k: v for k, v in os.environ.items() if k.startswith(("VBA_", "GPU_"))}), flush=True)
