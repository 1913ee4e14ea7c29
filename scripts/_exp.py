import sys
sys.path.insert(0, "/root/repo")
import bench
from mc_slam_amd import backend
wins = bench.make_windows([("c3", 100 + i, False) for i in range(16)], 1)
batch = [wins[i % 16] for i in range(4096)]
ba = backend.LocalBA(0)
ba.upload(batch); ba.run()
ba.set_profile(True)
ba.upload(batch); ba.run()
pf = ba.get_profile()
print({k: (round(v["ms"], 1), v["launches"]) for k, v in pf.items() if isinstance(v, dict)})
