import sys, os, collections
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
from mc_slam_amd import synth, backend
B = int(sys.argv[1]); prof = int(sys.argv[2]) if len(sys.argv) > 2 else 0
wins = [synth.config_c3(seed=100 + i) for i in range(4)]
ba = backend.LocalBA(0)
ba.upload([wins[i % 4] for i in range(B)])
ba.set_profile(bool(prof))
ba.run(); q, r = ba.download()
print("B", B, "prof", prof, collections.Counter((x.its_done, x.status) for x in r).most_common(6))
bad = [i for i, x in enumerate(r) if x.its_done != r[i % 4].its_done]
print("first deviating windows", bad[:10], "count", len(bad))
