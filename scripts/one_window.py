import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mc_slam_amd import synth, backend
p = synth.config_c3(seed=100)
ba = backend.LocalBA(0)
ba.upload([p])
for _ in range(5):
    ba.run()
