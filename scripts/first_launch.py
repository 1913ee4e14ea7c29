"""Exploration: one run of a 2048-window C3 batch (for a rocprofv3 kernel trace: compare the FIRST launch of a kernel between builds)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mc_slam_amd import synth, backend
wins = [synth.config_c3(seed=100 + i) for i in range(8)]
ba = backend.LocalBA(0)
ba.upload([wins[i % 8] for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2048)])
ba.run()
