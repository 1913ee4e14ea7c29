#!/bin/bash
# diagnostic build with in-kernel shader-clock stamps (k_chol_step4) -> ab/stamps.so ; run: VBA_LIB=ab/stamps.so python scripts/stamps.py
cd "$(dirname "$0")/../mc_slam_amd/csrc" && mkdir -p ../../ab && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DVBA_STAMPS -DVBA_TEST_HOOKS -o ../../ab/stamps.so vislam_ba.hip
