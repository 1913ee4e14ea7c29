#!/bin/bash
# VGPR / scratch / occupancy table of every kernel of the backend (cross-compiles, no GPU needed)
cd "$(dirname "$0")/../mc_slam_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c vislam_ba.hip -o /tmp/regs.o -Rpass-analysis=kernel-resource-usage 2>&1 |
  grep -E "Function Name|VGPRs:|ScratchSize|Occupancy" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | paste - - - - | awk '{print}' | sed 's/Function Name: //'
