#!/bin/bash
# kernel trace of a few single-window solves: per-kernel stats (run on the GPU box)
root=$(cd "$(dirname "$0")/.." && pwd); out=$root/gpurun_out/one_trace
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/scripts/one_window.py > /dev/null 2>&1
cat $out/*/*_kernel_stats.csv | cut -d, -f1-4,6,7 | head -30
