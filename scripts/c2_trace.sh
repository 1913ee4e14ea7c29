#!/bin/bash
# kernel stats of the C2 (vision-only, XYZ landmarks, LM) resident batch: scripts/c2_trace.sh
root=$(cd "$(dirname "$0")/.." && pwd); out=$root/gpurun_out/c2_trace
cd /tmp; export TMPDIR=/tmp; export VBA_STREAMS=1
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/bench.py --workload c2 --steps 2 --warmup 1 --no-cpu-baseline --e2e-steps 0 --single-reps 0 --gen-procs 1 > $root/gpurun_out/c2_trace.json 2>/dev/null
cp $out/*/*_kernel_stats.csv $root/gpurun_out/c2_kernel_stats.csv
