#!/bin/bash
# kernel stats of a resident batch run (run on the GPU box): scripts/batch_trace.sh <n_windows>
root=$(cd "$(dirname "$0")/.." && pwd); out=$root/gpurun_out/batch_trace
cd /tmp; export TMPDIR=/tmp; export VBA_STREAMS=1
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/scripts/quick_ab.py ${1:-4096} 1 > /dev/null 2>&1
cat $out/*/*_kernel_stats.csv | cut -d, -f1-4,6,7 | head -14
