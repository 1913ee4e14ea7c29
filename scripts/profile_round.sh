#!/bin/bash
# The three rocprofv3 passes behind profiles/<tag>_*: kernel trace + stats, FETCH_SIZE, WRITE_SIZE (PMC passes on their own, as the
# MI355X guide prescribes), then the per-kernel traffic summary bench.py reads.  Run on the GPU box: scripts/profile_round.sh <tag>
set -e
tag=$1; root=$(cd "$(dirname "$0")/.." && pwd); out=$root/gpurun_out/prof_$tag
cd /tmp; export TMPDIR=/tmp
# one window group / one stream: every launch then covers the whole batch, as in the profiled run bench.py takes its roofline from
export VBA_STREAMS=1
args="$root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --e2e-steps 0 --single-reps 0 --gen-procs 1 --distinct 32"
echo "pass 1: kernel trace"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $args > $out.bench_trace.json 2>/dev/null
echo "pass 2: FETCH_SIZE"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $args > /dev/null 2>&1
echo "pass 3: WRITE_SIZE"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $args > /dev/null 2>&1
python3 $root/scripts/pmc_summary.py $out/trace/*/*_kernel_trace.csv $out/fetch/*/*_counter_collection.csv $out/write/*/*_counter_collection.csv $out.traffic.json 4096
cp $out/trace/*/*_kernel_stats.csv $out.kernel_stats.csv
rm -rf $out/trace/*/*_kernel_trace.csv $out/fetch $out/write
