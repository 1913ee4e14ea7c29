#!/bin/bash
# What do the gather kernels really fetch?  L2 -> fabric read requests by size (32 / 64 / 128 B), L2 reads by sector, hits / misses:
# scripts/pmc_tcc.sh <tag> [kernels...]   (rocprofv3 PMC passes on their own, no tracing)
set -e
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd); out=$root/gpurun_out/tcc_$tag
mkdir -p $out; cd /tmp; export TMPDIR=/tmp VBA_STREAMS=1
args="$root/bench.py --workload c3 --steps 1 --warmup 0 --no-cpu-baseline --e2e-steps 0 --single-reps 0 --gen-procs 1 --distinct 32"
i=0
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_READ_sum TCC_READ_SECTORS_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  echo "pmc pass $i: $grp"
  rocprofv3 --pmc $grp --output-format csv -d $out/p$i -- python3 $args > /dev/null 2>&1 || echo "pass $i failed"
done
python3 $root/scripts/pmc_kernels.py $out "$@" > $root/gpurun_out/tcc_$tag.txt
rm -rf $out
