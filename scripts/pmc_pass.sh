#!/bin/bash
# rocprofv3 PMC passes (each on its own, no tracing) over one bench workload: scripts/pmc_pass.sh <tag> <workload> <kernel names...>
set -e
tag=$1; wl=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd); out=$root/gpurun_out/pmc_$tag
mkdir -p $out; cd /tmp; export TMPDIR=/tmp VBA_STREAMS=1
dist=""; [ "$wl" = c3 ] && dist="--distinct 32"
args="$root/bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --e2e-steps 0 --single-reps 0 --gen-procs 1 $dist --no-pcg"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  echo "pmc pass $i: $grp"
  rocprofv3 --pmc $grp --output-format csv -d $out/p$i -- python3 $args > /dev/null 2>&1 || echo "pass $i failed"
done
python3 $root/scripts/pmc_kernels.py $out "$@" > $root/gpurun_out/pmc_$tag.txt
rm -rf $out
