#!/bin/bash
# PMC counters of one kernel on 1024 resident C3 windows: bash scripts/kernel_pmc.sh <kernel regex> <tag> [library]
# (separate passes per counter set, --pmc alone: no trace options; the program itself behind "--")
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
re="$1"; tag="$2"; lib="${3:-mc_slam_amd/csrc/libvislam_ba.so}"
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVES" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  VBA_LIB=$lib timeout -k 10 240 rocprofv3 --pmc $set --kernel-include-regex "$re" -d gpurun_out/pmc_$tag/set$i -o out --output-format csv -- python3 scripts/quick_ab.py 1024 1 > gpurun_out/pmc_$tag.set$i.log 2>&1 || { echo "FAILED $tag set$i"; exit 1; }
  echo "set$i done" >> gpurun_out/pmc_$tag.progress
done
python3 - "$tag" <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob("gpurun_out/pmc_%s/*/**/*counter_collection.csv" % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
with open("gpurun_out/pmc_%s.txt" % tag, "w") as o:
    for k in sorted(tot): o.write("%-34s %16.0f  (%d dispatches)\n" % (k, tot[k] / n[k], n[k]))
print(open("gpurun_out/pmc_%s.txt" % tag).read())
PY
