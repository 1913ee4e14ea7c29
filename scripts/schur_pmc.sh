#!/bin/bash
# PMC counters of the Schur kernel for two builds (ab/quad0.so, ab/quad1.so): bash scripts/schur_pmc.sh
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in quad0 quad1; do
  for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
             "TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TD_TD_BUSY_sum" \
             "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
    tag=$(echo $set | cut -d' ' -f1)
    VBA_LIB=ab/$v.so rocprofv3 --pmc $set --kernel-include-regex "k_schur_all\\(" -d gpurun_out/pmc_$v/$tag -o out --output-format csv -- python3 scripts/quick_ab.py 1024 1 > gpurun_out/pmc_$v.$tag.log 2>&1 || echo "FAILED $v $tag"
  done
done
python3 - <<'PY'
import csv, glob, collections
for v in ("quad0", "quad1"):
    tot = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob("gpurun_out/pmc_%s/*/**/*counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(f)):
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(v, {k: (round(tot[k] / max(1, n[k])), n[k]) for k in sorted(tot)})
PY
