#!/bin/bash
# A/B of several builds inside ONE gpurun call (boxes differ): scripts/ab.sh <batch sizes> <lib>...
set -e
sizes=$1; shift
for lib in "$@"; do
  echo "== $lib"
  VBA_LIB=$lib timeout -k 10 200 python scripts/perf_probe.py 16 "$sizes"
done
