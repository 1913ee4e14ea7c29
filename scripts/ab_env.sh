#!/bin/bash
# same build, different environment settings, inside ONE gpurun call: scripts/ab_env.sh <sizes> "VAR=val" "VAR=val" ...
set -e
sizes=$1; shift
for e in "$@"; do
  echo "== $e"
  env $e timeout -k 10 200 python scripts/perf_probe.py 16 "$sizes"
done
