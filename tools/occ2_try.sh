#!/bin/bash
set -o pipefail
for cfg in "64 300 B" "128 300 B" "192 300 B" "256 300 B" "64 300 A" "128 300 A"; do
  set -- $cfg
  echo "=== B=$1 T=$2 set $3"
  AB_BATCH=$1 AB_FRAMES=$2 AB_SET=$3 timeout -k 10 300 python3 tools/ab_bench.py PAULE_HIP_FUSED_OCC2=0,1 4 5 2>&1 | tail -n 3
  AB_SET=$3 PAULE_HIP_FUSED_OCC2=0 timeout -k 10 100 python3 tools/fwd_launch_probe.py "X=1" $1 $2 2>&1 | tail -n 1
done
