#!/bin/bash
# A/B of the fused launches outside their default range (long sequences, fewer than 129 rows).  usage: tools/fused_small_ab.sh
mkdir -p gpurun_out
for cfg in cfg5 cfg5_128; do
  for v in "PAULE_HIP_FUSED=0" "PAULE_HIP_FUSED=1 PAULE_HIP_FUSED_MIN_B=1" "PAULE_HIP_FUSED=3 PAULE_HIP_FUSED_MIN_B=1"; do
    echo "=== $cfg $v"
    env $v timeout -k 10 300 python3 bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep -E 'timed region|Error|error|status' | cut -c1-200
  done
done
