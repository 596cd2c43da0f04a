#!/bin/bash
# f32 batches of more 16-row groups than the chip holds: chains (lstm_chain_f32.hip) vs groups taking turns / launch-per-step kernels
for b in 96 160 256; do
  echo "## B=$b f32 T=300"
  AB_DTYPE=f32 AB_BATCH=$b timeout -k 10 400 python3 tools/ab_bench.py PAULE_HIP_F32_CHAINS=0,-1 2 5 2>&1 | grep -E 'median|rror'
done
echo "## B=256: forced chain counts"
AB_DTYPE=f32 AB_BATCH=256 timeout -k 10 400 python3 tools/ab_bench.py PAULE_HIP_F32_CHAINS=3,4,6,8 2 5 2>&1 | grep -E 'median|rror'
