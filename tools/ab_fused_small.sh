#!/bin/bash
# small batches: chunk pipelines (PAULE_HIP_FUSED=0) vs pipelined forward + fused backward launch (=2) vs both fused (=3)
V="PAULE_HIP_FUSED=0/PAULE_HIP_FUSED=2,PAULE_HIP_FUSED_MIN_B=1/PAULE_HIP_FUSED=3,PAULE_HIP_FUSED_MIN_B=1"
for b in 1 8 16 32 48; do echo "## B=$b T=300"; AB_BATCH=$b timeout -k 10 200 python3 tools/ab_bench.py "$V" 3 10 2>&1 | grep -E 'median|rror' ; done
for b in 16 32; do echo "## B=$b T=2000"; AB_BATCH=$b AB_FRAMES=2000 timeout -k 10 300 python3 tools/ab_bench.py "$V" 2 4 2>&1 | grep -E 'median|rror' ; done
