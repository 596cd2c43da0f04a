#!/bin/bash
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
# (1) the shipped fused backward launch, repeatedly, against the per-layer path (looking for the intermittent corruption of the 8-wave experiment)
for i in 1 2 3 4; do timeout -k 10 200 python3 tools/fused_bwd_diag.py 128x40 100x40 256x40 2>&1 | grep -v "amdgpu.ids\|plan\|      t=" ; done | tee gpurun_out/r3_fused_bwd_diag_shipped.txt | grep -c "bad batch rows \[\] (0)"
grep -v "bad batch rows \[\] (0)" gpurun_out/r3_fused_bwd_diag_shipped.txt | head
# (2) crossover table
: > gpurun_out/r3_ab_fused_range.txt
for b in 48 64 96 128 144 160 192 224 256; do AB_BATCH=$b PAULE_HIP_FUSED_MIN_B=1 timeout -k 10 300 python3 tools/ab_bench.py PAULE_HIP_FUSED=0,1,3 3 10 2>&1 | grep -v "amdgpu.ids\|final CP" | sed "s/^/B=$b  /" | tee -a gpurun_out/r3_ab_fused_range.txt; done
