#!/bin/bash
# A/B of the number of time chunks of the small-batch pipelines (PAULE_HIP_WAVEFRONT) at several shapes; one process per shape.
set -e
mkdir -p gpurun_out
out=gpurun_out/r03_ab_wavefront_chunks.txt
: > $out
for spec in "16 2000 4,8,16,24,32" "1 2000 4,8,16,32" "1 300 4,6,8,12,16" "16 300 4,6,8,12,16" "48 300 4,8,12" "32 2000 4,16,32" "16 1000 4,8,16,32"; do
  set -- $spec
  echo "B=$1 T=$2" >> $out
  AB_BATCH=$1 AB_FRAMES=$2 timeout -k 10 300 python3 tools/ab_bench.py PAULE_HIP_WAVEFRONT=$3 3 10 >> $out 2>&1
done
cat $out
