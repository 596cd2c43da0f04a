#!/bin/bash
# timing of two builds of the CORE library back to back on one box (PAULE_HIP_LIB selects the build; libpaule_hip_alt.so = a build with other -D options)
# usage: tools/ab_two_libs.sh [B] [variant]
B=${1:-256}; V=${2:-PAULE_HIP_FUSED=1}
for rep in 1 2 3; do
for lib in libpaule_hip_core.so libpaule_hip_alt.so; do
  echo "== $lib B=$B"
  AB_BATCH=$B PAULE_HIP_LIB=$PWD/paule_amd/csrc/$lib timeout -k 10 200 python3 tools/ab_bench.py "$V" 3 20 2>&1 | grep -E 'median|rror'
done; done
