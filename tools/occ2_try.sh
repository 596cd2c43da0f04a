#!/bin/bash
set -o pipefail
D=$PWD/paule_amd/csrc
for r in 1 2 3 4; do
for v in c p n; do
  echo "=== core_$v"
  PAULE_HIP_LIB=$D/libpaule_hip_core_$v.so timeout -k 10 200 python3 tools/occ2_probe.py "PAULE_HIP_FUSED_OCC2=1" 2>&1 | tail -n 1
done
done
