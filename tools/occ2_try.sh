#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python3 -m pytest tests/test_hip_parity.py -x -q -m gpu -k "two_per_cu" > gpurun_out/occ2_test.log 2>&1
echo "test rc=$?"; tail -n 15 gpurun_out/occ2_test.log | cut -c1-300
timeout -k 10 300 python3 tools/ab_bench.py PAULE_HIP_FUSED_OCC2=0,1 6 10 > gpurun_out/occ2_ab.log 2>&1
echo "ab rc=$?"; tail -n 5 gpurun_out/occ2_ab.log | cut -c1-300
