#!/bin/bash
# round 3, GPU run 2: the two-waves-per-SIMD forward launch: bit-identity test, then A/B
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 500 python3 -m pytest tests/test_hip_parity.py -x -q -p no:cacheprovider -k "test_fused_forward_is_bit_identical" > gpurun_out/r3_w8_test.log 2>&1
rc=$?; tail -25 gpurun_out/r3_w8_test.log
if [ $rc -ne 0 ]; then echo "bit-identity test rc=$rc: stopping"; exit $rc; fi
PAULE_HIP_DEBUG_FUSED=1 timeout -k 10 300 python3 tools/ab_bench.py PAULE_HIP_FUSED_W8=0,1 6 10 > gpurun_out/r3_ab_w8.txt 2>&1 || { tail -20 gpurun_out/r3_ab_w8.txt; exit 1; }
grep -v "amdgpu.ids" gpurun_out/r3_ab_w8.txt | sort | uniq -c | sort -rn | head -20
for b in 64 128; do AB_BATCH=$b timeout -k 10 300 python3 tools/ab_bench.py PAULE_HIP_FUSED_W8=0,1 4 10 > gpurun_out/r3_ab_w8_b$b.txt 2>&1 || { tail -20 gpurun_out/r3_ab_w8_b$b.txt; exit 1; }; grep -v amdgpu.ids gpurun_out/r3_ab_w8_b$b.txt; done
