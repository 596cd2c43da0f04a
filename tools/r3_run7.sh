#!/bin/bash
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_hip_parity.py -q -p no:cacheprovider -k "fused_backward or fused_launches or cfg5_128 or cfg3_bf16_against_oracle_rows or bf16_sweep_ragged or across_kernel_families or rounding_emulation or census" > gpurun_out/r3_tests_c.log 2>&1
rc=$?; tail -12 gpurun_out/r3_tests_c.log
[ $rc -eq 0 ] || exit $rc
for b in 64 128 256; do AB_BATCH=$b timeout -k 10 300 python3 tools/ab_bench.py PAULE_HIP_FUSED=1,3 4 10 2>&1 | grep -v amdgpu.ids | sed "s/^/B=$b  /" | tee -a gpurun_out/r3_ab_fused_bwd8.txt; done
AB_BATCH=128 AB_FRAMES=2000 timeout -k 10 300 python3 tools/ab_bench.py PAULE_HIP_FUSED=1,3 2 4 2>&1 | grep -v amdgpu.ids | sed "s/^/B=128 T=2000  /" | tee -a gpurun_out/r3_ab_fused_bwd8.txt
