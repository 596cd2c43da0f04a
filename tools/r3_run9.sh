#!/bin/bash
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 300 python3 tools/ab_bench.py PAULE_HIP_BWD_WAVES=8,16 8 10 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_ab_bwd_waves16.txt
[ ${PIPESTATUS[0]} -eq 0 ] || exit 1
AB_SET=B timeout -k 10 300 python3 tools/ab_bench.py PAULE_HIP_BWD_WAVES=8,16 4 10 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r3_ab_bwd_waves16.txt
[ ${PIPESTATUS[0]} -eq 0 ] || exit 1
AB_BATCH=144 AB_FRAMES=61 timeout -k 10 300 python3 tools/ab_bench.py PAULE_HIP_BWD_WAVES=8,16 4 10 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r3_ab_bwd_waves16.txt
AB_BATCH=270 AB_FRAMES=17 timeout -k 10 300 python3 tools/ab_bench.py PAULE_HIP_BWD_WAVES=8,16 4 10 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r3_ab_bwd_waves16.txt
