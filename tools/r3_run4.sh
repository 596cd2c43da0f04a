#!/bin/bash
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
export PAULE_HIP_FUSED_W8=0
timeout -k 10 300 python3 tools/ab_bench.py PAULE_HIP_BWD_STREAM=0,1 8 10 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_ab_bwd_stream.txt
[ ${PIPESTATUS[0]} -eq 0 ] || exit 1
AB_SET=B timeout -k 10 300 python3 tools/ab_bench.py PAULE_HIP_BWD_STREAM=0,1 4 10 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_ab_bwd_stream_setB.txt
[ ${PIPESTATUS[0]} -eq 0 ] || exit 1
AB_BATCH=144 AB_FRAMES=61 timeout -k 10 300 python3 tools/ab_bench.py PAULE_HIP_BWD_STREAM=0,1 2 10 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_ab_bwd_stream_ragged.txt
[ ${PIPESTATUS[0]} -eq 0 ] || exit 1
(cd paule_amd/csrc && make stamps > /dev/null 2>&1) && PAULE_HIP_FUSED=0 timeout -k 10 200 python3 tools/sweep_stamps.py 256 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_sweep_stamps_stream.txt
