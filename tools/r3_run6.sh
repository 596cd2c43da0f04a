#!/bin/bash
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 800 python3 -m pytest tests/test_hip_parity.py -q -p no:cacheprovider --durations=15 -k "long_sequences or full_size or rounding_emulation or census or fused_forward_is_bit or fused_backward" > gpurun_out/r3_tests_b.log 2>&1
rc=$?; tail -30 gpurun_out/r3_tests_b.log
[ $rc -eq 0 ] || exit $rc
AB_BATCH=144 AB_FRAMES=61 timeout -k 10 300 python3 tools/ab_bench.py PAULE_HIP_BWD_STREAM=0,1 8 10 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_ab_bwd_stream_ragged.txt
