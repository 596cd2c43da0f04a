#!/bin/bash
# round 3, GPU run 1: 8-wave backward sweep A/B (bit-identity is printed by ab_bench) + stamps
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 300 python3 tools/ab_bench.py "PAULE_HIP_FUSED=1,PAULE_HIP_BWD_WAVES=4/PAULE_HIP_FUSED=1,PAULE_HIP_BWD_WAVES=8" 8 10 > gpurun_out/r3_ab_bwd_waves.txt 2>&1 || { tail -20 gpurun_out/r3_ab_bwd_waves.txt; exit 1; }
cat gpurun_out/r3_ab_bwd_waves.txt
AB_SET=B timeout -k 10 300 python3 tools/ab_bench.py PAULE_HIP_BWD_WAVES=4,8 6 10 > gpurun_out/r3_ab_bwd_waves_setB.txt 2>&1 || { tail -20 gpurun_out/r3_ab_bwd_waves_setB.txt; exit 1; }
cat gpurun_out/r3_ab_bwd_waves_setB.txt
(cd paule_amd/csrc && make stamps > /dev/null 2>&1) && PAULE_HIP_FUSED=0 timeout -k 10 200 python3 tools/sweep_stamps.py 256 > gpurun_out/r3_sweep_stamps.txt 2>&1; cat gpurun_out/r3_sweep_stamps.txt
