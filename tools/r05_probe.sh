#!/bin/bash
# Round 5 probe: how long is a chain-step when a backward workgroup serves C groups in turn, and what do the
# existing fused backward roles cost at 256 rows.  usage: tools/r05_probe.sh [parts]
set -o pipefail
mkdir -p gpurun_out
parts=${@:-chains stamps fused}
for part in $parts; do
  case $part in
    chains) timeout -k 10 400 python3 tools/ab_bench.py PAULE_HIP_BWD_CHAINS=0,1,2,3,4 4 10 2>&1 | tee gpurun_out/r05_ab_chains.txt || exit 1 ;;
    stamps) for c in 0 1 2 3 4; do
              echo "=== PAULE_HIP_BWD_CHAINS=$c" | tee -a gpurun_out/r05_chain_stamps.txt
              PAULE_HIP_BWD_CHAINS=$c timeout -k 10 200 python3 tools/sweep_stamps.py 256 2>&1 | tee -a gpurun_out/r05_chain_stamps.txt || exit 1
            done ;;
    pf)     timeout -k 10 400 python3 tools/ab_bench.py PAULE_HIP_BWD_PF=0,1,2,4 4 10 2>&1 | tee gpurun_out/r05_ab_pf.txt || exit 1
            timeout -k 10 400 python3 tools/ab_bench.py "PAULE_HIP_BWD_PF=1,PAULE_HIP_BWD_PF_DIST=2/PAULE_HIP_BWD_PF=1,PAULE_HIP_BWD_PF_DIST=5/PAULE_HIP_BWD_PF=1,PAULE_HIP_BWD_PF_DIST=8" 4 10 2>&1 | tee -a gpurun_out/r05_ab_pf.txt || exit 1
            for c in 0 1; do
              echo "=== PAULE_HIP_BWD_PF=$c" | tee -a gpurun_out/r05_pf_stamps.txt
              PAULE_HIP_BWD_PF=$c timeout -k 10 200 python3 tools/sweep_stamps.py 256 2>&1 | tee -a gpurun_out/r05_pf_stamps.txt || exit 1
            done ;;
    fused)  timeout -k 10 300 python3 tools/ab_bench.py PAULE_HIP_FUSED=1,3 4 10 2>&1 | tee gpurun_out/r05_ab_fused_bwd_256.txt || exit 1
            PAULE_HIP_FUSED=3 timeout -k 10 200 python3 tools/fused_stamps.py 256 300 2>&1 | tee gpurun_out/r05_fused_bwd_256_stamps.txt || exit 1 ;;
  esac
done
