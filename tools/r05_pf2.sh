set -o pipefail
timeout -k 10 400 python3 tools/ab_bench.py PAULE_HIP_BWD_PF=0,4,6,8,9 4 10 2>&1 | grep -v amdgpu | tee gpurun_out/r05_ab_pf2.txt || exit 1
timeout -k 10 400 python3 tools/ab_bench.py "PAULE_HIP_BWD_PF=8,PAULE_HIP_BWD_PF_DIST=1/PAULE_HIP_BWD_PF=8,PAULE_HIP_BWD_PF_DIST=2/PAULE_HIP_BWD_PF=8,PAULE_HIP_BWD_PF_DIST=4/PAULE_HIP_BWD_PF=8,PAULE_HIP_BWD_PF_DIST=6" 4 10 2>&1 | grep -v amdgpu | tee -a gpurun_out/r05_ab_pf2.txt || exit 1
PAULE_HIP_BWD_PF=8 timeout -k 10 200 python3 tools/sweep_stamps.py 256 2>&1 | grep -v amdgpu | tee gpurun_out/r05_pf8_stamps.txt
