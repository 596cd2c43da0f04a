set -o pipefail
timeout -k 10 400 python3 tools/ab_bench.py "PAULE_HIP_BWD_PF=0/PAULE_HIP_BWD_PF=8/PAULE_HIP_BWD_PF=0,PAULE_HIP_BWD_PF_TOUCH=2/PAULE_HIP_BWD_PF=0,PAULE_HIP_BWD_PF_TOUCH=3/PAULE_HIP_BWD_PF=0,PAULE_HIP_BWD_PF_TOUCH=5/PAULE_HIP_BWD_PF=8,PAULE_HIP_BWD_PF_TOUCH=3" 4 10 2>&1 | grep -v amdgpu | tee gpurun_out/r05_ab_touch.txt || exit 1
PAULE_HIP_BWD_PF=0 PAULE_HIP_BWD_PF_TOUCH=3 timeout -k 10 200 python3 tools/sweep_stamps.py 256 2>&1 | grep -v amdgpu | tee gpurun_out/r05_touch_stamps.txt
