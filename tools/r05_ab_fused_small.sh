# alt (before) vs core (after) on the shapes that run lstm_fused.hip's one-per-CU launches
for spec in "128 300" "64 300" "128 2000" "16 2000"; do
  set -- $spec
  for lib in libpaule_hip_alt.so libpaule_hip_core.so libpaule_hip_alt.so libpaule_hip_core.so; do
    echo -n "B=$1 T=$2 $lib: "
    AB_BATCH=$1 AB_FRAMES=$2 PAULE_HIP_LIB=$PWD/paule_amd/csrc/$lib timeout -k 10 250 python3 tools/ab_bench.py PAULE_HIP_GEMM_BIG=1 3 $([ $2 -gt 1000 ] && echo 5 || echo 20) 2>&1 | grep -E 'median|rror'
  done
done
