# set B (and set A as control) on variants of lstm_fused2.hip: r4f = round 4's source, core = shipped, alt / alt2 = experiments
for spec in "B 256" "B 64" "A 256"; do
  set -- $spec
  for lib in libpaule_hip_r4f.so libpaule_hip_core.so libpaule_hip_alt.so libpaule_hip_alt2.so libpaule_hip_r4f.so libpaule_hip_core.so libpaule_hip_alt.so libpaule_hip_alt2.so; do
    [ -f paule_amd/csrc/$lib ] || continue
    echo -n "set $1 B=$2 $lib: "
    AB_SET=$1 AB_BATCH=$2 PAULE_HIP_LIB=$PWD/paule_amd/csrc/$lib timeout -k 10 200 python3 tools/ab_bench.py PAULE_HIP_GEMM_BIG=1 3 20 2>&1 | grep -E 'median|rror'
  done
done
