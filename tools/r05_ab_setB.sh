# set B (and set A as control) on two builds of the core: alt = before, core = after (tools/ab_bench.py, ms per iteration)
for spec in "B 256" "B 64" "A 224" "A 2048"; do
  set -- $spec
  for lib in libpaule_hip_alt.so libpaule_hip_core.so libpaule_hip_alt.so libpaule_hip_core.so; do
    [ -f paule_amd/csrc/$lib ] || continue
    echo -n "set $1 B=$2 $lib: "
    AB_SET=$1 AB_BATCH=$2 PAULE_HIP_LIB=$PWD/paule_amd/csrc/$lib timeout -k 10 250 python3 tools/ab_bench.py PAULE_HIP_GEMM_BIG=1 3 $([ $2 -gt 1000 ] && echo 4 || echo 20) 2>&1 | grep -E 'median|rror'
  done
done
