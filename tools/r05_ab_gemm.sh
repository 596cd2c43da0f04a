# old vs new narrow-GEMM tile rule: cfg2 (f32, B = 64, acoustic) and B = 64 bf16
for rep in 1 2; do
for lib in libpaule_hip_alt.so libpaule_hip_core.so; do
  echo "== $lib cfg2"
  AB_OBJECTIVE=acoustic AB_DTYPE=f32 AB_BATCH=64 PAULE_HIP_LIB=$PWD/paule_amd/csrc/$lib timeout -k 10 200 python3 tools/ab_bench.py PAULE_HIP_FUSED=1 3 20 2>&1 | grep -E 'median|rror'
  echo "== $lib B=64 bf16"
  AB_BATCH=64 PAULE_HIP_LIB=$PWD/paule_amd/csrc/$lib timeout -k 10 200 python3 tools/ab_bench.py PAULE_HIP_FUSED=3 3 20 2>&1 | grep -E 'median|rror'
done; done
