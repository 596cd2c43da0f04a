# alt (before) vs core (after): bench lines, ms per iteration
for c in cfg3 cfg3_soma cfg5_128 cfg2 cfg1_bf16 cfg5; do
  for lib in libpaule_hip_alt.so libpaule_hip_core.so libpaule_hip_alt.so libpaule_hip_core.so; do
    echo -n "$c $lib "
    PAULE_HIP_LIB=$PWD/paule_amd/csrc/$lib timeout -k 10 300 python3 bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
  done
done
