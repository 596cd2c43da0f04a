# gemm.hip's tiles (PAULE_HIP_GEMM_BIG=0) against gemm_big.hip's 256 x 256 tiles, same box, bench lines of the configurations with large bf16 products
timeout -k 10 300 python3 tools/ab_bench.py PAULE_HIP_GEMM_BIG=0,1 4 20 2>&1 | grep -E "median|identical"
for c in cfg3_soma cfg4_1gpu cfg3_setC; do
  for big in 0 1 0 1; do
    echo "== $c PAULE_HIP_GEMM_BIG=$big"
    PAULE_HIP_GEMM_BIG=$big timeout -k 10 300 python3 bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
  done
done
