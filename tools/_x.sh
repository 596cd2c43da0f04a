set -e
mkdir -p gpurun_out
timeout -k 10 500 python3 bench.py --steps 20 --warmup 3 > gpurun_out/b_cfg3.log 2>&1; grep '^{' gpurun_out/b_cfg3.log > gpurun_out/r03_bench_cfg3.json
for c in cfg3_setB cfg5_128 cfg2 cfg5; do
  timeout -k 10 400 python3 bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/b_$c.log 2>&1; grep '^{' gpurun_out/b_$c.log > gpurun_out/r03_bench_$c.json
done
python3 - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_bench_cfg*.json')):
    d=json.loads(open(f).read()); print(f, d['ms_per_step'], d['roofline']['kernel'], d['roofline']['traffic'])
P
