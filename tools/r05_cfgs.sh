# PAULE_HIP_BWD_PF=0 (off) against the default (auto): tools/r05_cfgs.sh CONFIG ...
set -o pipefail
for c in "$@"; do
  for pf in 0 -1 0 -1; do
    echo -n "$c PAULE_HIP_BWD_PF=$pf: " | tee -a gpurun_out/r05_cfgs.txt
    PAULE_HIP_BWD_PF=$pf timeout -k 10 300 python3 bench.py --config $c --steps 8 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['roofline'].get('kernel'), round(d['roofline'].get('avg_launch_us') or 0,1))" | tee -a gpurun_out/r05_cfgs.txt || exit 1
  done
done
