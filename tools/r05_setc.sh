set -o pipefail
for rep in 1 2 3; do for pf in 0 -1; do
  echo -n "cfg3_setC PF=$pf: "
  PAULE_HIP_BWD_PF=$pf timeout -k 10 300 python3 bench.py --config cfg3_setC --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" || exit 1
done; done
