#!/bin/bash
# Measurement session of a round on the GPU box: bench lines, kernel stats, PMC passes (separate FETCH_SIZE / WRITE_SIZE runs, the
# profiler in front of the program itself).  usage: tools/gpu_measure.sh TAG [part ...]   parts: bench cpufull stats pmc pmc_xcd pmc_others others trainstats
set -o pipefail
tag=$1; shift
parts=${@:-bench stats pmc}
mkdir -p gpurun_out
export TMPDIR=/tmp
# gpurun_out/ does not travel to the box: entries of an earlier call of this session live in profiles/traffic.json (copied there by hand);
# start from them when they were taken on these kernel sources, so that pmc and pmc_others may run in separate calls
python3 - "$tag" <<'PYEOF'
import json, os, sys
sys.path.insert(0, os.getcwd())
import bench
src, dst = "profiles/traffic.json", f"gpurun_out/{sys.argv[1]}_traffic.json"
if os.path.exists(src) and not os.path.exists(dst):
    t = json.load(open(src))
    if t.get("source_digest") == bench.source_digest():
        json.dump(t, open(dst, "w"), indent=1, sort_keys=True)
PYEOF
run() {  # name, timeout, command...
  local name=$1 tmo=$2; shift 2
  echo "=== $name"
  timeout -k 10 "$tmo" "$@" > "gpurun_out/${tag}_$name.log" 2>&1
  local rc=$?
  echo "=== $name rc=$rc"; tail -n 3 "gpurun_out/${tag}_$name.log" | cut -c1-300
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name timed out: stopping"; exit 1; fi
}
pmc_pair() {  # config, suffix, extra env assignments...
  local cfg=$1 suf=$2; shift 2
  rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
  run pmc_fetch_${cfg}${suf} 500 env "$@" rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -- python3 bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline
  run pmc_write_${cfg}${suf} 500 env "$@" rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -- python3 bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline
  python3 tools/pmc_summary.py gpurun_out/pmc_f gpurun_out/pmc_w ${cfg}${suf} gpurun_out/${tag}_traffic.json | tee gpurun_out/${tag}_pmc_summary_${cfg}${suf}.txt
}
for part in $parts; do
  case $part in
    bench) run bench_cfg3 500 python3 bench.py --steps 20 --warmup 3
           grep '^{' gpurun_out/${tag}_bench_cfg3.log > gpurun_out/${tag}_bench_cfg3.json ;;
    cpufull) run bench_cfg3_cpufull 900 python3 bench.py --steps 20 --warmup 3 --cpu-full
             grep '^{' gpurun_out/${tag}_bench_cfg3_cpufull.log > gpurun_out/${tag}_bench_cfg3_cpufull.json ;;
    stats) rm -rf gpurun_out/prof
           run rocprof_cfg3 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline
           cp $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_cfg3_kernel_stats.csv ;;
    pmc) pmc_pair cfg3 "" PAULE_HIP_XCD_FAST=2 ;;
    pmc_xcd) pmc_pair cfg3 _xcdfast0 PAULE_HIP_XCD_FAST=0
             pmc_pair cfg3 _unfused PAULE_HIP_FUSED=0 ;;
    pmc_others) pmc_pair cfg3_setB "" PAULE_HIP_XCD_FAST=2
                pmc_pair cfg5_128 "" PAULE_HIP_XCD_FAST=2
                pmc_pair cfg2 "" PAULE_HIP_XCD_FAST=2
                pmc_pair cfg5 "" PAULE_HIP_XCD_FAST=2
                pmc_pair cfg3_soma "" PAULE_HIP_XCD_FAST=2
                pmc_pair cfg4_1gpu "" PAULE_HIP_XCD_FAST=2
                pmc_pair cfg1_bf16 "" PAULE_HIP_XCD_FAST=2 ;;
    others) for c in cfg1 cfg1_bf16 cfg2 cfg2_setB cfg3_setB cfg3_setC cfg3_f32 cfg5 cfg5_setB cfg5_128 cfg4_1gpu cfg3_soma train8; do   # train8 etc. print one JSON line each
              run bench_$c 400 python3 bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline
              grep '^{' gpurun_out/${tag}_bench_$c.log > gpurun_out/${tag}_bench_$c.json
            done ;;
    trainstats) rm -rf gpurun_out/prof_t
           run rocprof_train8 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_t -- python3 tools/train_bench.py 8 8 bf16 20
           cp $(find gpurun_out/prof_t -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_train8_kernel_stats.csv ;;
    *) echo "unknown part $part" ;;
  esac
done
