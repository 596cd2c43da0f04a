#!/bin/bash
# One GPU-box session.  usage: tools/gpu_check.sh [step ...]
# steps: pytest  pytest_k (PYTEST_K=expr)  smoke  bench  bench_cfg2  bench_f32  bench_train  bench_nograph  bench_all  stress  rocprof  pmc
# (default: pytest smoke bench bench_cfg2 rocprof)
# Stops after a timed-out / killed step (never start another GPU step after that).
set -o pipefail
mkdir -p gpurun_out
rm -rf gpurun_out/prof gpurun_out/pmc_fetch gpurun_out/pmc_write   # stale profiles of earlier calls
export TMPDIR=/tmp
STEPS=${@:-pytest smoke bench bench_cfg2 rocprof}
step() {  # name, timeout, command...
    local name=$1 tmo=$2; shift 2
    echo "=== $name" | tee -a gpurun_out/session.log
    timeout -k 10 "$tmo" "$@" > "gpurun_out/$name.log" 2>&1
    local rc=$?
    echo "=== $name rc=$rc" | tee -a gpurun_out/session.log
    tail -n 12 "gpurun_out/$name.log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name timed out / was killed: stopping"; exit $rc; fi
    return $rc
}
: > gpurun_out/session.log
for s in $STEPS; do
  case $s in
    pytest) step pytest_gpu 900 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider --durations=12 ;;
    pytest_k) step pytest_k 600 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider -k "${PYTEST_K:-f32}" ;;
    smoke) step smoke 300 python -c "import __graft_entry__ as g; g.smoke()" ;;
    bench) step bench 400 python bench.py --steps 10 --warmup 2 ;;
    bench_cfg2) step bench_cfg2 300 python bench.py --config cfg2 --steps 10 --warmup 2 --no-cpu-baseline ;;
    bench_f32) step bench_f32 300 python bench.py --config cfg3_f32 --steps 5 --warmup 1 --no-cpu-baseline ;;
    bench_train) step bench_train 300 python bench.py --config train8 --steps 30 --warmup 3
                 step bench_train_f32 300 python bench.py --config train8_f32 --steps 30 --warmup 3 --no-cpu-baseline ;;
    bench_nograph) step bench_nograph 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-graph ;;
    bench_all) for c in cfg1 cfg3_setB cfg3_setC cfg3_soma cfg5 cfg4_1gpu cfg3_f32; do
                 step bench_$c 300 python bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline; done ;;
    stress) step capture_stress 600 python tools/microbench/capture_stress.py 20 ;;
    rocprof) rm -rf gpurun_out/prof
             step rocprof 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline
             find gpurun_out/prof -name "*stats*" | head ;;
    pmc) rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
         step pmc_fetch 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
         step pmc_write 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
         python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write cfg3 gpurun_out/traffic.json | tee gpurun_out/pmc_summary.log ;;
    *) echo "unknown step $s" ;;
  esac
done
