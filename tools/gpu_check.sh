#!/bin/bash
# One GPU-box session: parity tests -> smoke -> bench -> rocprofv3 kernel stats.  Stops after a timed-out/killed step.
# usage: tools/gpu_check.sh [pytest-args...]
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
step() {  # name, timeout, command...
    local name=$1 tmo=$2; shift 2
    echo "=== $name" | tee -a gpurun_out/session.log
    timeout -k 10 "$tmo" "$@" > "gpurun_out/$name.log" 2>&1
    local rc=$?
    echo "=== $name rc=$rc" | tee -a gpurun_out/session.log
    tail -n 15 "gpurun_out/$name.log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name timed out / was killed: stopping"; exit $rc; fi
    return $rc
}
: > gpurun_out/session.log
step pytest_gpu 900 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider "$@"
step smoke 300 python -c "import __graft_entry__ as g; g.smoke()"
step bench 600 python bench.py --steps 10 --warmup 2
step bench_cfg2 300 python bench.py --config cfg2 --steps 10 --warmup 2 --no-cpu-baseline
step rocprof 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline
ls -R gpurun_out/prof | head -30
