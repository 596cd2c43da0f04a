# kernel timeline of the last iteration: tools/r05_trace.sh CONFIG [ENV=VAL ...]
set -o pipefail
cfg=${1:-cfg3}; shift
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rm -rf gpurun_out/prof_tl
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tl -- python3 bench.py --config $cfg --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r05_trace_bench.log 2>&1 || exit 1
f=$(find gpurun_out/prof_tl -name "*kernel_trace.csv" | head -1)
python3 tools/iteration_timeline.py $f
