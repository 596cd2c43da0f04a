set -o pipefail
export TMPDIR=/tmp PAULE_HIP_BWD_PF=8 PAULE_HIP_BWD_PF_DIST=4
rm -rf gpurun_out/prof_tl
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tl -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r05_trace_bench.log 2>&1 || exit 1
f=$(find gpurun_out/prof_tl -name "*kernel_trace.csv" | head -1)
python3 tools/iteration_timeline.py $f | tee gpurun_out/r05_iteration_timeline.txt
