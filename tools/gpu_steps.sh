#!/bin/bash
# Runs the given commands (one per argument) in order on the GPU box, each under its own timeout, logging to gpurun_out/<tag>_<n>.log;
# stops at the first one that timed out or was killed (no further GPU step after a hang), keeps going after ordinary failures.
# usage: tools/gpu_steps.sh TAG SECONDS "cmd 1" "cmd 2" ...
tag=$1; lim=$2; shift 2
mkdir -p gpurun_out
n=0
for cmd in "$@"; do
  n=$((n+1))
  echo "=== [$n] $cmd" | tee gpurun_out/${tag}_${n}.log
  timeout -k 10 "$lim" bash -c "$cmd" >> gpurun_out/${tag}_${n}.log 2>&1
  rc=$?
  echo "=== rc $rc" | tee -a gpurun_out/${tag}_${n}.log
  tail -n 25 gpurun_out/${tag}_${n}.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $n timed out: stopping"; exit 1; fi
done
exit 0
