#!/bin/bash
# round-3 evidence, step 1: rocprofv3 stats + PMC of each workload (tests/probes/run_rocprof.sh).  Summarise afterwards, off the
# box, with tests/probes/summarize_rocprof.py <tag>_<workload> <workload>; then step 2 (bench lines): gpu_r03_bench_lines.sh
#   tests/probes/gpu_r03_evidence.sh <tag> <workload> ...
TAG=$1; shift
mkdir -p gpurun_out
for WL in "$@"; do
  echo "== $WL: rocprof"
  bash tests/probes/run_rocprof.sh ${TAG}_$WL $WL 3 > gpurun_out/${TAG}_${WL}_rocprof.log 2>&1; tail -1 gpurun_out/${TAG}_${WL}_rocprof.log
done
