#!/bin/bash
PY=$(python3 -c 'import os,sys;print(os.path.realpath(sys.executable))')  # the interpreter itself: no shim / wrapper exec after the profiler has initialised the GPU
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
for V in "" "shuffle"; do
  OUT=$R/gpurun_out/prof_trace_$V; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- $PY $R/tests/probes/gpu_trace_bench.py ${1:-1000000} 1920 1080 $V > $OUT/out.txt 2>&1
  grep traversal $OUT/out.txt
  grep check_hit $OUT/*/*kernel_stats.csv | sed 's/.*)",//'
done
