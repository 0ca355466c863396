#!/bin/bash
# PMC + stats for an arbitrary command: tests/probes/run_rocprof_cmd.sh <tag> <python-script-and-args...>
PY=$(python3 -c 'import os,sys;print(os.path.realpath(sys.executable))')  # the interpreter itself: no shim / wrapper exec after the profiler has initialised the GPU
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="$PY $R/$*"
echo "interpreter: $PY; rocprofv3 --kernel-trace --stats -- python3 $* (stats); rocprofv3 --pmc <set> --kernel-trace -- python3 $* (one run per set)" > $OUT/command.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.out 2> $OUT/trace.err || exit 1
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc_$i -- $CMD > $OUT/pmc_$i.out 2> $OUT/pmc_$i.err || echo "pmc set $i failed: $SET" >> $OUT/failed.txt
  echo "set $i: $SET" >> $OUT/sets.txt
done
echo done
