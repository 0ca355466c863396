#!/bin/bash
# Collects the rocprofv3 evidence for one bench.py workload on the GPU box (run through gpurun):
#   tests/probes/run_rocprof.sh <tag> [workload] [steps for the stats run]  -> gpurun_out/prof_<tag>/{trace,pmc_*}/...
# kernel-trace + stats in one run; every PMC set in its own run (no other trace domains), as
# MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE cannot share a pass).
PY=$(python3 -c 'import os,sys;print(os.path.realpath(sys.executable))')  # the interpreter itself: no shim / wrapper exec after the profiler has initialised the GPU
set -o pipefail
TAG=${1:-r02}
WL=${2:-rtweekend1}
STEPS=${3:-3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
python3 -c "import sys; sys.path.insert(0, '$R'); import bench; print(bench.source_hash())" > $OUT/source_hash.txt
echo "$WL" > $OUT/workload.txt
BENCH="$PY $R/bench.py --workload $WL --steps $STEPS --warmup 1 --no-cpu-baseline --no-walk-stats"
echo "interpreter: $PY; rocprofv3 --kernel-trace --stats -- python3 bench.py --workload $WL --steps $STEPS --warmup 1 --no-cpu-baseline --no-walk-stats (stats); rocprofv3 --pmc <set> --kernel-trace -- python3 bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --no-walk-stats (one run per set)" > $OUT/command.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace_bench.json 2> $OUT/trace.err || exit 1
BENCH1="$PY $R/bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --no-walk-stats"
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_WR SQ_INSTS_FLAT"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc_$i -- $BENCH1 > $OUT/pmc_$i.json 2> $OUT/pmc_$i.err || echo "pmc set $i failed: $SET" >> $OUT/failed.txt
  echo "set $i: $SET" >> $OUT/sets.txt
done
echo "done $TAG $WL"
