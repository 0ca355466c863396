#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun):
#   tests/run_rocprof.sh <tag>      -> gpurun_out/prof_<tag>/{trace,pmc_*}/...
# kernel-trace + stats in one run; every PMC set in its own run (no other trace domains).
set -o pipefail
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace_bench.json 2> $OUT/trace.err || exit 1
BENCH1="python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline"
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc_$i -- $BENCH1 > $OUT/pmc_$i.json 2> $OUT/pmc_$i.err || echo "pmc set $i failed: $SET" >> $OUT/failed.txt
  echo "set $i: $SET" >> $OUT/sets.txt
done
find $OUT -name "*.csv" | head -50 > $OUT/files.txt
echo done
