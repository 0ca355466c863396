#!/bin/bash
# round-3 evidence, step 2: the bench line of each workload (default steps), kept under gpurun_out/<tag>_bench_<workload>.json
TAG=$1; shift
mkdir -p gpurun_out
for WL in "$@"; do
  echo "== $WL: bench"
  timeout -k 10 900 python bench.py --workload $WL > gpurun_out/${TAG}_bench_$WL.json 2> gpurun_out/${TAG}_bench_$WL.err; echo "rc=$?"; cut -c1-260 gpurun_out/${TAG}_bench_$WL.json
done
