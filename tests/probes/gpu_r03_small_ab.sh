#!/bin/bash
# same-box A/B of alternate builds on BASELINE configs 2 and 3 (256 spp), interleaved: tests/probes/gpu_r03_small_ab.sh <tag> <rounds> lib1.so lib2.so ...
TAG=$1; ROUNDS=$2; shift; shift
{
for ROUND in $(seq 1 $ROUNDS); do
for L in "$@"; do
  echo "== $L round $ROUND"
  RT_HIP_LIB=$PWD/raytracing-rust_amd/$L timeout -k 10 200 python tests/probes/gpu_perf_probe.py 256 2>&1 | grep -E "spp:|False"
done
done
} | tee gpurun_out/${TAG}_small_ab.log
