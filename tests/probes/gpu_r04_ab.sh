#!/bin/bash
# same-box A/B, interleaved: tests/probes/gpu_r04_ab.sh <tag> <rounds> lib1.so lib2.so ...   (libs under raytracing-rust_amd/)
TAG=$1; ROUNDS=$2; shift; shift
mkdir -p gpurun_out
{
for ROUND in $(seq 1 $ROUNDS); do
for L in "$@"; do
  echo "== $L round $ROUND"
  RT_HIP_LIB=$PWD/raytracing-rust_amd/$L timeout -k 10 200 python tests/probes/gpu_r04_ab.py $SCENES 2>&1 | grep -E "ms|rror"
done
done
} | tee gpurun_out/${TAG}_ab.log
