#!/bin/bash
# tests/probes/gpu_r04_19.sh <tag> lib1.so lib2.so ...: mesh1m + mesh10m, two interleaved rounds
TAG=$1; shift
mkdir -p gpurun_out
L=gpurun_out/${TAG}_mesh_ab.log; : > $L
for R in 1 2; do
  for B in "$@"; do
    echo "== $B round $R" >> $L
    RT_HIP_LIB=$PWD/raytracing-rust_amd/$B timeout -k 10 400 python tests/probes/gpu_r04_mesh_ab.py mesh1m mesh10m 2>&1 | grep -E "ms|rror" >> $L || exit 1
  done
done
cat $L
