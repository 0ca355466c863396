#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/r04w_small_ab.log; : > $L
for R in 1 2; do
  for B in ab_full_head.so librt_hip.so; do
    echo "== $B round $R" >> $L
    RT_HIP_LIB=$PWD/raytracing-rust_amd/$B timeout -k 10 300 python tests/probes/gpu_r04_small_ab.py 2>&1 | grep -E "ms|rror" >> $L || exit 1
  done
done
cat $L
L=gpurun_out/r04w_mesh_ab.log; : > $L
for R in 1 2; do
  for B in ab_full_head.so librt_hip.so; do
    echo "== $B round $R" >> $L
    RT_HIP_LIB=$PWD/raytracing-rust_amd/$B timeout -k 10 400 python tests/probes/gpu_r04_mesh_ab.py mesh1m mesh10m 2>&1 | grep -E "ms|rror" >> $L || exit 1
  done
done
cat $L
