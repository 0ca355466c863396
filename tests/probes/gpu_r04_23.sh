#!/bin/bash
# coarse simple kernels at five waves (t5), + full-feature kernels at a five-wave budget (t5f5), against four (t4): every small-scene kernel
mkdir -p gpurun_out
L=gpurun_out/r04ae_small_ab.log; : > $L
for R in 1 2; do
  for B in ab_full_t4.so ab_full_t5.so ab_full_t5f5.so; do
    echo "== $B round $R" >> $L
    RT_HIP_LIB=$PWD/raytracing-rust_amd/$B timeout -k 10 300 python tests/probes/gpu_r04_small_ab.py 2>&1 | grep -E "ms|rror" >> $L || exit 1
  done
done
cut -c1-190 $L
