#!/bin/bash
# coarse spheres-only kernels at a six-wave register budget (ab_full_sw6) against four (ab_full_sw4): every small-scene kernel, same box, interleaved
mkdir -p gpurun_out
L=gpurun_out/r04ac_small_ab.log; : > $L
for R in 1 2; do
  for B in ab_full_sw4.so ab_full_sw6.so; do
    echo "== $B round $R" >> $L
    RT_HIP_LIB=$PWD/raytracing-rust_amd/$B timeout -k 10 300 python tests/probes/gpu_r04_small_ab.py 2>&1 | grep -E "ms|rror" >> $L || exit 1
  done
done
cut -c1-190 $L
