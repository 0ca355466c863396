#!/bin/bash
# small-scene kernels beyond the headline pair, three full builds, same box, interleaved
mkdir -p gpurun_out
L=gpurun_out/r04s_small_ab.log; : > $L
for R in 1 2; do
  for B in ab_full_old.so ab_full_sph.so ab_full_all.so; do
    echo "== $B round $R" >> $L
    RT_HIP_LIB=$PWD/raytracing-rust_amd/$B timeout -k 10 300 python tests/probes/gpu_r04_small_ab.py 2>&1 | grep -E "ms|rror" >> $L || exit 1
  done
done
cat $L
