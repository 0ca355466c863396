#!/bin/bash
# is the automatic sample_split (16 at 1080p x 1024) still the best one at six / five waves per SIMD?  explicit splits, product library
mkdir -p gpurun_out
L=gpurun_out/r04am_split_sweep.log; : > $L
for S in 8 16 32 64; do
  echo "== sample_split $S" >> $L
  SPLIT=$S RT_HIP_LIB=$PWD/raytracing-rust_amd/librt_hip.so timeout -k 10 200 python tests/probes/gpu_r04_ab.py 2>&1 | grep -E "ms|rror" >> $L
done
cut -c1-120 $L
