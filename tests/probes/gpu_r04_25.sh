#!/bin/bash
# config 3 with the tiny scene staged in LDS (default) against read from global memory, product library, interleaved
mkdir -p gpurun_out
L=gpurun_out/r04ao_scene_lds.log; : > $L
for R in 1 2; do for V in 1 0; do
  echo "== scene_in_lds $V round $R" >> $L
  SCENE_LDS=$V SCENES=overshadowed RT_HIP_LIB=$PWD/raytracing-rust_amd/librt_hip.so timeout -k 10 200 python tests/probes/gpu_r04_ab.py 2>&1 | grep -E "ms|rror" >> $L
done; done
cut -c1-110 $L
