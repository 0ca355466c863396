#!/bin/bash
# same-box A/B of alternate builds on the FULL-feature kernels (all five materials and textures, coarse schedule; 200 k triangles plus a
# glass and a GGX sphere, fine schedule), interleaved: tests/probes/gpu_r03_full_ab.sh <tag> <rounds> lib1.so lib2.so ...
TAG=$1; ROUNDS=$2; shift; shift
{
for ROUND in $(seq 1 $ROUNDS); do
for L in "$@"; do
  echo "== $L round $ROUND"
  for M in 1 0; do RT_HIP_LIB=$PWD/raytracing-rust_amd/$L timeout -k 10 200 python tests/probes/gpu_scene_bench.py all_materials 1920 1080 64 $M 3 2>&1 | grep kernel | sort -k5 -n | head -1; done
  RT_HIP_LIB=$PWD/raytracing-rust_amd/$L timeout -k 10 200 python tests/probes/gpu_full_mesh_ab.py 2>&1 | grep kernel
done
done
} | tee gpurun_out/${TAG}_full_ab.log
