#!/bin/bash
# A/B builds on the 1M-triangle mesh: tests/probes/gpu_ab_mesh.sh lib1.so lib2.so ...
for L in "$@"; do
  echo "== $L"
  for M in 0 1; do RT_HIP_LIB=$PWD/raytracing-rust_amd/$L python tests/probes/gpu_mesh_bench.py 1000000 1920 1080 8 $M 2 2>&1 | tail -1; done
done
