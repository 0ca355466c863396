#!/bin/bash
# A/B builds on the 10M-triangle mesh (1080p, 4 spp): tests/probes/gpu_ab_mesh10m.sh lib1.so lib2.so ...
for L in "$@"; do
  echo "== $L"
  for M in 0 1; do RT_HIP_LIB=$PWD/raytracing-rust_amd/$L python tests/probes/gpu_mesh_bench.py 10000000 1920 1080 4 $M 2 2>&1 | tail -1; done
done
