#!/bin/bash
# same-box A/B on the 1 M-triangle mesh with the builds interleaved and repeated (single numbers on this workload wander by
# several per cent): tests/probes/gpu_r03_mesh_ab.sh <tag> lib1.so lib2.so ...
TAG=$1; shift
{
for ROUND in $(seq 1 ${ROUNDS:-3}); do
for L in "$@"; do
  echo "== $L round $ROUND (1 M triangles, 8 spp, naive / MIS: best of 4)"
  for M in ${METHODS:-0 1}; do RT_HIP_LIB=$PWD/raytracing-rust_amd/$L timeout -k 10 300 python tests/probes/gpu_mesh_bench.py 1000000 1920 1080 8 $M 4 2>&1 | grep kernel | sort -k2 -n | head -1; done
done
done
} | tee gpurun_out/${TAG}_mesh_ab.log
