#!/bin/bash
# round 3, call h: VALU issue microbenchmark; Philox A/B on the small scenes; wide-tree line layout A/B (RT_HIP_WIDE_LAYOUT) on 1 M and 10 M triangles
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r03h_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r03h_tests.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/valu_issue tests/probes/microbench/valu_issue.hip && timeout -k 10 120 /tmp/valu_issue | tee gpurun_out/r03h_valu_issue.txt
{
for ROUND in 1 2; do
for L in ab_now.so ab_philox.so; do
  echo "== $L (small scenes, 256 spp)"
  RT_HIP_LIB=$PWD/raytracing-rust_amd/$L timeout -k 10 200 python tests/probes/gpu_perf_probe.py 256 2>&1 | grep -E "spp:|False"
done
done
} | tee gpurun_out/r03h_philox_ab.log
{
for ROUND in 1 2; do
for LAYOUT in 0 1; do
  echo "== line layout $LAYOUT round $ROUND (1 M triangles, 8 spp, naive / MIS: best of 4)"
  for M in 0 1; do RT_HIP_WIDE_LAYOUT=$LAYOUT timeout -k 10 300 python tests/probes/gpu_mesh_bench.py 1000000 1920 1080 8 $M 4 2>&1 | grep kernel | sort -k2 -n | head -1; done
done
done
for LAYOUT in 0 1; do
  echo "== line layout $LAYOUT (10 M triangles, 4 spp, naive / MIS: best of 3)"
  for M in 0 1; do RT_HIP_WIDE_LAYOUT=$LAYOUT timeout -k 10 400 python tests/probes/gpu_mesh_bench.py 10000000 1920 1080 4 $M 3 2>&1 | grep kernel | sort -k2 -n | head -1; done
done
} | tee gpurun_out/r03h_layout_ab.log
