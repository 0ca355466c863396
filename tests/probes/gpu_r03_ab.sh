#!/bin/bash
# round 3 GPU call: the whole GPU suite on the default library, then same-box A/B of alternate builds on the four bench workloads
#   tests/probes/gpu_r03_ab.sh <tag> lib1.so lib2.so ...     (libs relative to raytracing-rust_amd/)
set -o pipefail
TAG=$1; shift
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/${TAG}_tests.log
tail -4 gpurun_out/${TAG}_tests.log
{
for ROUND in 1 2; do
for L in "$@"; do
  echo "== $L (small scenes, 256 spp)"
  RT_HIP_LIB=$PWD/raytracing-rust_amd/$L timeout -k 10 200 python tests/probes/gpu_perf_probe.py 256 2>&1 | grep -E "spp:|False"
done
done
for L in "$@"; do
  echo "== $L (1 M triangles, 8 spp, naive then MIS)"
  for M in 0 1; do RT_HIP_LIB=$PWD/raytracing-rust_amd/$L timeout -k 10 300 python tests/probes/gpu_mesh_bench.py 1000000 1920 1080 8 $M 3 2>&1 | tail -1; done
done
} | tee gpurun_out/${TAG}_ab.log
