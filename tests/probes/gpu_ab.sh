#!/bin/bash
# A/B alternate builds of the library on the GPU box: tests/probes/gpu_ab.sh <spp> lib1.so lib2.so ...
SPP=$1; shift
for L in "$@"; do
  echo "== $L"
  RT_HIP_LIB=$PWD/raytracing-rust_amd/$L python tests/probes/gpu_perf_probe.py $SPP 2>&1 | grep -E "spp:|False"
done
