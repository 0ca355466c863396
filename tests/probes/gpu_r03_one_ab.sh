#!/bin/bash
# same-box A/B of builds that hold only the headline kernels (make EXTRA=-DRT_ONLY_HEADLINE: 12 s instead of 95) on BASELINE config 2
# (and 3 with SCENES="rtweekend1 overshadowed") at 256 spp, interleaved: tests/probes/gpu_r03_one_ab.sh <tag> <rounds> lib1.so lib2.so ...
TAG=$1; ROUNDS=$2; shift; shift
{
for ROUND in $(seq 1 $ROUNDS); do
for L in "$@"; do
  for SC in ${SCENES:-rtweekend1}; do
    echo -n "$L round $ROUND: "
    RT_HIP_LIB=$PWD/raytracing-rust_amd/$L timeout -k 10 100 python tests/probes/gpu_perf_probe_one.py ${SPP:-256} $SC 2>&1 | grep -E "kernel|rror"
  done
done
done
} | tee gpurun_out/${TAG}_one_ab.log
