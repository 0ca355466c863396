#!/bin/bash
# same-box interleaved A/B of the pinned scene pointers (nodes4 / prims held in scalar registers across the fine schedule's inner
# loops) on the two mesh workloads in bench.py's own configuration; equal checksums = bit-identical frames
mkdir -p gpurun_out
L=gpurun_out/r04p_pin_ab.log; : > $L
for R in 1 2; do
  for B in ab_full_nopin.so ab_full_pin.so; do
    echo "== $B round $R" >> $L
    RT_HIP_LIB=$PWD/raytracing-rust_amd/$B timeout -k 10 400 python tests/probes/gpu_r04_mesh_ab.py mesh1m mesh10m >> $L 2>&1 || exit 1
  done
done
cat $L
