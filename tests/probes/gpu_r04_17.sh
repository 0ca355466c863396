#!/bin/bash
# the round's second kernel generation (loop without untouched-state arms, fused regions everywhere, scalar work range, host-chosen
# workgroup size): GPU suite, then HEAD~ full build against the tree's on the small-scene kernels and the two mesh workloads
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r04v_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04v_pytest.log
L=gpurun_out/r04v_small_ab.log; : > $L
for R in 1 2; do
  for B in ab_full_head.so librt_hip.so; do
    echo "== $B round $R" >> $L
    RT_HIP_LIB=$PWD/raytracing-rust_amd/$B timeout -k 10 300 python tests/probes/gpu_r04_small_ab.py 2>&1 | grep -E "ms|rror" >> $L || exit 1
  done
done
cat $L
L=gpurun_out/r04v_mesh_ab.log; : > $L
for R in 1 2; do
  for B in ab_full_head.so librt_hip.so; do
    echo "== $B round $R" >> $L
    RT_HIP_LIB=$PWD/raytracing-rust_amd/$B timeout -k 10 400 python tests/probes/gpu_r04_mesh_ab.py mesh1m mesh10m 2>&1 | grep -E "ms|rror" >> $L || exit 1
  done
done
cat $L
