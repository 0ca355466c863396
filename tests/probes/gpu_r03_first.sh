#!/bin/bash
# round 3, first GPU call: the arithmetic self-test, a parity subset, then the same-box A/B of the round-2 library against the new one
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lean_arithmetic or golden_images or render_matches_oracle or every_kernel_variant" > gpurun_out/r03a_tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r03a_tests.log
tail -5 gpurun_out/r03a_tests.log
for L in ab_r02.so librt_hip.so ab_r02.so librt_hip.so; do
  echo "== $L"
  RT_HIP_LIB=$PWD/raytracing-rust_amd/$L timeout -k 10 200 python tests/probes/gpu_perf_probe.py 256 2>&1 | grep -E "spp:|False"
done | tee gpurun_out/r03a_ab.log
