#!/bin/bash
# round 4, fourth GPU call: same-box A/B of the verified constant divisions on configs 2 and 3; section clocks without the histogram atomics
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tests/probes/gpu_r04_ab.sh r04d 3 ab_base.so ab_constdiv.so
RT_HIP_LIB=$R/raytracing-rust_amd/ab_stats_nohist.so timeout -k 10 300 python tests/probes/gpu_stats_probe.py 1024 16 > gpurun_out/r04d_stats_nohist.txt 2>&1; cat gpurun_out/r04d_stats_nohist.txt
