#!/bin/bash
# round 4, third GPU call: per-section wall-clock shares of configs 2 and 3 at the bench's own depth and split (-DRT_STATS build)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04c
mkdir -p $O
cd $R
RT_HIP_LIB=$R/raytracing-rust_amd/librt_hip_stats.so timeout -k 10 300 python tests/probes/gpu_stats_probe.py 1024 16 > $O/stats_1024_split16.txt 2>&1; cat $O/stats_1024_split16.txt
RT_HIP_LIB=$R/raytracing-rust_amd/librt_hip_stats.so timeout -k 10 300 python tests/probes/gpu_stats_probe.py 1024 1 > $O/stats_1024_split1.txt 2>&1; cat $O/stats_1024_split1.txt
RT_HIP_LIB=$R/raytracing-rust_amd/librt_hip_stats.so timeout -k 10 300 python tests/probes/gpu_hist_probe.py 1024 16 > $O/hist.txt 2>&1; tail -30 $O/hist.txt
