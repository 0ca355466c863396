#!/bin/bash
# round 4, second GPU call: the new multi-device / split tests, the general-kernel numbers, the per-section wall-clock shares of
# configs 2 and 3 from the -DRT_STATS build
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04b
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "multi_device or full_size or work_claims or graph or bench_prints" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
timeout -k 10 300 python tests/probes/gpu_r04_general.py > $O/general.txt 2> $O/general.err; cat $O/general.txt
RT_HIP_LIB=$R/raytracing-rust_amd/librt_hip_stats.so timeout -k 10 300 python tests/probes/gpu_stats_probe.py 64 16 > $O/stats_split16.txt 2>&1; cat $O/stats_split16.txt
