#!/bin/bash
# round 4 evidence on the FINAL build: GPU suite, rocprofv3 stats + PMC of the four workloads (+ the 500-sphere scene), then
# (tests/probes/gpu_r04_evidence_b.sh, after kernel_counters.json has been refreshed from these) the bench lines
set -o pipefail
TAG=${TAG:-r04z}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/${TAG}
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/${TAG}/pytest.log
tail -3 gpurun_out/${TAG}/pytest.log
for WL in rtweekend1 overshadowed spheres500 mesh1m mesh10m; do
  rm -rf gpurun_out/prof_${TAG}_$WL
  bash tests/probes/run_rocprof.sh ${TAG}_$WL $WL 3 > gpurun_out/${TAG}/rocprof_$WL.log 2>&1; tail -1 gpurun_out/${TAG}/rocprof_$WL.log
done
