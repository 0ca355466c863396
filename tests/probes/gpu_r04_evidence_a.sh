#!/bin/bash
# round 4 evidence, part A: the whole GPU suite, then rocprofv3 stats + PMC of configs 2, 3 and 4 on the final build
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r04k
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04k/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r04k/pytest.log
tail -3 gpurun_out/r04k/pytest.log
for WL in rtweekend1 overshadowed mesh1m mesh10m; do
  bash tests/probes/run_rocprof.sh r04k_$WL $WL 3 > gpurun_out/r04k/rocprof_$WL.log 2>&1; tail -1 gpurun_out/r04k/rocprof_$WL.log
done
