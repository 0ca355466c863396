#!/bin/bash
# round 4, first GPU call: the parity suite on the new build, then the VALU issue microbenchmark with in-kernel cycle counters
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r04a
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04a/pytest.log 2>&1 || { tail -30 gpurun_out/r04a/pytest.log; exit 1; }
tail -3 gpurun_out/r04a/pytest.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_issue tests/probes/microbench/valu_issue.hip && timeout -k 10 300 /tmp/valu_issue > gpurun_out/r04a/valu_issue.txt 2>&1
tail -20 gpurun_out/r04a/valu_issue.txt
