#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tests/probes/gpu_r04_ab.sh r04e 2 ab_constdiv.so ab_batch4.so ab_batch8.so ab_ntrace.so ab_ntrace_b4.so
