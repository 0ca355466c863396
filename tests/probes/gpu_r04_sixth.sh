#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tests/probes/gpu_r04_ab.sh r04f 2 ab_ntrace.so ab_noslp.so
