#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
SCENES="overshadowed" bash tests/probes/gpu_r04_ab.sh r04o 3 ab_y_prev.so ab_y_base.so
