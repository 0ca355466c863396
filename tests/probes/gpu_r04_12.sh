#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tests/probes/gpu_r04_ab.sh r04m 3 ab_v_sums.so ab_w_base.so ab_v_coord.so
