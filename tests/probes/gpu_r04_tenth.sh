#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tests/probes/gpu_r04_ab.sh r04j 2 ab_v_uroot_light.so ab_v_sums.so ab_v_coord.so ab_v_waveguard.so
