#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tests/probes/gpu_r04_ab.sh r04l 2 ab_w_base.so ab_w_thr40.so ab_w_thr44.so ab_w_thr52.so ab_w_thr56.so ab_w_starve64.so ab_w_starve256.so
