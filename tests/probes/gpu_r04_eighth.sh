#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tests/probes/gpu_r04_ab.sh r04h 2 ab_noslp.so ab_v_rolled.so ab_v_unroll.so ab_v_hoist.so ab_v_sched_default.so ab_v_sched_ilp.so ab_v_ifcvt.so ab_v_ifcvt8.so ab_v_nolicm.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/random_fetch tests/probes/microbench/random_fetch.hip 2>/dev/null && timeout -k 10 300 /tmp/random_fetch > gpurun_out/r04h_random_fetch.txt 2>&1; cat gpurun_out/r04h_random_fetch.txt
