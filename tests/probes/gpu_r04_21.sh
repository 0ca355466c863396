#!/bin/bash
# wave-uniform walk of tiny trees: parity on both headline scenes (small frames, both integrators), then same-box A/B at full size
mkdir -p gpurun_out
RT_HIP_LIB=$PWD/raytracing-rust_amd/ab_u_new.so timeout -k 10 300 python tests/probes/gpu_perf_probe.py 64 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04y_parity.log
grep -q "bit-exact: False" gpurun_out/r04y_parity.log && exit 1
bash tests/probes/gpu_r04_ab.sh r04y 2 ab_u_prev.so ab_u_new.so
