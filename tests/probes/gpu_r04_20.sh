#!/bin/bash
# issue microbenchmark at 5 and 6 waves per SIMD (the config-2 kernel runs at 6 since round 4's loop restructure) + section clocks of the final kernels
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_issue tests/probes/microbench/valu_issue.hip && timeout -k 10 300 /tmp/valu_issue 5 6 > gpurun_out/r04z_valu_issue_w5_w6.txt 2>&1
grep -E "v_fma_f32  |s_alu|v_rcp|v_mul_lo|v_pk" gpurun_out/r04z_valu_issue_w5_w6.txt | cut -c1-150
RT_HIP_LIB=$PWD/raytracing-rust_amd/ab_stats_nohist.so timeout -k 10 300 python tests/probes/gpu_stats_probe.py 1024 0 > gpurun_out/r04z_section_shares_1024spp.txt 2>&1
cat gpurun_out/r04z_section_shares_1024spp.txt | cut -c1-900
