#!/bin/bash
# Static view of the headline kernels (BASELINE configs 2 and 3) without the 98 s full build:
#   tests/probes/isa_headline.sh out.s [extra hipcc flags]  -> assembly + census of the two instantiations
OUT=${1:-/tmp/headline.s}; shift
cd "$(dirname "$0")/../../raytracing-rust_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fno-gpu-rdc -Wno-unused-function \
  -mllvm -amdgpu-sched-strategy=max-memory-clause -DRT_ONLY_HEADLINE --cuda-device-only -S rt_render.hip -o "$OUT" "$@" 2>/dev/null || exit 1
python3 ../../tests/probes/isa_census.py "$OUT" 13render_kernel | grep -E "^ \"_Z|total|\"valu|salu|moves|v_mov|cndmask|lane_spill|\"div\"|useful|branches|sgpr|vgpr|private"
