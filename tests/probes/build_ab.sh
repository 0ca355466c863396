#!/bin/bash
# builds raytracing-rust_amd/ab_<name>.so holding only the headline kernels (configs 2 and 3 runnable; 14 s instead of 95):
#   tests/probes/build_ab.sh <name> [source dir (default: the tree's csrc)] [extra hipcc flags ...]
NAME=$1; SRC=${2:-$(dirname "$0")/../../raytracing-rust_amd/csrc}; shift; shift
OUT=$(cd "$(dirname "$0")/../../raytracing-rust_amd" && pwd)/ab_$NAME.so
cd "$SRC" || exit 1
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fno-gpu-rdc -Wno-unused-function \
  -mllvm -amdgpu-sched-strategy=max-memory-clause -DRT_ONLY_HEADLINE -DRT_HEADLINE_RUNNABLE "$@" -shared -o "$OUT" rt_api.cpp rt_build.cpp rt_render.hip 2>&1 | grep -E "error|Error"
ls -la "$OUT"
