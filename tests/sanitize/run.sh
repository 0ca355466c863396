#!/bin/bash
# AddressSanitizer + UBSan over the CPU-side code (GPU sanitizers are not available on the pool):
#   1. the oracle (gcc) under the oracle test files
#   2. the product's host scene build (csrc/rt_build.cpp, clang host-only) on random, degenerate and large scenes,
#      checked against the oracle's tree while it runs
# Run from the repo root:  bash tests/sanitize/run.sh
set -e
R=$(pwd); T=/tmp/rt_sanitize; mkdir -p $T
gcc -O1 -g -std=c11 -D_GNU_SOURCE -fPIC -ffp-contract=off -fno-fast-math -fno-math-errno -mfma -mavx2 -fsanitize=address,undefined \
    -fno-omit-frame-pointer -shared -o $T/liboracle_asan.so oracle/ora_geometry.c oracle/ora_bvh.c oracle/ora_shading.c oracle/ora_render.c -lm -lpthread
ORACLE_LIB=$T/liboracle_asan.so LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0 \
    python -m pytest tests/test_oracle_units.py tests/test_oracle_bvh.py tests/test_oracle_render.py tests/test_oracle_statistics.py -x -q -p no:cacheprovider 2>&1 | tee $T/oracle.log | tail -2
! grep -q "runtime error\|AddressSanitizer" $T/oracle.log
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 --cuda-host-only -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fsanitize=address,undefined \
    -fno-omit-frame-pointer -shared-libsan -I$R/raytracing-rust_amd/csrc -shared -o $T/libhostbuild_asan.so tests/sanitize/host_build_wrap.cpp raytracing-rust_amd/csrc/rt_build.cpp
LD_PRELOAD=$(/opt/rocm/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0 python tests/sanitize/host_build_under_asan.py 2>&1 | tee $T/host.log | tail -2
! grep -q "runtime error\|AddressSanitizer" $T/host.log
#   3. the whole library with the HOST code instrumented (the device code is compiled as usual), under the CPU tests that
#      go through it: output stage writers, ABI checks, host-only scenes, the .ssml / OBJ reader
make -C raytracing-rust_amd/csrc OUT=$T/librt_hip_asan.so EXTRA="-fsanitize=address,undefined -shared-libsan -fno-omit-frame-pointer -g" > $T/lib_build.log 2>&1
RT_HIP_LIB=$T/librt_hip_asan.so RT_HIP_NO_TORCH_PRELOAD=1 LD_PRELOAD=$(/opt/rocm/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0 \
    python -m pytest tests/test_output.py tests/test_abi.py tests/test_host_bvh.py tests/test_ssml.py -x -q -p no:cacheprovider 2>&1 | tee $T/lib.log | tail -2
! grep -q "runtime error\|AddressSanitizer" $T/lib.log
echo "sanitizers: clean"
