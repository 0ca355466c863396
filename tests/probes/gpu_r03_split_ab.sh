#!/bin/bash
# same-box A/B of headline-only builds on BASELINE config 2 at 1024 spp for several sample_split values:
#   tests/probes/gpu_r03_split_ab.sh <tag> "<splits>" lib1.so lib2.so ...
TAG=$1; SPLITS=$2; shift; shift
{
for L in "$@"; do
  RT_HIP_LIB=$PWD/raytracing-rust_amd/$L timeout -k 10 200 python - "$L" $SPLITS <<'P'
import importlib, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
pkg = importlib.import_module("raytracing-rust_amd"); hb = importlib.import_module("raytracing-rust_amd.hip_backend"); abi = pkg.abi
import scenes
ls = scenes.load_ssml("rtweekend1"); g = hb.HipScene(ls.scene); cam = hb.camera_new(**ls.camera_params)
for S in [int(x) for x in sys.argv[2:]]:
    o = abi.default_render_opts(1920, 1080, 1024); o.sample_split = S
    g.render(cam, o); best = 1e9
    for _ in range(3):
        g.render(cam, o); best = min(best, g.last_kernel_ms()[0])
    print(f"{sys.argv[1]} split {S:2d}: kernel {best:7.2f} ms", flush=True)
P
done
} | tee gpurun_out/${TAG}_split_ab.log
