"""Renders one of the .ssml scenes a few times (for rocprof): python tests/probes/gpu_scene_bench.py scene W H spp method reps"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("raytracing-rust_amd"); hb = importlib.import_module("raytracing-rust_amd.hip_backend")
import scenes
abi = pkg.abi
name = sys.argv[1]
W, H, SPP, method, reps = (int(x) for x in sys.argv[2:7])
if name == "all_materials":
    g = hb.HipScene(scenes.all_materials())
    cam = hb.camera_new(**scenes.ALL_MATERIALS_CAMERA)
else:
    ls = scenes.load_ssml(name)
    g = hb.HipScene(ls.scene)
    cam = hb.camera_new(**ls.camera_params)
opts = abi.default_render_opts(W, H, SPP, method=method, seed=1)
for _ in range(reps):
    img, rays = g.render(cam, opts)
    ms = g.last_kernel_ms()[0]
    print(f"{name} method {method}: kernel {ms:.1f} ms rays {rays} Msamples/s {W*H*SPP/ms/1e3:.1f}", flush=True)
