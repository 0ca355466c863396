import importlib, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
pkg = importlib.import_module("raytracing-rust_amd"); hb = importlib.import_module("raytracing-rust_amd.hip_backend"); abi = pkg.abi
import scenes
for name in ("rtweekend1", "overshadowed"):
    ls = scenes.load_ssml(name); g = hb.HipScene(ls.scene); cam = hb.camera_new(**ls.camera_params)
    for trav in (0, 1):
        g.set_tuning(abi.RT_TUNE_TRAVERSAL, trav)
        o = abi.default_render_opts(1920, 1080, 256)
        g.render(cam, o); g.render(cam, o)
        print(name, "schedule", "fine" if trav else "coarse", "kernel ms", round(g.last_kernel_ms()[0], 2), flush=True)
