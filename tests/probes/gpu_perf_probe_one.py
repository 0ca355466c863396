"""one render of BASELINE configs[1] (rtweekend1, 1080p, MIS) at a given depth, for profilers: python gpu_perf_probe_one.py [spp]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("raytracing-rust_amd"); hb = importlib.import_module("raytracing-rust_amd.hip_backend")
import scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
name = sys.argv[2] if len(sys.argv) > 2 else "rtweekend1"
ls = scenes.load_ssml(name)
g = hb.HipScene(ls.scene); cam = hb.camera_new(**ls.camera_params)
o = pkg.abi.default_render_opts(1920, 1080, spp, method=1, seed=1)
g.render(cam, o)
print(f"{name} {spp} spp: kernel {g.last_kernel_ms()[0]:.2f} ms")
