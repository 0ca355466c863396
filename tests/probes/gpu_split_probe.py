"""render-kernel time of the whole config-2 frame on one GPU for several (spp, sample_split) pairs"""
import importlib, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
pkg = importlib.import_module("raytracing-rust_amd"); hb = importlib.import_module("raytracing-rust_amd.hip_backend"); abi = pkg.abi
import scenes
ls = scenes.load_ssml("rtweekend1"); g = hb.HipScene(ls.scene); cam = hb.camera_new(**ls.camera_params)
for spp, S in ((1024, 1), (1024, 2), (1024, 4), (1024, 8), (1024, 16), (1024, 64), (2048, 1), (512, 1), (256, 1), (1024, 1)):
    o = abi.default_render_opts(1920, 1080, spp); o.sample_split = S
    g.render(cam, o); g.render(cam, o)
    ms = g.last_kernel_ms()[0]
    print(f"spp {spp:5d} split {S:3d}: kernel {ms:7.2f} ms  {ms/spp*1024:7.2f} ms per 1024 spp", flush=True)
