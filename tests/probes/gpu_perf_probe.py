"""Quick perf + parity probe on the GPU box: python tests/probes/gpu_perf_probe.py [spp]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("raytracing-rust_amd"); hb = importlib.import_module("raytracing-rust_amd.hip_backend")
import oracle as O, scenes
abi = pkg.abi
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for name in ("rtweekend1", "overshadowed"):
    ls = scenes.load_ssml(name)
    hs = hb.HipScene(ls.scene); cam = hb.camera_new(**ls.camera_params)
    # parity at small size
    os_ = O.Scene(ls.scene)
    for method in (0, 1):
        o = abi.default_render_opts(320, 180, 8, method=method)
        a, ra = hs.render(cam, o); b, rb = os_.render(O.camera_new(**ls.camera_params), o)
        print(name, "method", method, "bit-exact:", np.array_equal(a, b), ra == rb)
    for method, mname in ((1, "mis"), (0, "naive")):
        opts = abi.default_render_opts(1920, 1080, spp, method=method)
        hs.render(cam, opts)
        best = 1e9
        for _ in range(3):
            hs.render(cam, opts); best = min(best, hs.last_kernel_ms()[0])
        print(f"{name} {mname} {spp}spp: kernel {best:.2f} ms  {1920*1080*spp/best/1e3:.1f} Msamples/s")
