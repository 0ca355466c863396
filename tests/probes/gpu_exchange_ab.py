"""RT_TUNE_EXCHANGE on/off on the two small-scene workloads, same box, same library: kernel time, bit-equality of the
frames and (with the stats build, RT_HIP_LIB=.../librt_hip_stats.so) the lanes per super-phase.
python tests/probes/gpu_exchange_ab.py [spp]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("raytracing-rust_amd"); hb = importlib.import_module("raytracing-rust_amd.hip_backend")
import scenes
abi = pkg.abi
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for name in ("rtweekend1", "overshadowed"):
    ls = scenes.load_ssml(name)
    hs = hb.HipScene(ls.scene); cam = hb.camera_new(**ls.camera_params)
    frames = {}
    for xchg in (0, 1, 0, 1):
        hs.set_tuning(abi.RT_TUNE_EXCHANGE, xchg)
        opts = abi.default_render_opts(1920, 1080, spp, method=1)
        img, rays = hs.render(cam, opts)
        best = 1e9
        for _ in range(3):
            hs.render(cam, opts); best = min(best, hs.last_kernel_ms()[0])
        li = hs.last_launch_info()
        same = "" if xchg not in frames else f" same as first: {np.array_equal(frames[xchg][0], img) and frames[xchg][1] == rays}"
        if 1 - xchg in frames:
            same += f" == other mode: {np.array_equal(frames[1 - xchg][0], img) and frames[1 - xchg][1] == rays}"
        frames[xchg] = (img, rays)
        print(f"{name} xchg={xchg} {spp}spp: kernel {best:.2f} ms {1920*1080*spp/best/1e3:.1f} Msamples/s  lds {li['lds_bytes']} blocks/cu {li['blocks_per_cu']}{same}", flush=True)
