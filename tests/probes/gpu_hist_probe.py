"""Lane-count histograms of the coarse schedule from the -DRT_STATS build (RT_HIP_LIB=.../librt_hip_stats.so):
python tests/probes/gpu_hist_probe.py spp split [scene]"""
import ctypes as C, importlib, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
pkg = importlib.import_module("raytracing-rust_amd"); hb = importlib.import_module("raytracing-rust_amd.hip_backend"); abi = pkg.abi
import scenes
name = sys.argv[3] if len(sys.argv) > 3 else "rtweekend1"
ls = scenes.load_ssml(name); g = hb.HipScene(ls.scene)
cp = dict(ls.camera_params)
if len(sys.argv) > 4:  # look somewhere else: "down" = at the ground only, "up" = at the sky only
    cp["lookat"] = {"down": (0.0, 0.3, -1.0), "up": (0.0, 0.3, 1.0)}[sys.argv[4]]
    cp["fov"] = 40.0
cam = hb.camera_new(**cp)
o = abi.default_render_opts(1920, 1080, int(sys.argv[1])); o.sample_split = int(sys.argv[2])
out = (C.c_ulonglong * 260)()
hb.lib().rt_debug_hist(out, 1)
g.render(cam, o)
hb.lib().rt_debug_hist(out, 1)
for k, what in enumerate(("PRIMARY iterations by participating lanes", "... by lanes waiting for BOUNCE then", "acquire events by needy lanes", "BOUNCE iterations by lanes")):
    h = [out[k * 65 + i] for i in range(65)]; tot = max(1, sum(h))
    print(what + f" (total {sum(h)}):")
    print("   " + " ".join(f"{i}:{100*v/tot:.1f}" for i, v in enumerate(h) if v * 200 > tot))
