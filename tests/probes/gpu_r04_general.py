"""Round 4: what the kernels that are NOT compiled for rtweekend1's own tree and materials deliver.
  1. config 2 (rtweekend1 1920x1080x1024 MIS) with RT_TUNE_FEATURE_SET = 0: the general spheres-only kernel on the headline workload
  2. ~500 random spheres (the RTIOW cover shape: tests/scenes.py random_spheres(500), Lerp sky sampled at 100 x 100), 1920x1080x256
     MIS, the library's automatic choices (pruned walk, coarse schedule)
Prints one JSON object per case (kernel, ms, Msamples/s, split)."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("raytracing-rust_amd"); hb = importlib.import_module("raytracing-rust_amd.hip_backend")
import scenes
abi = pkg.abi


def run(what, g, cam, W, H, spp, reps=5):
    o = abi.default_render_opts(W, H, spp, method=abi.RT_METHOD_MIS, seed=1)
    o.sample_split = 0
    best = 1e30
    for _ in range(reps):
        g.render(cam, o)
        best = min(best, g.last_kernel_ms()[0])
    li = g.last_launch_info()
    print(json.dumps({"case": what, "kernel": li["kernel"], "kernel_ms_best_of_%d" % reps: best, "Msamples_per_s": W * H * spp / best / 1e3,
                      "sample_split": li["sample_split"], "waves_per_simd": li["waves_per_simd"], "pruned": li["pruned"], "fine": li["fine"]}), flush=True)


ls = scenes.load_ssml("rtweekend1")
cam = hb.camera_new(**ls.camera_params)
g = hb.HipScene(ls.scene)
run("rtweekend1, the library's choice", g, cam, 1920, 1080, 1024)
g.set_tuning(abi.RT_TUNE_FEATURE_SET, 0)
run("rtweekend1, RT_TUNE_FEATURE_SET=0 (general spheres-only kernel)", g, cam, 1920, 1080, 1024)
sc = scenes.random_spheres(500, seed=7, sampler_res=(100, 100))
cam500 = hb.camera_new(origin=(0.0, -30.0, 6.0), lookat=(0.0, 0.0, 0.0), vup=(0.0, 0.0, 1.0), fov=50.0, aspect_ratio=16.0 / 9.0, aperture=0.0, focus_dist=10.0)
g500 = hb.HipScene(sc)
run("500 random spheres, the library's choice", g500, cam500, 1920, 1080, 256)
g500.set_tuning(abi.RT_TUNE_SCHEDULE, 1)
run("500 random spheres, fine schedule", g500, cam500, 1920, 1080, 256)
