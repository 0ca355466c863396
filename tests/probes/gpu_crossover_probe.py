import importlib, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
pkg = importlib.import_module("raytracing-rust_amd"); hb = importlib.import_module("raytracing-rust_amd.hip_backend"); abi = pkg.abi
import scenes
cam_params = dict(origin=(0.0, -30.0, 0.0), lookat=(0, 0, 0), vup=(0, 0, 1), fov=50.0, aspect_ratio=16 / 9, aperture=0.0, focus_dist=10.0)
for n in (8, 32, 64, 128, 256, 512, 1024, 2048, 16384):
    g = hb.HipScene(scenes.random_spheres(n, seed=1, emissive_every=0, sampler_res=(100, 100)))
    cam = hb.camera_new(**cam_params)
    row = []
    for method in (1, 0):
        for (trav, sched) in ((0, 0), (1, 0), (1, 1)):
            g.set_tuning(abi.RT_TUNE_TRAVERSAL, trav)
            g.set_tuning(abi.RT_TUNE_SCHEDULE, sched)
            o = abi.default_render_opts(1920, 1080, 32, method=method)
            g.render(cam, o); g.render(cam, o)
            row.append(g.last_kernel_ms()[0])
    print(f"n={n:6d}  MIS: exhaustive+coarse {row[0]:8.2f} pruned+coarse {row[1]:8.2f} pruned+fine {row[2]:8.2f} | naive: {row[3]:8.2f} {row[4]:8.2f} {row[5]:8.2f}", flush=True)
