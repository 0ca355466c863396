import importlib, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
pkg = importlib.import_module("raytracing-rust_amd"); hb = importlib.import_module("raytracing-rust_amd.hip_backend"); abi = pkg.abi
import scenes
for n, edge in ((1000, 1.0), (10000, 0.5), (100000, 0.2), (1000000, 0.05), (200000, 0.3)):
    g = hb.HipScene(scenes.random_triangle_mesh(n, seed=42, edge=edge))
    cam = hb.camera_new(**scenes.MESH_CAMERA)
    row = []
    for method in (1, 0):
        for (trav, sched) in ((1, 0), (1, 1)):
            g.set_tuning(abi.RT_TUNE_TRAVERSAL, trav); g.set_tuning(abi.RT_TUNE_SCHEDULE, sched)
            o = abi.default_render_opts(1920, 1080, 8, method=method)
            g.render(cam, o); g.render(cam, o)
            row.append(g.last_kernel_ms()[0])
    print(f"tris={n:8d} edge={edge}: MIS pruned+coarse {row[0]:8.2f} pruned+fine {row[1]:8.2f} | naive {row[2]:8.2f} {row[3]:8.2f}", flush=True)
