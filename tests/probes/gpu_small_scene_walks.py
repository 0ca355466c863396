import importlib, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
pkg = importlib.import_module("raytracing-rust_amd"); hb = importlib.import_module("raytracing-rust_amd.hip_backend"); abi = pkg.abi
import scenes
cases = [("overshadowed", scenes.load_ssml("overshadowed").scene, scenes.load_ssml("overshadowed").camera_params)]
for n in (30, 60, 100):
    cases.append((f"tris{n}", scenes.random_triangle_mesh(n, seed=42, edge=3.0), scenes.MESH_CAMERA))
for name, sc, camp in cases:
    g = hb.HipScene(sc); cam = hb.camera_new(**camp)
    row = []
    for trav in (0, 1):
        g.set_tuning(abi.RT_TUNE_TRAVERSAL, trav)
        o = abi.default_render_opts(1920, 1080, 64, method=1)
        g.render(cam, o); g.render(cam, o)
        row.append(g.last_kernel_ms()[0])
    print(f"{name}: MIS 64spp exhaustive two-child {row[0]:.2f} ms, pruned wide {row[1]:.2f} ms", flush=True)
