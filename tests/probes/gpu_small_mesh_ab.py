"""coarse-schedule kernels on small triangle scenes (A/B builds via RT_HIP_LIB)"""
import importlib, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
pkg = importlib.import_module("raytracing-rust_amd"); hb = importlib.import_module("raytracing-rust_amd.hip_backend"); abi = pkg.abi
import scenes
for n, edge in ((50, 3.0), (300, 2.0), (1000, 1.0), (2000, 1.0)):
    g = hb.HipScene(scenes.random_triangle_mesh(n, seed=42, edge=edge, emissive_every=50))
    cam = hb.camera_new(**scenes.MESH_CAMERA)
    row = []
    for method in (1, 0):
        o = abi.default_render_opts(1920, 1080, 32, method=method)
        g.render(cam, o); g.render(cam, o)
        row.append(g.last_kernel_ms()[0])
    print(f"tris={n:6d}: MIS {row[0]:8.2f} ms  naive {row[1]:8.2f} ms", flush=True)
