"""a big mesh that needs the full-feature kernels (one glass sphere and one GGX sphere among 200 k Lambertian triangles)"""
import importlib, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
pkg = importlib.import_module("raytracing-rust_amd"); hb = importlib.import_module("raytracing-rust_amd.hip_backend"); abi = pkg.abi
import scenes
sc = scenes.random_triangle_mesh(200000, seed=42, edge=0.3)
sc.sphere((0.0, -12.0, 0.0), 1.5, sc.refract(sc.solid((1.0, 1.0, 1.0)), 1.5))
sc.sphere((3.0, -12.0, 0.0), 1.0, sc.trowbridge_reitz(sc.solid((0.9, 0.7, 0.4)), 0.4, (1.5, 1.5, 1.5), 1.0))
g = hb.HipScene(sc); cam = hb.camera_new(**scenes.MESH_CAMERA)
for method in (1, 0):
    o = abi.default_render_opts(1920, 1080, 8, method=method)
    g.render(cam, o); g.render(cam, o)
    print(f"method {method}: kernel {g.last_kernel_ms()[0]:.2f} ms", flush=True)
