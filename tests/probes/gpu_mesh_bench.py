"""Renders the synthetic random-triangle mesh a few times (for rocprof): python tests/probes/gpu_mesh_bench.py n W H spp method reps
(SPLIT=<S> in the environment: rt_render_opts.sample_split)"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("raytracing-rust_amd"); hb = importlib.import_module("raytracing-rust_amd.hip_backend")
import scenes
abi = pkg.abi
n, W, H, SPP, method, reps = (int(x) for x in sys.argv[1:7])
g = hb.HipScene(scenes.random_triangle_mesh(n, seed=42, extent=10.0 if n <= 2000000 else 20.0))
cam = hb.camera_new(**scenes.MESH_CAMERA)
opts = abi.default_render_opts(W, H, SPP, method=method, seed=42)
opts.sample_split = int(os.environ.get("SPLIT", "1"))
for _ in range(reps):
    img, rays = g.render(cam, opts)
    print(f"kernel {g.last_kernel_ms()[0]:.1f} ms rays {rays} Msamples/s {W*H*SPP/g.last_kernel_ms()[0]/1e3:.1f} Mrays/s {rays/g.last_kernel_ms()[0]/1e3:.1f}", flush=True)
