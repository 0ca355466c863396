"""What a register-lean traversal kernel achieves on the big meshes: rt_check_hit (check_hit_kernel: one ray per lane,
the wide walk, 76 VGPRs = 6 waves/SIMD, nothing else in flight) on incoherent rays.  Run under
  rocprofv3 --kernel-trace --stats -- python3 tests/probes/gpu_trace_rate.py [n_triangles] [n_rays]
and read check_hit_kernel's time; this prints the end-to-end call time (PCIe copies included) and the hit fraction."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("raytracing-rust_amd"); hb = importlib.import_module("raytracing-rust_amd.hip_backend")
import scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 8000000
ext = 10.0 if n <= 2000000 else 20.0
g = hb.HipScene(scenes.random_triangle_mesh(n, seed=42, extent=ext))
rng = np.random.default_rng(1)
# "bounce-like" rays: origins inside the cloud, isotropic directions (what secondary rays of the 1080p frame look like)
org = rng.uniform(-ext, ext, (n_rays, 3)).astype(np.float32)
d = rng.normal(size=(n_rays, 3)).astype(np.float32)
for rep in range(3):
    t0 = time.time()
    h = g.check_hit(org, d)
    dt = time.time() - t0
    print(f"{n} triangles, {n_rays} incoherent rays: call {dt*1e3:.1f} ms (with PCIe), hit fraction {(h['index'] != np.uint64(pkg.abi.NO_INDEX)).mean():.3f}", flush=True)
