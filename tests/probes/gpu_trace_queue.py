"""The experiment behind DESIGN.md section 8.1: a traversal-ONLY persistent kernel (diagnostic build) on incoherent rays,
at several occupancies, against the megakernel's node-step rate.
  RT_HIP_LIB=raytracing-rust_amd/librt_hip_stats.so python tests/probes/gpu_trace_queue.py [n_triangles] [n_rays]"""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("raytracing-rust_amd"); hb = importlib.import_module("raytracing-rust_amd.hip_backend")
import scenes
abi = pkg.abi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 8000000
ext = 10.0 if n <= 2000000 else 20.0
g = hb.HipScene(scenes.random_triangle_mesh(n, seed=42, extent=ext))
rng = np.random.default_rng(1)
org = rng.uniform(-ext, ext, (n_rays, 3)).astype(np.float32)  # bounce-like rays: origins inside the cloud, isotropic directions
d = rng.normal(size=(n_rays, 3)).astype(np.float32)
rays = np.ascontiguousarray(np.concatenate([org, d], axis=1))
ref = g.check_hit(org[:200000], d[:200000])
depth = g.wide_tree()[2]
print(f"{n} triangles, {n_rays} rays, stack depth {depth}, hit fraction {(ref['index'] != np.uint64(abi.NO_INDEX)).mean():.3f}")
for waves, cap in ((4, depth), (4, 20), (5, 20), (6, 20), (8, 20), (8, 16), (3, depth)):
    t = np.zeros(n_rays, dtype=np.float32); p = np.zeros(n_rays, dtype=np.uint32)
    ms = C.c_float(); steps = C.c_ulonglong()
    rc = hb.lib().rt_debug_trace_queue(g._h, rays.ctypes.data_as(C.c_void_p), C.c_uint64(n_rays), C.c_int(waves), C.c_uint32(cap),
                                       t.ctypes.data_as(C.POINTER(C.c_float)), p.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(ms), C.byref(steps))
    assert rc == 0, hb.lib().rt_last_error()
    hit = ref["index"] != np.uint64(abi.NO_INDEX)
    ok = np.array_equal(p[:200000][hit], ref["index"][hit].astype(np.uint32)) and np.array_equal(t[:200000][hit], ref["t"][hit]) and \
        (p[:200000][~hit] == 0xFFFFFFFF).all()
    print(f"  {waves} waves/SIMD, {cap:2d} stack entries in LDS: {ms.value:7.2f} ms  {n_rays/ms.value/1e3:7.1f} M rays/s  "
          f"{steps.value/n_rays:6.1f} node steps/ray  {steps.value/ms.value/1e6:6.1f} G node steps/s  same hits as rt_check_hit: {ok}", flush=True)
