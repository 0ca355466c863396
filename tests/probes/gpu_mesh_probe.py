"""Large synthetic mesh probe (BASELINE configs 3/4 shape) on the GPU box:
python tests/probes/gpu_mesh_probe.py <n_triangles> <W> <H> <spp> [extent]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("raytracing-rust_amd"); hb = importlib.import_module("raytracing-rust_amd.hip_backend")
import oracle as O, scenes
abi = pkg.abi
n = int(sys.argv[1]); W, H, SPP = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
extent = float(sys.argv[5]) if len(sys.argv) > 5 else 10.0
t = time.time(); sc = scenes.random_triangle_mesh(n, seed=42, extent=extent); print("generate", round(time.time() - t, 2), "s", flush=True)
t = time.time(); g = hb.HipScene(sc); print("rt_scene_create (BVH build + upload)", round(time.time() - t, 2), "s", g.counts(), flush=True)
cam = hb.camera_new(**scenes.MESH_CAMERA)
for method, mname in ((1, "mis"), (0, "naive")):
    opts = abi.default_render_opts(W, H, SPP, method=method, seed=42)
    t = time.time(); img, rays = g.render(cam, opts); dt = time.time() - t
    ms, _ = g.last_kernel_ms()
    print(f"{mname}: wall {dt:.2f}s kernel {ms:.1f} ms  {W*H*SPP/ms/1e3:.1f} Msamples/s rays {rays} mean {img.mean(axis=(0,1))} finite {np.isfinite(img).all()}", flush=True)
    if method == 1:
        keep = img
# oracle check on a sparse shard of tiles (reference-semantic exhaustive traversal on the CPU)
t = time.time(); c = O.Scene(sc); print("oracle scene", round(time.time() - t, 2), "s", flush=True)
sub = abi.default_render_opts(W, H, SPP, method=1, seed=42)
sub.shard_index, sub.shard_count = 5, max(2, (W // 8) * (H // 8) // 40)
t = time.time(); ref, _ = c.render(O.camera_new(**scenes.MESH_CAMERA), sub); print("oracle shard render", round(time.time() - t, 2), "s", flush=True)
order = hb.shard_pixel_order(sub); order = order[order != np.uint64(abi.NO_INDEX)].astype(np.int64)
a = keep.reshape(-1, 3)[order]; b = ref.reshape(-1, 3)[order]
print("checked pixels", len(order), "bit-exact", np.array_equal(a, b), "max|d|", float(np.abs(a - b).max()))
