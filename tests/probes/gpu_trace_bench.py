"""Pure traversal throughput: rt_check_hit on primary camera rays of the synthetic mesh.
python tests/probes/gpu_trace_bench.py n_tris W H [shuffle]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("raytracing-rust_amd"); hb = importlib.import_module("raytracing-rust_amd.hip_backend")
import scenes
abi = pkg.abi
n, W, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
shuffle = len(sys.argv) > 4
g = hb.HipScene(scenes.random_triangle_mesh(n, seed=42))
cam = hb.camera_new(**scenes.MESH_CAMERA)
o = np.float32(list(cam.origin)); ll = np.float32(list(cam.lower_left)); hz = np.float32(list(cam.horizontal)); vt = np.float32(list(cam.vertical))
ys, xs = np.mgrid[0:H, 0:W]
# 8x8 tile order like the render kernel
tiles = (ys // 8) * (W // 8) + (xs // 8); inner = (ys % 8) * 8 + (xs % 8)
order = np.argsort((tiles * 64 + inner).reshape(-1), kind="stable")
u = ((xs + 0.5) / (W - 1)).astype(np.float32).reshape(-1)[order]; v = (1 - (ys + 0.5) / (H - 1)).astype(np.float32).reshape(-1)[order]
d = ll[None] + hz[None] * u[:, None] + vt[None] * v[:, None] - o[None]
org = np.tile(o, (len(d), 1))
if shuffle:
    p = np.random.default_rng(0).permutation(len(d)); d = d[p]
for mode in (1, 0):
    g.set_traversal(mode)
    t = time.time(); h = g.check_hit(org, d); dt = time.time() - t
    print("traversal mode", mode, "rays", len(d), "hits", int((h["index"] != np.uint64(abi.NO_INDEX)).sum()), "wall", round(dt, 3), flush=True)
