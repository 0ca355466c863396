"""RT_TUNE_EXCHANGE on/off under the fine schedule (random triangle meshes), same box, same library: kernel time and
bit-equality of the frames.  python tests/probes/gpu_exchange_mesh_ab.py [n_triangles] [spp]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("raytracing-rust_amd"); hb = importlib.import_module("raytracing-rust_amd.hip_backend")
import scenes
abi = pkg.abi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
g = hb.HipScene(scenes.random_triangle_mesh(n, seed=42, extent=10.0 if n <= 2000000 else 20.0)); cam = hb.camera_new(**scenes.MESH_CAMERA)
for method in (1, 0):
    frames = {}
    for xchg in (0, 1, 0, 1):
        g.set_tuning(abi.RT_TUNE_EXCHANGE, xchg)
        opts = abi.default_render_opts(1920, 1080, spp, method=method)
        img, rays = g.render(cam, opts)
        best = 1e9
        for _ in range(2):
            g.render(cam, opts); best = min(best, g.last_kernel_ms()[0])
        li = g.last_launch_info()
        same = "" if not frames else f" == first: {np.array_equal(frames[0][0], img) and frames[0][1] == rays}"
        frames.setdefault(0, (img, rays))
        print(f"tris={n} method={method} xchg={xchg} {spp}spp: kernel {best:.2f} ms {1920*1080*spp/best/1e3:.1f} Msamples/s  block {li['block_threads']} lds {li['lds_bytes']} blocks/cu {li['blocks_per_cu']}{same}", flush=True)
