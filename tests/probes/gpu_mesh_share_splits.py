import importlib, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
pkg = importlib.import_module("raytracing-rust_amd"); hb = importlib.import_module("raytracing-rust_amd.hip_backend"); abi = pkg.abi
import scenes
g = hb.HipScene(scenes.random_triangle_mesh(1000000, seed=42)); cam = hb.camera_new(**scenes.MESH_CAMERA)
for shards, S in ((8, 16), (8, 32), (8, 64), (4, 32), (4, 64), (2, 32)):
    o = abi.default_render_opts(1920, 1080, 256, method=1, seed=42)
    o.sample_split = S
    o.shard_index, o.shard_count, o.output_layout = 0, shards, abi.RT_LAYOUT_SHARD
    g.render(cam, o); g.render(cam, o)
    print(f"shards {shards}: split {S:2d}  kernel {g.last_kernel_ms()[0]:8.1f} ms", flush=True)
