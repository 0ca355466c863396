"""BASELINE config 2 (and one GPU's 1/8 share of it) for several sample_split values: the tail of a launch whose items are whole pixels
against the price of more, shorter items.  python tests/probes/gpu_split_sweep.py [scene]"""
import importlib, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
pkg = importlib.import_module("raytracing-rust_amd"); hb = importlib.import_module("raytracing-rust_amd.hip_backend"); abi = pkg.abi
import scenes
name = sys.argv[1] if len(sys.argv) > 1 else "rtweekend1"
ls = scenes.load_ssml(name); g = hb.HipScene(ls.scene); cam = hb.camera_new(**ls.camera_params)
for shards in (1, 8):
    for S in (1, 2, 4, 8, 16, 32, 64):
        o = abi.default_render_opts(1920, 1080, 1024); o.sample_split = S; o.shard_index, o.shard_count = 0, shards
        o.output_layout = abi.RT_LAYOUT_SHARD
        g.render(cam, o)
        best = 1e9
        for _ in range(2):
            g.render(cam, o); best = min(best, g.last_kernel_ms()[0])
        print(f"{name} shards {shards} split {S:2d}: kernel {best:7.2f} ms", flush=True)
