"""One GPU's share of an N-way sharded frame for several sample_split values (rehearses multi-GPU strong scaling)."""
import importlib, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
pkg = importlib.import_module("raytracing-rust_amd"); hb = importlib.import_module("raytracing-rust_amd.hip_backend"); abi = pkg.abi
import scenes
ls = scenes.load_ssml("rtweekend1"); g = hb.HipScene(ls.scene); cam = hb.camera_new(**ls.camera_params)
base = None
for shards in (1, 2, 4, 8):
    for S in (1, 4, 16, 32, 64):
        if shards == 1 and S > 4: continue
        if shards == 2 and S > 16: continue
        worst = 0.0
        for idx in sorted({0, shards - 1}):
            o = abi.default_render_opts(1920, 1080, 1024); o.sample_split = S; o.shard_index, o.shard_count = idx, shards
            o.output_layout = abi.RT_LAYOUT_SHARD
            g.render(cam, o); g.render(cam, o)
            worst = max(worst, g.last_kernel_ms()[0])
        if base is None: base = worst
        print(f"shards {shards} split {S:2d}: kernel {worst:7.2f} ms  ideal {base/shards:6.2f}  efficiency {base/shards/worst:5.2f}", flush=True)
