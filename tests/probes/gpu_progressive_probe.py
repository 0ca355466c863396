"""Progressive rendering cost: rt_sample_image (double-buffered, copy stream) vs a loop of blocking rt_render calls.
python tests/probes/gpu_progressive_probe.py [W H spp]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("raytracing-rust_amd"); hb = importlib.import_module("raytracing-rust_amd.hip_backend")
import numpy as np
import scenes
abi = pkg.abi
W, H, SPP = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (1920, 1080, 64)
ls = scenes.load_ssml("rtweekend1"); g = hb.HipScene(ls.scene); cam = hb.camera_new(**ls.camera_params)
opts = abi.default_render_opts(W, H, SPP, seed=1)
g.render(cam, abi.default_render_opts(W, H, 1))
for batch in (1, 4, 16):
    image = np.zeros(W * H * 3, dtype=np.float32)
    hb.RandomSampler(batch=batch).sample_image(opts, cam, g, (image, hb.running_mean_update))  # warm-up (allocations)
    image[:] = 0
    t0 = time.perf_counter()
    hb.RandomSampler(batch=batch).sample_image(opts, cam, g, (image, hb.running_mean_update))
    t_new = time.perf_counter() - t0
    image2 = np.zeros(W * H * 3, dtype=np.float32)
    t0 = time.perf_counter()
    done = 0
    while done < SPP:
        o = abi.default_render_opts(W, H, batch, seed=1); o.sample_begin = done
        img, rays = g.render(cam, o)
        done += batch
        p = hb.SamplerProgress(0); p.samples_completed = batch; p.current_image = img.reshape(-1)
        hb.running_mean_update(image2, p, done)
    t_old = time.perf_counter() - t0
    t0 = time.perf_counter()
    hb.RandomSampler(batch=batch).sample_image(opts, cam, g, (None, lambda d, p, i: False))
    t_nocb = time.perf_counter() - t0
    print(f"batch {batch:3d}: rt_sample_image {t_new*1e3:8.1f} ms (no-op callback {t_nocb*1e3:7.1f} ms)   blocking rt_render loop {t_old*1e3:8.1f} ms   "
          f"max |diff| {np.abs(image - image2).max():.2e}", flush=True)
