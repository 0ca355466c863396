"""Same-box A/B of alternate builds on BASELINE configs 2 and 3 at their full depth and the library's automatic split:
RT_HIP_LIB=... python tests/probes/gpu_r04_ab.py [scene ...]  ->  one line per scene: best and median kernel ms of 5 renders"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("raytracing-rust_amd"); hb = importlib.import_module("raytracing-rust_amd.hip_backend")
import scenes
for name in (sys.argv[1:] or ["rtweekend1", "overshadowed"]):
    ls = scenes.load_ssml(name)
    g = hb.HipScene(ls.scene); cam = hb.camera_new(**ls.camera_params)
    if "SCENE_LDS" in os.environ:  # 0: tiny scenes are read from global memory instead of being staged into LDS
        g.set_tuning(pkg.abi.RT_TUNE_SCENE_IN_LDS, int(os.environ["SCENE_LDS"]))
    if "FEATURE_SET" in os.environ:  # e.g. 0: the general spheres-only kernel on rtweekend1 instead of the two-sphere special case
        g.set_tuning(pkg.abi.RT_TUNE_FEATURE_SET, int(os.environ["FEATURE_SET"]))
    o = pkg.abi.default_render_opts(1920, 1080, int(os.environ.get("SPP", "1024")), method=1, seed=1)
    o.sample_split = int(os.environ.get("SPLIT", "0"))
    ms = []
    for _ in range(int(os.environ.get("REPS", "5"))):
        img, rays = g.render(cam, o)
        ms.append(g.last_kernel_ms()[0])
    ms.sort()
    print(f"{name}: best {ms[0]:.2f} ms  median {ms[len(ms)//2]:.2f} ms  split {g.last_launch_info()['sample_split']}  block {g.last_launch_info()['block_threads']} x {g.last_launch_info()['blocks_per_cu']}/CU  {g.last_launch_info()['kernel'][18:70]}  rays {rays}  checksum {float(img.astype('float64').sum()):.9e}", flush=True)
