"""Same-box A/B of alternate FULL builds over the small-scene (coarse-schedule) kernels beyond the two headline ones:
RT_HIP_LIB=... python tests/probes/gpu_r04_small_ab.py  ->  per case: kernel, best kernel ms of 3, checksum"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("raytracing-rust_amd"); hb = importlib.import_module("raytracing-rust_amd.hip_backend")
import scenes
abi = pkg.abi
MESHCAM = dict(scenes.MESH_CAMERA)
S500CAM = {"origin": (0.0, -30.0, 6.0), "lookat": (0.0, 0.0, 0.0), "vup": (0.0, 0.0, 1.0), "fov": 50.0, "aspect_ratio": 16.0 / 9.0, "aperture": 0.0, "focus_dist": 10.0}
def ssml(name):
    ls = scenes.load_ssml(name); return ls.scene, ls.camera_params
CASES = [
    ("rtweekend1 MIS", lambda: ssml("rtweekend1"), 1, 256, {}),
    ("rtweekend1 MIS general kernel", lambda: ssml("rtweekend1"), 1, 256, {"feature_set": 0}),
    ("rtweekend1 naive", lambda: ssml("rtweekend1"), 0, 256, {}),
    ("overshadowed MIS", lambda: ssml("overshadowed"), 1, 256, {}),
    ("overshadowed naive", lambda: ssml("overshadowed"), 0, 256, {}),
    ("spheres500 MIS", lambda: (scenes.random_spheres(500, seed=7, sampler_res=(100, 100)), S500CAM), 1, 64, {}),
    ("spheres500 naive", lambda: (scenes.random_spheres(500, seed=7, sampler_res=(100, 100)), S500CAM), 0, 64, {}),
    ("2000 triangles MIS", lambda: (scenes.random_triangle_mesh(2000, seed=42, extent=2.0, edge=0.3, emissive_every=50), MESHCAM), 1, 64, {}),
    ("all_materials MIS", lambda: (scenes.all_materials(), dict(scenes.ALL_MATERIALS_CAMERA)), 1, 64, {}),
    ("all_materials naive", lambda: (scenes.all_materials(), dict(scenes.ALL_MATERIALS_CAMERA)), 0, 64, {}),
]
for label, make, method, spp, env in CASES:
    desc, camp = make()
    g = hb.HipScene(desc); cam = hb.camera_new(**camp)
    if "feature_set" in env:
        g.set_tuning(abi.RT_TUNE_FEATURE_SET, env["feature_set"])
    o = abi.default_render_opts(1920, 1080, spp, method=method, seed=1)
    o.sample_split = 0
    ms = []
    for _ in range(4):
        img, rays = g.render(cam, o); ms.append(g.last_kernel_ms()[0])
    li = g.last_launch_info()
    print(f"{label}: best {min(ms[1:]):.2f} ms  {li['kernel']}  block {li['block_threads']} x {li['blocks_per_cu']}/CU  rays {rays}  checksum {float(img.astype('float64').sum()):.9e}", flush=True)
    del g
