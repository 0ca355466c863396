"""Same-box A/B of alternate FULL builds on bench.py's two mesh workloads (its scene, camera, options, automatic split):
RT_HIP_LIB=... python tests/probes/gpu_r04_mesh_ab.py [mesh1m|mesh10m ...] -> best / median kernel ms and the image checksum"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module("raytracing-rust_amd"); hb = importlib.import_module("raytracing-rust_amd.hip_backend")
for name in (sys.argv[1:] or ["mesh1m", "mesh10m"]):
    desc, camp = bench.load_workload(pkg, name)
    g = hb.HipScene(desc); cam = hb.camera_new(**camp)
    o = bench.workload_opts(pkg.abi, name, spp=int(os.environ["SPP"]) if "SPP" in os.environ else None)
    o.sample_split = int(os.environ.get("SPLIT", "0"))
    ms = []
    for _ in range(int(os.environ.get("REPS", "3"))):
        img, rays = g.render(cam, o)
        ms.append(g.last_kernel_ms()[0])
    ms.sort()
    print(f"{name}: best {ms[0]:.1f} ms  median {ms[len(ms)//2]:.1f} ms  split {g.last_launch_info()['sample_split']}  rays {rays}  checksum {float(img.astype('float64').sum()):.9e}", flush=True)
    del g
