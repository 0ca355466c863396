"""Schedule statistics of the fine (big-tree) kernels from the -DRT_STATS build:
RT_HIP_LIB=raytracing-rust_amd/librt_hip_stats.so python tests/probes/gpu_stats_fine.py [n_triangles] [extent]"""
import ctypes as C, importlib, os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
pkg = importlib.import_module("raytracing-rust_amd"); hb = importlib.import_module("raytracing-rust_amd.hip_backend"); abi = pkg.abi
import scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
g = hb.HipScene(scenes.random_triangle_mesh(n, seed=42, extent=float(sys.argv[2]) if len(sys.argv) > 2 else (10.0 if n <= 2000000 else 20.0))); cam = hb.camera_new(**scenes.MESH_CAMERA)
g.set_tuning(abi.RT_TUNE_EXCHANGE, int(os.environ.get("RT_EXCHANGE", "0")))
names = ["GEN", "NODE(x16)", "LEAF", "SHADE", "LIGHT", "SCATTER", "NARROW"]
PH = 7
for method in (0, 1):
    o = abi.default_render_opts(1920, 1080, 8, method=method)
    out = (C.c_ulonglong * 64)()
    hb.lib().rt_debug_stats(out, 1)
    g.render(cam, o)
    hb.lib().rt_debug_stats(out, 1)
    tot = sum(out[5 + k] for k in range(PH))
    print("method", method, "kernel ms", round(g.last_kernel_ms()[0], 1), g.last_launch_info()["kernel"], "stack cap/lds", g.last_launch_info()["lds_bytes"])
    for k in range(PH):
        it, act = out[5 + k], out[5 + PH + k]
        if it: print(f"   {names[k]:9s} iters {it:10d} ({100*it/tot:5.1f}%)  avg lanes {act/it:5.1f}/64")
    ns = 1920 * 1080 * 8
    print(f"   lane node steps {out[20]} ({out[20]/ns:.1f}/sample)  primitive tests {out[21]} ({out[21]/ns:.2f}/sample)  max stack {out[22]}")
    clk = [out[50 + k] for k in range(PH + 1)]
    tot_c = max(1, sum(clk))
    print("   wave wall-clock share: " + "  ".join(f"{nm} {100*v/tot_c:.1f}%" for nm, v in zip(names + ["vote+claim"], clk)))
