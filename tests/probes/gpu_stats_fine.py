import ctypes as C, importlib, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
pkg = importlib.import_module("raytracing-rust_amd"); hb = importlib.import_module("raytracing-rust_amd.hip_backend"); abi = pkg.abi
import scenes
g = hb.HipScene(scenes.random_triangle_mesh(int(sys.argv[1]) if len(sys.argv) > 1 else 1000000, seed=42)); cam = hb.camera_new(**scenes.MESH_CAMERA)
names = ["GEN", "NODE(x8)", "LEAF", "SHADE", "LIGHT", "SCATTER"]
for method in (0, 1):
    o = abi.default_render_opts(1920, 1080, 8, method=method)
    out = (C.c_ulonglong * 64)()
    hb.lib().rt_debug_stats(out, 1)
    g.render(cam, o)
    hb.lib().rt_debug_stats(out, 1)
    tot = sum(out[5 + k] for k in range(6))
    print("method", method, "kernel ms", round(g.last_kernel_ms()[0], 1))
    for k in range(6):
        it, act = out[5 + k], out[11 + k]
        if it: print(f"   {names[k]:9s} iters {it:10d} ({100*it/tot:5.1f}%)  avg lanes {act/it:5.1f}/64")
    n = 1920 * 1080 * 8
    print(f"   lane node steps {out[20]} ({out[20]/n:.1f}/sample)  primitive tests {out[21]} ({out[21]/n:.2f}/sample)  max stack {out[22]}")
    print(f"   dead node visits {out[23]/max(1,out[20]):.3f} of all; with a pruned hit child {out[24]/max(1,out[20]):.3f}; all hit children pruned {out[25]/max(1,out[20]):.3f}")
    clk = [out[50 + k] for k in range(7)]
    tot_c = max(1, sum(clk))
    print("   wave wall-clock share: " + "  ".join(f"{nm} {100*v/tot_c:.1f}%" for nm, v in zip(["GEN", "NODE", "LEAF", "SHADE", "LIGHT", "SCATTER", "vote+claim"], clk)))
