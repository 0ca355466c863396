import sys, ctypes as C
sys.path.insert(0,"/root/repo"); sys.path.insert(0,"/root/repo/oracle"); sys.path.insert(0,"/root/repo/tests")
import oracle as O, scenes, numpy as np
lib=C.CDLL("/tmp/rt_sanitize/libhostbuild_asan.so")
abi=scenes.abi
def check(sc):
    s = O.Scene(sc); want = s.nodes(); desc = sc.desc()
    out = np.zeros(len(want)+16, dtype=want.dtype); n = C.c_uint64(); order = np.zeros(sc.n_primitives+1, dtype=np.uint64)
    assert lib.dbg_build(C.byref(desc), out.ctypes.data_as(C.c_void_p), C.c_uint64(len(out)), C.byref(n), order.ctypes.data_as(C.c_void_p), C.c_uint64(len(order))) == 0
    assert out[:n.value].tobytes() == want.tobytes()
for seed in range(60):
    check(scenes.random_everything(seed)[0])
for split in (0,1,2):
    for n in (1,2,3,5,64,257,5000):
        check(scenes.random_spheres(n, seed=n, split_type=split, emissive_every=7))
    check(scenes.random_triangle_mesh(70000, seed=3, extent=4.0, edge=0.4, split_type=split, sampler_res=(8,8)))
check(scenes.all_materials())
print("asan/ubsan run complete")
