"""Counts, with the oracle under REFERENCE traversal semantics, the work per sample of the bench
workloads and prints the algorithmic bytes per sample of SURVEY section 8(d):
    bytes(sample) = 32*N_node + 36*N_tri + 16*N_sph + 52*N_hits + 64*N_sky_ops + 12/spp
Run:  python tests/probes/count_algorithmic_bytes.py   (writes profiles/algorithmic_bytes.json)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import oracle as O  # noqa: E402
import scenes  # noqa: E402

abi = scenes.abi


def count(name, sc, cam_params, w, h, spp, method, full_spp):
    s = O.Scene(sc)
    cam = O.camera_new(**cam_params)
    img, rays, c = s.render(cam, abi.default_render_opts(w, h, spp, method=method), want_counters=True)
    n = w * h * spp
    per = {k: v / n for k, v in c.items()}
    b = 32 * per["node_tests"] + 36 * per["triangle_tests"] + 16 * per["sphere_tests"] + 52 * per["closest_hits"] + \
        64 * per["sky_ops"] + 12.0 / full_spp
    return {"workload": name, "counted_on": f"{w}x{h}x{spp}", "per_sample": per, "bytes_per_sample": b,
            "rays_shot_per_sample": rays / n}


def main():
    out = {}
    for name in ("rtweekend1", "overshadowed"):
        ls = scenes.load_ssml(name)
        for method, mname in ((abi.RT_METHOD_MIS, "mis"), (abi.RT_METHOD_NAIVE, "naive")):
            r = count(name, ls.scene, ls.camera_params, 1920, 1080, 4, method, 1024)
            out[f"{name}_1920x1080_{mname}"] = r
            print(name, mname, round(r["bytes_per_sample"], 2), {k: round(v, 3) for k, v in r["per_sample"].items()})
    # cfg4: the synthetic 1 M-triangle mesh (BASELINE configs[3]); one pass at quarter resolution is enough
    mesh = scenes.random_triangle_mesh(1000000, seed=42, extent=10.0)
    for method, mname in ((abi.RT_METHOD_MIS, "mis"), (abi.RT_METHOD_NAIVE, "naive")):
        r = count("mesh1m", mesh, scenes.MESH_CAMERA, 960, 540, 1, method, 256)
        out[f"mesh1m_1920x1080_{mname}"] = r
        print("mesh1m", mname, round(r["bytes_per_sample"], 2), {k: round(v, 3) for k, v in r["per_sample"].items()})
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "profiles", "algorithmic_bytes.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
