"""Generates the golden fixtures under tests/golden/ from the CPU oracle.

The reference ships no golden images or hit-record vectors and cannot be built or seeded here
(SURVEY section 0, 8(c)), so these vectors are outputs of the oracle itself: they freeze the oracle
(any later edit that changes a pixel fails tests/test_oracle_render.py) and give the GPU box a
reference it can check the HIP kernels against without recomputing.  Run from the repo root:
    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import oracle as O  # noqa: E402
import scenes  # noqa: E402

abi = scenes.abi
W, H = 64, 36


def cases():
    for name in ("rtweekend1", "overshadowed", "pyramid"):
        ls = scenes.load_ssml(name)
        yield name, ls.scene, ls.camera_params, 16
    yield "all_materials", scenes.all_materials(), scenes.ALL_MATERIALS_CAMERA, 8
    yield "mesh2000", scenes.random_triangle_mesh(2000, seed=42, extent=3.0, edge=0.5, emissive_every=100,
                                                  sampler_res=(20, 10)), scenes.MESH_CAMERA | {"origin": (0.0, -9.0, 0.0)}, 8
    yield "structured", scenes.structured_meshes(3, 32, (20, 10)), scenes.STRUCTURED_CAMERA, 8


def main():
    meta = {}
    for name, sc, cam_params, spp in cases():
        s = O.Scene(sc)
        cam = O.camera_new(**cam_params)
        for method, mname in ((abi.RT_METHOD_NAIVE, "naive"), (abi.RT_METHOD_MIS, "mis")):
            opts = abi.default_render_opts(W, H, spp, method=method, seed=1)
            img, rays = s.render(cam, opts, n_threads=4)
            np.save(os.path.join(HERE, f"{name}_{W}x{H}_s{spp}_{mname}.npy"), img)
            meta[f"{name}_{mname}"] = {"rays_shot": int(rays), "spp": spp, "seed": 1}
        rng = np.random.default_rng(1234)
        org = np.tile(np.asarray(cam_params["origin"], dtype=np.float32), (512, 1))
        org[256:] += rng.normal(size=(256, 3)).astype(np.float32)
        dirs = rng.normal(size=(512, 3)).astype(np.float32)
        hits = s.check_hit(org, dirs)
        np.save(os.path.join(HERE, f"{name}_hits.npy"), hits)
        np.save(os.path.join(HERE, f"{name}_rays.npy"), np.concatenate([org, dirs], axis=1))
    with open(os.path.join(HERE, "golden_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", len(meta), "images")


if __name__ == "__main__":
    main()
