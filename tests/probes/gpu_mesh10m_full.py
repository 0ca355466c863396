"""BASELINE configs[4] at its STATED depth on one rank: the synthetic 10 M-triangle scene, 4096 x 4096, ALL 4096 passes of
rank 0's 1/8 tile shard in one render (about 80 s of GPU), with the CPU oracle rendering six of that shard's tiles at the same
depth and the pixels compared bit for bit; and SURVEY 8(d)'s algorithmic bytes per sample for this workload, counted by the
oracle under reference traversal semantics on a sparse tile subset.  Writes gpurun_out/<tag>_mesh10m_full.json.
    python tests/probes/gpu_mesh10m_full.py <tag> [spp] [sample_split]     (the oracle renders its tiles at the same split;
                                                                            0 = the library's automatic split, resolved first)"""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import bench
pkg = importlib.import_module("raytracing-rust_amd"); hb = importlib.import_module("raytracing-rust_amd.hip_backend"); abi = pkg.abi
import oracle as O

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
split = int(sys.argv[3]) if len(sys.argv) > 3 else 1
name = "mesh10m"
w = bench.WORKLOADS[name]
t0 = time.time()
scene_desc, cam_params = bench.load_workload(pkg, name)
print(f"scene generated in {time.time() - t0:.1f} s", flush=True)
t0 = time.time()
g = hb.HipScene(scene_desc)
print(f"product: BVH + upload {time.time() - t0:.1f} s", flush=True)
cam = hb.camera_new(**cam_params)
opts = bench.workload_opts(abi, name, spp)          # shard 0 of 8, frame layout
opts.output_layout = abi.RT_LAYOUT_SHARD
if split == 0:  # the library's automatic choice (what bench.py runs), resolved here because the oracle takes explicit splits only
    split = g.auto_sample_split(opts)
    print(f"automatic sample_split for this launch: {split}", flush=True)
opts.sample_split = split
t0 = time.time()
shard, rays = g.render(cam, opts)
wall = time.time() - t0
kernel_ms = g.last_kernel_ms()[0]
n_samples = (w["width"] * w["height"] // 8) * spp
print(f"GPU: {spp} passes of the shard: kernel {kernel_ms / 1e3:.2f} s, wall {wall:.2f} s, {n_samples / kernel_ms / 1e3:.1f} Msamples/s, rays {rays}", flush=True)
order = hb.shard_pixel_order(opts)

t0 = time.time()
c = O.Scene(scene_desc)
print(f"oracle: BVH {time.time() - t0:.1f} s", flush=True)
ocam = O.camera_new(**cam_params)
# six of the shard's tiles: shard (0, 8 * m) owns every (8 m)-th tile of the frame, all of them tiles of shard (0, 8)
tiles_owned = (w["width"] // 8) * (w["height"] // 8) // 8
m = tiles_owned // 6
sub = bench.workload_opts(abi, name, spp)
sub.shard_index, sub.shard_count = 0, 8 * m
sub.sample_split = split
t0 = time.time()
ref, ref_rays = c.render(ocam, sub, n_threads=os.cpu_count())   # the oracle writes its shard's pixels into a whole frame
t_or = time.time() - t0
sub_pixels = np.array([int(p) for p in hb.shard_pixel_order(sub) if p != abi.NO_INDEX])
pos = {int(p): i for i, p in enumerate(order) if p != abi.NO_INDEX}
idx = np.array([pos[p] for p in sub_pixels])
ref_valid = ref.reshape(-1, 3)[sub_pixels]
got = shard.reshape(-1, 3)[idx]
same = bool(np.array_equal(got, ref_valid))
n_tiles = len(idx) // 64
print(f"oracle: {n_tiles} tiles x {spp} passes in {t_or:.1f} s; pixels identical: {same}; max |d| {float(np.abs(got - ref_valid).max()):.3g}", flush=True)

# SURVEY 8(d) algorithmic bytes, reference traversal semantics, on every 512th tile of the shard at 2 passes
cnt = bench.workload_opts(abi, name, 2)
cnt.shard_index, cnt.shard_count = 0, 8 * 512
t0 = time.time()
_, crays, counters = c.render(ocam, cnt, n_threads=os.cpu_count(), want_counters=True)
n_cnt = (w["width"] * w["height"] // (8 * 512)) * 2
per = {k: v / n_cnt for k, v in counters.items()}
alg = 32 * per["node_tests"] + 36 * per["triangle_tests"] + 16 * per["sphere_tests"] + 52 * per["closest_hits"] + 64 * per["sky_ops"] + 12.0 / w["full_spp"]
print(f"oracle counters on {n_cnt} samples in {time.time() - t0:.1f} s: {alg:.0f} algorithmic bytes per sample", flush=True)
out = {"workload": "mesh10m: 10 M triangles, 4096x4096, shard 0 of 8 (BASELINE configs[4], one rank)", "spp": spp, "sample_split": split, "samples": n_samples,
       "kernel_s": kernel_ms / 1e3, "wall_s_rt_render": wall, "Msamples_per_s": n_samples / kernel_ms / 1e3, "rays_shot": int(rays),
       "oracle_check": {"tiles": n_tiles, "passes": spp, "pixels_identical": same, "oracle_seconds": t_or, "threads": os.cpu_count()},
       "survey_8d_algorithmic": {"bytes_per_sample": alg, "per_sample": per, "counted_on": f"every 512th tile of the shard x 2 passes ({n_cnt} samples)",
                                 "requested_GBps": alg * n_samples / (kernel_ms / 1e3) / 1e9},
       "source_hash": bench.source_hash(), "launch": g.last_launch_info()}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"{tag}_mesh10m_full.json"), "w"), indent=1)
print(json.dumps(out)[:600])
assert same
