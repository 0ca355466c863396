#!/usr/bin/env python3
"""bench.py -- Msamples/s of the HIP path tracer on BASELINE.json's headline configuration.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

One "step" = one complete render of the workload's frame, scene already resident in HBM.  The default
workload is BASELINE.json configs[1]: scenes/rtweekend1.ssml at 1920x1080, 1024 spp, MIS, max_depth 50 =
2.123 G samples per step.  Other workloads: `overshadowed` = configs[2], `mesh1m` =
configs[3] on one GPU, `mesh10m` = one rank's 1/8 shard of configs[4] (the only scene that exceeds the
Infinity Cache), `cfg1` = configs[0] (CPU oracle only, no GPU).

Multi-GPU (--gpus N > 1): one process per GPU.  Launched bare (`python bench.py --gpus N`), this process
is only a launcher: it starts N rank children (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
environment) BEFORE anything here touches the GPU, waits for them and exits non-zero if any of them
failed.  Launched by `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the ranks already
exist and each process is one of them.  The image's 8x8 tiles are interleaved over the ranks
(samplers/random_sampler.rs:45-52 partitions by pixel chunk the same way: no cross-chunk dependency), every
rank renders its tiles, ONE RCCL gather per step assembles the frame on rank 0 (strong scaling: the frame is
fixed).  Rank 0 prints ONE JSON line.

`roofline` names the bound that actually limits the dominant kernel:
  * rtweekend1 / overshadowed: the scene lives in SGPRs / LDS, HBM sees 12 bytes per pixel per frame, so the
    bound is VALU issue: VALU wave-instructions per sample (PMC, profiles/kernel_counters.json, measured on
    the kernel build named there) x samples / kernel time against CUs x 4 SIMDs x clock / 2.
  * mesh1m: dependent random 64-byte node fetches: L1->L2 requests per second against the measured
    random-fetch ceiling of the memory system (profiles/r01d_random_fetch_microbench.txt).
  * mesh10m: HBM: bytes the pruned walk requests per sample (counted live by the -DRT_STATS diagnostic
    build on an untimed pass) x samples / kernel time against 8 TB/s, with the fabric bytes the PMC
    counters saw as `traffic`.
SURVEY 8(d)'s algorithmic bytes (reference traversal semantics, counted by the oracle) stay as a secondary
field; they are requested bytes, not HBM bytes, and never a roofline fraction.
`cpu_baseline` times the CPU oracle (oracle/, a restatement of the reference's algorithm -- the Rust binary
cannot be built in this pipeline) on this box's host cores on a bounded sample of the same workload.  The
oracle is used for nothing else here.
"""
import argparse
import hashlib
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6 290 GB/s measured achievable
HBM_ACHIEVABLE_GBS = 6290.0
CLOCK_HZ = 2.4e9               # max engine clock (MI355X_MICROARCH.md chip-level parameters)
SIMDS = 256 * 4
VALU_ISSUE_PEAK = SIMDS * CLOCK_HZ / 2.0  # one wave64 VALU instruction per 2 cycles per SIMD-32
# What a SIMD really spends per wave-instruction with four waves on it, MEASURED with in-kernel cycle counters (s_memtime /
# s_memrealtime around the loop: tests/probes/microbench/valu_issue.hip -> profiles/r04_valu_issue.txt; the chip held 2.2 - 2.39 GHz
# there, not the 1.75 GHz round 3's wall-clock reading would have needed to be consistent with 2 cycles):
VALU_PLAIN_CYCLES_MEASURED = 2.5     # v_fma / v_xor / v_mov / v_cndmask (2.27 with eight waves; one wave alone 5.1)
VALU_HALF_RATE_CYCLES_MEASURED = 4.25  # v_mul_lo/hi_u32, v_mad_u64_u32, v_pk_*_f32
VALU_TRANS_CYCLES_MEASURED = 8.25    # v_rcp_f32, v_sqrt_f32
SALU_CYCLES_MEASURED = 2.15          # a scalar instruction of the same wave, interleaved 1:1 with VALU: 4.65 per pair, i.e. NOT free
# ... and with other wave counts (wall-clock x the clock the waves read; profiles/r04_valu_issue.txt for 8, r04z_valu_issue_w5_w6.txt
# for 5 and 6 -- the config-2 kernel runs six waves per SIMD since round 4): (plain, half-rate, transcendental, scalar)
ISSUE_CYCLES_BY_WAVES = {4: (2.5, 4.25, 8.25, 2.15), 5: (2.4, 4.2, 8.1, 2.05), 6: (2.36, 4.03, 7.85, 1.96), 8: (2.27, 4.1, 8.1, 2.0)}
# dependent random record fetches, every lane its own chain, in the walk's real access shape -- three 16-byte pieces of a 64-byte
# record, four waves per SIMD (tests/probes/microbench/random_fetch.hip -> profiles/r04h_random_fetch_microbench.txt): 237 G records/s
# when the set is L2-resident per XCD (2 MB), 57 G/s from 128 MB up AND at 1 GB, beyond the Infinity Cache: past the L2s every record
# costs one 128-byte fabric line, 57 G lines/s = 7.3 TB/s.  (Round 1's 115 / 57 were measured at 8 MB -- a blend -- and 134 MB.)
L2_FETCH_CEILING_RESIDENT = 237.0e9
L2_FETCH_CEILING_BEYOND = 57.0e9
RANDOM_LINE_CEILING_GBS = L2_FETCH_CEILING_BEYOND * 128.0 / 1e9  # 7 296 GB/s of 128-byte lines over the fabric

# bytes: SURVEY 8(d) algorithmic bytes per sample under REFERENCE traversal semantics, counted by
# tests/probes/count_algorithmic_bytes.py (32 B/node test, 16 B/sphere test, 36 B/triangle test, 52 B/closest
# hit, 64 B/sky op, 12 B/pixel) -> profiles/algorithmic_bytes.json
WORKLOADS = {
    "rtweekend1": {"width": 1920, "height": 1080, "spp": 1024, "seed": 1, "bytes": 446.77, "bound": "valu_issue",
                   "label": "scenes/rtweekend1.ssml", "config": "BASELINE configs[1]"},
    "overshadowed": {"width": 1920, "height": 1080, "spp": 1024, "seed": 1, "bytes": 488.42, "bound": "valu_issue",
                     "label": "scenes/overshadowed.ssml", "config": "BASELINE configs[2]"},
    # NOT a BASELINE config: ~500 random spheres under a sampled Lerp sky (the RTIOW cover shape) -- the general spheres-only kernels
    # (pruned walk over the wide tree, coarse schedule), i.e. what the small-scene path delivers beyond rtweekend1's two-sphere special case
    "spheres500": {"width": 1920, "height": 1080, "spp": 256, "seed": 7, "bytes": None, "bound": "valu_issue",
                   "label": "500 random spheres (tests/scenes.py random_spheres(500, seed=7), sky sampled at 100x100)",
                   "config": "not a BASELINE config: the general small-scene path", "spheres": 500,
                   "camera": {"origin": (0.0, -30.0, 6.0), "lookat": (0.0, 0.0, 0.0), "vup": (0.0, 0.0, 1.0), "fov": 50.0, "aspect_ratio": 16.0 / 9.0,
                              "aperture": 0.0, "focus_dist": 10.0}},
    "mesh1m": {"width": 1920, "height": 1080, "spp": 256, "seed": 42, "bytes": 38990.98, "bound": "l2_request_rate",
               "label": "synthetic 1M random-triangle mesh", "config": "BASELINE configs[3] on one GPU",
               "triangles": 1000000, "extent": 10.0},
    # configs[4] is 4096 spp on 8 GPUs; one GPU's share of it is every 8th tile.  The cost of a sample does not
    # depend on how many follow it, so the timed frame carries 64 of the 4096 passes (about 7 s per step).
    "mesh10m": {"width": 4096, "height": 4096, "spp": 64, "full_spp": 4096, "seed": 42, "bytes": 206415.04, "bound": "hbm",
                "label": "synthetic 10M random-triangle mesh", "config": "one rank's 1/8 tile shard of BASELINE configs[4]",
                "triangles": 10000000, "extent": 20.0, "shard": (0, 8), "aspect": 1.0},
}


def source_hash():
    """sha256 over the kernel sources: ties PMC-derived counters to the build they were measured on."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "raytracing-rust_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h", ".cpp")) or name == "Makefile":  # the Makefile carries the compiler flags
            h.update(open(os.path.join(csrc, name), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "rt_detmath.h"), "rb").read())
    return h.hexdigest()[:16]


def useful_valu_share(kernel):
    """1 - (v_mov / v_cndmask / v_readlane / v_writelane share of the kernel's VALU instructions), from the committed static
    census of the build's assembly (profiles/valu_census.json, written by tests/probes/isa_census.py --write and keyed by the
    hash of the kernel sources): the roofline fraction counts every issued VALU instruction, this says how many of them
    compute.  Static (per instruction of the binary, cold paths included), not a dynamic count."""
    p = os.path.join(ROOT, "profiles", "valu_census.json")
    if not os.path.exists(p):
        return {"useful_valu_share": None}
    census = json.load(open(p))
    k = census.get("kernels", {}).get(kernel)
    if not k:
        return {"useful_valu_share": None}
    return {"useful_valu_share": k["useful_valu_share"], "valu_census": {x: k[x] for x in ("valu", "moves", "v_mov", "v_cndmask", "lane_spill", "div", "salu")},
            "valu_census_issue_cycles_static_mix": k.get("valu_issue_cycles_static_mix"),
            "valu_census_measured_on_this_build": census.get("source_hash") == source_hash()}


def load_workload(pkg, name):
    """(scene description, camera parameters) of a workload"""
    w = WORKLOADS[name]
    if "triangles" in w:  # the synthetic random-triangle meshes (generator: tests/scenes.py)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import scenes
        cam = dict(scenes.MESH_CAMERA)
        if "aspect" in w:
            cam["aspect_ratio"] = w["aspect"]
        return scenes.random_triangle_mesh(w["triangles"], seed=w["seed"], extent=w["extent"]), cam
    if "spheres" in w:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import scenes
        return scenes.random_spheres(w["spheres"], seed=w["seed"], sampler_res=(100, 100)), dict(w["camera"])
    ls = pkg.ssml.load_file(os.path.join(ROOT, "tests", "golden", "scenes", name + ".ssml"))
    return ls.scene, ls.camera_params


def workload_opts(abi, name, spp=None):
    w = WORKLOADS[name]
    o = abi.default_render_opts(w["width"], w["height"], spp or w["spp"], method=abi.RT_METHOD_MIS, seed=w["seed"])
    if "shard" in w:
        o.shard_index, o.shard_count = w["shard"]
    return o


def cpu_baseline(pkg, name, scene_desc, camera_params, target_seconds=20.0):
    """Time the CPU oracle on a bounded sample: same scene, resolution, seed, method and shard, fewer spp."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O  # the ONLY use of oracle/ in this file: the reported CPU baseline
    abi = pkg.abi
    w = WORKLOADS[name]
    cores = os.cpu_count() or 1
    s = O.Scene(scene_desc)
    cam = O.camera_new(**camera_params)
    base = workload_opts(abi, name, 1)
    owned = (w["width"] * w["height"]) // base.shard_count
    # calibration, untimed: every 16th tile of the shard at 2 spp (big meshes need seconds per full pass on the CPU)
    strip = workload_opts(abi, name, 2)
    strip.shard_index, strip.shard_count = base.shard_index, base.shard_count * 16
    t0 = time.time()
    s.render(cam, strip, n_threads=cores)
    per_spp = max((time.time() - t0) * 16.0 / 2.0, 1e-3)
    if per_spp < target_seconds / 4.0:  # cheap enough: one whole pass instead (every 16th tile misjudged the 500-sphere scene by 4 x)
        t0 = time.time()
        s.render(cam, base, n_threads=cores)
        per_spp = max(time.time() - t0, 1e-3)
    spp = int(max(1, min(512, w["spp"], target_seconds / per_spp)))
    timed = workload_opts(abi, name, spp)
    thin = 1
    while per_spp / thin > 1.5 * target_seconds and thin < 64:  # one pass of the whole shard is already too long: every thin-th tile of it
        thin *= 2
    timed.shard_index, timed.shard_count = base.shard_index, base.shard_count * thin
    t0 = time.time()
    s.render(cam, timed, n_threads=cores)
    dt = time.time() - t0
    return {"value": (owned // thin) * spp / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"{name} {w['width']}x{w['height']}" + (f" shard {base.shard_index}/{base.shard_count}" if base.shard_count > 1 else "") +
                      (f", every {thin}th tile of it" if thin > 1 else "") +
                      f", {spp} of {w.get('full_spp', w['spp'])} spp, MIS, max_depth 50, {dt:.1f} s of CPU work"}


def run_cfg1(args):
    """BASELINE configs[0]: rtweekend1 at 400x225x64 on the CPU path (the oracle; no GPU), with max_depth 8 as
    BASELINE says and 50 as the reference has it (integrators/mod.rs:7) -- both reported."""
    pkg = importlib.import_module("raytracing-rust_amd")
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    abi = pkg.abi
    ls = pkg.ssml.load_file(os.path.join(ROOT, "tests", "golden", "scenes", "rtweekend1.ssml"))
    s = O.Scene(ls.scene)
    cam = O.camera_new(**ls.camera_params)
    cores = os.cpu_count() or 1
    W, H, SPP = 400, 225, 64
    chunks = (W * H + 9999) // 10000  # random_sampler.rs:31-32,45: par_chunks_mut(10 000 px): at most 9 busy threads
    res = {}
    for depth in (8, 50):
        o = abi.default_render_opts(W, H, SPP, method=abi.RT_METHOD_MIS, seed=1)
        o.max_depth = depth
        for _ in range(args.warmup):
            s.render(cam, o, n_threads=cores)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            _, rays = s.render(cam, o, n_threads=cores)
        dt = (time.perf_counter() - t0) / args.steps
        res[depth] = {"Msamples_per_s": W * H * SPP / dt / 1e6, "ms_per_step": dt * 1e3, "rays_shot": int(rays)}
    out = {"metric": "Msamples/s on rtweekend1.ssml 400x225x64spp (CPU oracle)", "value": res[8]["Msamples_per_s"], "unit": "Msamples/s",
           "n_gpus": 0, "steps": args.steps, "warmup": args.warmup, "ms_per_step": res[8]["ms_per_step"], "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "scenes/rtweekend1.ssml 400x225 64spp MIS seed=1 on the CPU oracle (BASELINE configs[0]); value = max_depth 8",
                      "threads": cores, "chunks_per_pass": chunks,
                      "note": "per-pass barrier over 10 000-pixel chunks as the reference: 9 chunks, so at most 9 threads are busy",
                      "max_depth_8": res[8], "max_depth_50": res[50]},
           "cpu_baseline": {"value": res[50]["Msamples_per_s"], "unit": "Msamples/s", "cores": min(cores, chunks), "kind": "port",
                            "sample": "the whole of configs[0] at the reference's max_depth 50"}}
    print(json.dumps(out), flush=True)


def walk_stats_child(name):
    """Runs in a child process with RT_HIP_LIB = the -DRT_STATS diagnostic build: one untimed pass of the
    workload's frame at 2 spp; prints what the pruned walk did per sample."""
    import ctypes as C
    pkg = importlib.import_module("raytracing-rust_amd")
    hb = importlib.import_module("raytracing-rust_amd.hip_backend")
    scene_desc, camera_params = load_workload(pkg, name)
    g = hb.HipScene(scene_desc, device=0)
    cam = hb.camera_new(**camera_params)
    spp = 2 if "triangles" in WORKLOADS[name] else 64
    o = workload_opts(pkg.abi, name, spp)
    o.output_layout = pkg.abi.RT_LAYOUT_SHARD
    out = (C.c_ulonglong * 64)()
    hb.lib().rt_debug_stats(out, 1)
    g.render(cam, o)
    hb.lib().rt_debug_stats(out, 1)
    n = (o.width * o.height // o.shard_count) * spp
    res = {"counted_on": f"{o.width}x{o.height} shard {o.shard_index}/{o.shard_count} x {spp} spp"}
    if out[20]:  # fine schedule (big trees): what the pruned walk did
        res.update({"node_steps_per_sample": out[20] / n, "primitive_tests_per_sample": out[21] / n, "max_stack": int(out[22])})
    if out[0]:   # coarse schedule: lanes taking part in the two voted super-phases (live paths per wave iteration)
        res.update({"primary_phase_lanes": out[1] / out[0], "bounce_phase_lanes": out[3] / max(1, out[2]),
                    "primary_iterations_per_64_samples": out[0] * 64 / n, "bounce_iterations_per_64_samples": out[2] * 64 / n})
    print(json.dumps(res), flush=True)


def walk_stats(name):
    """(statistics, error): the untimed pass of the -DRT_STATS diagnostic build in a child process.  A failure there (library
    missing, child crash, timeout, unparsable output) is REPORTED -- in the JSON line and on stderr -- never retried, never silent."""
    lib = os.path.join(ROOT, "raytracing-rust_amd", "librt_hip_stats.so")
    if not os.path.exists(lib):
        return None, {"error": "librt_hip_stats.so not built (make -C raytracing-rust_amd/csrc)"}
    env = dict(os.environ, RT_HIP_LIB=lib)
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--walk-stats-child", name], env=env, capture_output=True,
                           text=True, timeout=900)
        if r.returncode != 0:
            return None, {"error": "walk-statistics child failed", "returncode": r.returncode, "stderr_tail": r.stderr[-800:]}
        return json.loads(r.stdout.strip().splitlines()[-1]), None
    except Exception as e:  # timeout, bad JSON
        return None, {"error": f"{type(e).__name__}: {e}"}


def spawn_ranks(n):
    """`python bench.py --gpus N` with no launcher around it: start the N rank processes from here.  This
    process imports neither torch nor the HIP library and never touches a GPU; it waits for the ranks and
    exits non-zero if any of them failed."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    deadline = None
    while any(p.poll() is None for p in procs):
        for p in procs:
            if p.poll() not in (None, 0) and deadline is None:
                deadline = time.time() + 30.0  # one rank died: the others will hang in a collective; give them 30 s
        if deadline is not None and time.time() > deadline:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        time.sleep(0.2)
    for p in procs:
        rc = rc or p.returncode
    sys.exit(1 if rc else 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-walk-stats", action="store_true")
    ap.add_argument("--workload", choices=sorted(WORKLOADS) + ["cfg1"], default="rtweekend1")
    ap.add_argument("--abi-devices", default=None,
                    help="comma-separated device list, e.g. 0,1,2,3: ONE process renders through rt_scene_create_multi (the C ABI's own "
                         "multi-GPU path: tiles t %% n, in-process gather into the first device) instead of one rank per GPU")
    ap.add_argument("--walk-stats-child", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.walk_stats_child:
        return walk_stats_child(args.walk_stats_child)
    name = args.workload
    if name == "cfg1":
        args.steps = args.steps or 3
        args.warmup = 1 if args.warmup is None else args.warmup
        return run_cfg1(args)
    w = WORKLOADS[name]
    # default: a timed region of 10 s or more (80 x 125 ms for the headline workload), so that coarse telemetry sees the GPU busy
    if args.steps is None:
        args.steps = {"rtweekend1": 80, "overshadowed": 50, "mesh1m": 6, "mesh10m": 2, "spheres500": 20}[name]
    if args.warmup is None:
        args.warmup = 3 if w["bound"] == "valu_issue" else 1

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus)  # never returns

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if "shard" in w and world > 1:
        raise SystemExit(f"workload {name} is one rank's shard by definition: run it with --gpus 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP back end has no CPU fallback")
    # RT_BENCH_REHEARSAL=1: the N ranks share GPU 0 and talk over gloo (RCCL wants one GPU per rank).  Exercises the
    # launcher, the rank environment, sharded rendering, gather + scatter, the max-over-ranks timing and the JSON line on
    # a one-GPU box; the number it prints is NOT a measurement (and says so).
    rehearsal = os.environ.get("RT_BENCH_REHEARSAL") == "1" and world > 1
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    comm_device = torch.device("cpu") if rehearsal else device
    use_dist = world > 1 or os.environ.get("RT_BENCH_FORCE_DIST") == "1"  # the latter: rehearse the RCCL path on one GPU
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")  # only missing in the one-rank rehearsal (RT_BENCH_FORCE_DIST)
        os.environ.setdefault("MASTER_PORT", "29533")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    pkg = importlib.import_module("raytracing-rust_amd")
    hb = importlib.import_module("raytracing-rust_amd.hip_backend")
    D = importlib.import_module("raytracing-rust_amd.distributed")
    abi = pkg.abi
    WIDTH, HEIGHT, SPP = w["width"], w["height"], w["spp"]

    scene_desc, camera_params = load_workload(pkg, name)
    t0 = time.time()
    abi_devices = [int(x) for x in args.abi_devices.split(",")] if args.abi_devices else None
    if abi_devices and (world > 1 or "shard" in w):
        raise SystemExit("--abi-devices is the one-process multi-GPU path: use it with --gpus 1 and an unsharded workload")
    # BVH build + upload: not part of the timed region
    scene = hb.HipScene(scene_desc, devices=abi_devices) if abi_devices else hb.HipScene(scene_desc, device=local_rank)
    build_s = time.time() - t0
    cam = hb.camera_new(**camera_params)

    opts = workload_opts(abi, name)
    # A lane folds one work item at a time; by default an item is a whole pixel (the reference's sequential running mean), so
    # a GPU cannot use more lanes than it owns pixels, and even one GPU with the whole 1080p frame has only eight items per
    # resident lane: the last ones run while most of the chip has nothing left (7 % of config 2).  So the passes of a pixel
    # are split into S chunks (rt_render_opts.sample_split: the same samples, chunk means combined in fixed order, defined
    # identically in the oracle; the image moves by ~1e-7).  S is the LIBRARY's automatic choice (sample_split = 0), asked for
    # through rt_scene_auto_sample_split -- the one rule the C ABI, this file and tests/test_gpu_parity.py::test_full_size_*
    # share: 16, 32, 64, 64 for 1, 2, 4, 8 GPUs on config 2.  RT_BENCH_SPLIT overrides (A/B measurements).
    if "shard" in w:  # one rank's shard of a larger job: render it packed, nothing to gather
        shard_index, shard_count = w["shard"]
    else:
        shard_index, shard_count = rank, world
    sopts = D.shard_opts(opts, shard_index, shard_count)
    if abi_devices:
        sopts = opts  # whole frame, RT_LAYOUT_FRAME: the scene shards over its devices by itself
    split = int(os.environ["RT_BENCH_SPLIT"]) if os.environ.get("RT_BENCH_SPLIT") else scene.auto_sample_split(sopts)
    opts.sample_split = sopts.sample_split = split
    gather = D.ShardGather(opts, rank, world, comm_device) if ("shard" not in w and not abi_devices) else None
    n_shard_floats = hb.output_floats(sopts)
    shard = torch.zeros(n_shard_floats // 3, 3, dtype=torch.float32, device=device) if gather is None else gather.new_shard_buffer()
    frame = torch.empty(HEIGHT * WIDTH, 3, dtype=torch.float32, device=comm_device) if (rank == 0 and gather is not None) else None
    d_rays = torch.zeros(1, dtype=torch.int64, device=device)
    stream = torch.cuda.current_stream(device)
    shard_dev = torch.zeros_like(shard, device=device) if rehearsal else shard

    def step():
        scene.render_device(cam, sopts, shard_dev.data_ptr(), d_rays.data_ptr(), stream.cuda_stream)
        if rehearsal:
            shard.copy_(shard_dev)  # through the host: gloo moves CPU tensors
        return gather.gather(shard, frame) if gather is not None else shard

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step()
    barrier()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        # HIP events recorded around the kernel on its launch stream; reading them waits for that
        # kernel only, which the next step's launch on the same stream would do anyway
        kernel_ms.append(scene.last_kernel_ms()[0])
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=comm_device)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    rays = D.reduce_rays(d_rays.clone().to(comm_device), world)
    launch = scene.last_launch_info()
    # SURVEY 8(d) defines the metric on rt_render: the frame delivered to a HOST buffer (+ one D2H of W x H x 12 bytes and the
    # host synchronisation).  Timed here next to the device-resident figure above, on a few untimed-for-`value` steps.
    host_frame_ms = None
    if world == 1:
        scene.render(cam, sopts)
        n_host = max(2, min(5, args.steps))
        t0 = time.perf_counter()
        for _ in range(n_host):
            scene.render(cam, sopts)
        host_frame_ms = (time.perf_counter() - t0) / n_host * 1e3

    # What the kernels that are NOT compiled for this scene's own tree and materials deliver on the same workload: the headline
    # workload's tree is one node over two single-sphere leaves, for which the library picks rt::FeatPair; RT_TUNE_FEATURE_SET = 0
    # names the general spheres-only set instead (same pixels: tests/test_gpu_parity.py::test_every_kernel_variant...).  Untimed
    # for `value`: a few extra steps on a second handle of the same scene.
    general = None
    if world == 1 and name == "rtweekend1" and not abi_devices:
        g2 = hb.HipScene(scene_desc, device=local_rank)
        g2.set_tuning(abi.RT_TUNE_FEATURE_SET, 0)
        best = None
        for _ in range(4):
            g2.render_device(cam, sopts, shard_dev.data_ptr(), d_rays.data_ptr(), stream.cuda_stream)
            ms = g2.last_kernel_ms()[0]
            best = ms if best is None else min(best, ms)
        general = {"kernel": g2.last_launch_info()["kernel"], "kernel_ms": best, "value_general_kernel": WIDTH * HEIGHT * SPP / best / 1e3,
                   "note": "RT_TUNE_FEATURE_SET=0: the spheres-only kernels that serve any sphere scene, best of 4 untimed-for-value launches"}
        g2.close()

    if rank == 0:
        samples_per_step = (WIDTH * HEIGHT // (w["shard"][1] if "shard" in w else 1)) * SPP
        value = samples_per_step * args.steps / elapsed / 1e6
        k_ms = sum(kernel_ms) / len(kernel_ms)
        # the dominant (only) kernel: one render launch per step on this rank, covering 1/world of the samples
        launch_samples = samples_per_step / world
        k_s = k_ms * 1e-3
        # PMC-derived per-sample counters of this kernel variant, with the hash of the sources they were measured on
        counters = {}
        cp = os.path.join(ROOT, "profiles", "kernel_counters.json")
        if os.path.exists(cp):
            counters = json.load(open(cp)).get(name, {})
        current = bool(counters) and counters.get("source_hash") == source_hash() and counters.get("kernel") == launch["kernel"]
        roof = {"bound": w["bound"], "kernel": launch["kernel"], "kernel_ms": k_ms,
                "launch": {k: launch[k] for k in ("block_threads", "n_blocks", "blocks_per_cu", "waves_per_simd", "lds_bytes", "n_cus",
                                                  "sky_in_lds", "scene_in_lds")},
                # HBM bytes per launch (PMC, measured at N = 1).  N > 1: a rank writes 1/N of the frame and reads the same scene,
                # so its share of the one-GPU figure is the estimate printed, and the note says that it is one
                "traffic": (counters.get("hbm_bytes_per_launch") / world) if counters.get("hbm_bytes_per_launch") else None,
                "traffic_note": None if world == 1 else f"per rank: the one-GPU PMC figure / {world} (frame writes dominate; not re-measured at N = {world})",
                "counters_source": counters.get("source"), "counters_measured_on_this_build": current}
        if w["bound"] == "valu_issue":
            vps = counters.get("valu_wave_instructions_per_sample")
            achieved = vps * launch_samples / k_s if vps else None
            roof.update({"achieved": achieved / 1e9 if achieved else None, "peak": VALU_ISSUE_PEAK / 1e9, "unit": "G VALU wave-instructions/s",
                         "frac": achieved / VALU_ISSUE_PEAK if achieved else None,
                         "valu_wave_instructions_per_sample": vps, "lane_utilisation": counters.get("valu_lane_utilisation"),
                         **useful_valu_share(launch["kernel"]),
                         "note": "scene and sky tables live in SGPRs/LDS; the only HBM traffic is the output: with sample_split S every work item "
                                 f"writes its 12-byte chunk mean and combine_chunks_kernel reads them back and writes the frame ({launch['sample_split']} x 12 + 12 B "
                                 f"per pixel written, {launch['sample_split']} x 12 read = {(2 * launch['sample_split'] + 1) * 12 * WIDTH * HEIGHT / 1e6:.0f} MB per frame, `traffic` "
                                 "is what the PMC counters saw), so the bound is VALU issue: 256 CUs x 4 SIMD-32 x 2.4 GHz / 2 cycles per wave64 instruction"})
            # `peak` is the nominal rate (one wave64 VALU instruction per 2 cycles per SIMD).  MEASURED on this chip with in-kernel
            # cycle counters (profiles/r04_valu_issue.txt), four waves per SIMD issuing nothing else: a plain VALU instruction holds
            # its SIMD for 2.5 cycles, integer multiplies and packed f32 for 4.25, v_rcp / v_sqrt for 8.25 -- and a SCALAR instruction
            # of the same waves for about 2.15 (v_fma + s_alu interleaved 1:1: 4.65 cycles per pair).  `issue_model` adds that up for
            # this kernel: dynamic VALU and SALU counts (PMC) x those costs, the VALU mix from the static census of the binary
            # (profiles/valu_census.json; cold paths included, so an estimate), against the SIMD cycles the launch had.
            waves = int(launch.get("waves_per_simd") or 4)
            c_plain, c_half, c_trans, c_salu = ISSUE_CYCLES_BY_WAVES.get(waves, ISSUE_CYCLES_BY_WAVES[4 if waves < 4 else 8])
            mix = roof.get("valu_census_issue_cycles_static_mix")  # (the census prices the mix at four waves' costs: rescaled by the plain cost)
            if mix:
                mix = mix * c_plain / VALU_PLAIN_CYCLES_MEASURED
            sps = counters.get("salu_wave_instructions_per_sample")
            clock = (counters.get("effective_clock_ghz") or CLOCK_HZ / 1e9) * 1e9
            have = SIMDS * clock * k_s
            need_valu = vps * launch_samples * (mix or c_plain) if vps else None
            need_salu = sps * launch_samples * c_salu if sps else None
            roof["issue_model"] = {
                "waves_per_simd": waves,
                "cycles": {"plain_valu": c_plain, "half_rate_valu": c_half, "transcendental": c_trans, "salu": c_salu, "valu_static_mix": mix},
                "salu_wave_instructions_per_sample": sps,
                "effective_clock_ghz_when_profiled": counters.get("effective_clock_ghz"),
                "frac_of_plain_valu_ceiling": achieved / (VALU_ISSUE_PEAK * 2.0 / c_plain) if achieved else None,
                "simd_cycles_needed_over_available_valu_only": need_valu / have if need_valu else None,
                "simd_cycles_needed_over_available_valu_plus_salu": (need_valu + need_salu) / have if (need_valu and need_salu) else None,
                "source": "profiles/r04_valu_issue.txt, r04z_valu_issue_w5_w6.txt + profiles/valu_census.json + the PMC profile named in counters_source",
                "note": "a value near 1 means the SIMDs' instruction issue is what the launch waits for: only fewer, or cheaper, instructions help"}
        elif w["bound"] == "l2_request_rate":
            rps = counters.get("l2_read_requests_per_sample")
            hit = counters.get("l2_hit_rate")
            achieved = rps * launch_samples / k_s if rps else None
            ceiling = None
            if hit is not None:  # harmonic blend of the two measured ceilings at the kernel's own L2 hit rate
                ceiling = 1.0 / (hit / L2_FETCH_CEILING_RESIDENT + (1.0 - hit) / L2_FETCH_CEILING_BEYOND)
            roof.update({"achieved": achieved / 1e9 if achieved else None, "peak": L2_FETCH_CEILING_RESIDENT / 1e9, "unit": "G L1->L2 read requests/s",
                         "frac": achieved / L2_FETCH_CEILING_RESIDENT if achieved else None,
                         "l2_read_requests_per_sample": rps, "l2_hit_rate": hit,
                         "blended_ceiling_at_this_hit_rate": ceiling / 1e9 if ceiling else None,
                         "frac_of_blended_ceiling": achieved / ceiling if (achieved and ceiling) else None,
                         "lane_utilisation": counters.get("valu_lane_utilisation"),
                         "note": "every node step is a dependent, effectively random fetch of 48 bytes of a 64-byte record; peak = the measured rate of "
                                 "such fetches over an L2-resident set (237 G/s), 57 G/s beyond the L2s -- Infinity Cache or HBM alike: one 128-byte fabric "
                                 "line each, 7.3 TB/s (profiles/r04h_random_fetch_microbench.txt); blended_ceiling = harmonic blend of the two at the "
                                 "kernel's own L2 hit rate (a model: the fractions are printed unclamped)"})
        ws = None
        if world == 1 and not args.no_walk_stats:
            ws, ws_error = walk_stats(name)
            if ws_error:
                roof["walk_stats_error"] = ws_error
                print(f"bench.py: walk statistics unavailable: {ws_error}", file=sys.stderr, flush=True)
        if ws and "primary_phase_lanes" in ws:
            # NS1: how full the waves are at the level cross-lane compaction could fix (paths that wait for the other
            # super-phase).  The VALU lane utilisation above is lower because lanes also idle INSIDE a phase (branches of
            # the shading code), which moving path state between lanes cannot touch.
            roof["phase_participation"] = dict(ws, note="lanes of 64 holding a path that takes part in the voted super-phase, -DRT_STATS build, untimed pass")
        if ws and "node_steps_per_sample" in ws:
            # a node step fetches the first 48 bytes of a 64-byte wide-node record (the compact node of round 3: three
            # 16-byte pieces; rounds 1-2 fetched all 64), a primitive test one 48-byte record
            wb = ws["node_steps_per_sample"] * 48 + ws["primitive_tests_per_sample"] * 48
            roof["pruned_walk"] = dict(ws, bytes_per_sample=wb, requested_GBps=wb * launch_samples / k_s / 1e9,
                                       note="counted by the -DRT_STATS build on an untimed pass: 48 B per node step + 48 B per primitive test")
        if w["bound"] == "hbm":
            wb = roof.get("pruned_walk", {}).get("bytes_per_sample")
            achieved = wb * launch_samples / k_s / 1e9 if wb else None
            measured = counters.get("hbm_bytes_per_launch") / counters.get("kernel_ms_when_measured") / 1e6 if counters.get("hbm_bytes_per_launch") else None
            roof.update({"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS if achieved else None,
                         "achievable_peak": HBM_ACHIEVABLE_GBS,
                         "fabric_GBps_from_pmc": measured, "fabric_frac_of_peak": measured / HBM_PEAK_GBS if measured else None,
                         "random_line_ceiling_GBps": RANDOM_LINE_CEILING_GBS,
                         "fabric_frac_of_random_line_ceiling": measured / RANDOM_LINE_CEILING_GBS if measured else None,
                         "l2_hit_rate": counters.get("l2_hit_rate"),
                         "lane_utilisation": counters.get("valu_lane_utilisation"),
                         "note": "achieved = bytes the pruned walk REQUESTS (48 B/node step + 48 B/primitive test, counted live) / kernel time; "
                                 "fabric_GBps_from_pmc = what crossed the fabric (Infinity Cache + HBM, 128-byte lines) in the committed profile "
                                 "of this build: above the requested bytes because a node step uses 48 bytes of the 128-byte line it arrives in "
                                 "(the line count, not the byte count, is what the fabric serves).  fabric bytes are NOT HBM bytes (Infinity-Cache hits are "
                                 "counted, MI355X_MICROARCH.md): the ceiling they are compared with is the measured rate of dependent random 128-byte line "
                                 "fetches beyond the L2s, 57 G lines/s = 7.3 TB/s, the same from the Infinity Cache and from HBM "
                                 "(profiles/r04h_random_fetch_microbench.txt).  Fewer lines alone do not shorten the launch: eight-child nodes moved 22 % "
                                 "fewer lines over the fabric and the launch got 6 % slower "
                                 "(profiles/archive_r03/r04w8_mesh10m_pmc.json against r03n_mesh10m_pmc.json there; HISTORY.md): the launch follows lane-level 16-byte "
                                 "fetches and VALU instructions, HBM is the nearest roofline"})
        ab = w["bytes"]
        ap = os.path.join(ROOT, "profiles", "algorithmic_bytes.json")  # the oracle's own count, when it has been made for this workload
        if os.path.exists(ap):
            ab = json.load(open(ap)).get(f"{name}_{WIDTH}x{HEIGHT}_mis", {}).get("bytes_per_sample", ab)
        if ab:
            roof["survey_8d_algorithmic"] = {"bytes_per_sample": ab, "requested_GBps": ab * launch_samples / k_s / 1e9,
                                             "note": "requested bytes under REFERENCE traversal semantics (no pruning, oracle-counted); not HBM bytes, not a roofline fraction"}
        out = {
            "metric": "Msamples/s on rtweekend1.ssml 1920x1080x1024spp" if name == "rtweekend1" else f"Msamples/s on {name} {WIDTH}x{HEIGHT}x{SPP}spp",
            "value": value,
            "unit": "Msamples/s",
            "n_gpus": world if not abi_devices else len(set(abi_devices)),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if not rehearsal else f"synthetic -- REHEARSAL: {world} ranks share one GPU over gloo; not a measurement",
            "config": {"workload": f"{w['label']} {WIDTH}x{HEIGHT} {SPP}spp MIS max_depth=50 rr=3 seed={w['seed']} ({w['config']})",
                       "parallelism": (f"{world} ranks (one per GPU), 8x8 tiles interleaved t % {world}, replicated BVH, one RCCL gather per frame"
                                       if world > 1 else
                                       (f"one process, devices {abi_devices} through rt_scene_create_multi (tiles t % {len(abi_devices)}, in-process gather)"
                                        if abi_devices else ("1 GPU" + (f", shard {w['shard'][0]} of {w['shard'][1]}" if "shard" in w else "")))),
                       "sample_split": launch["sample_split"],
                       "sample_split_rule": "RT_BENCH_SPLIT" if os.environ.get("RT_BENCH_SPLIT") else "rt_scene_auto_sample_split (the library's automatic choice)",
                       "ms_per_step_host_frame": host_frame_ms,
                       "value_host_frame": (samples_per_step / host_frame_ms / 1e3) if host_frame_ms else None,
                       "host_frame_note": "rt_render: the same step with the frame copied to a host buffer (SURVEY 8(d)'s wall-seconds of rt_render); "
                                          "`value` is the HBM-resident rate the bench contract asks for",
                       "abi_devices": abi_devices,
                       "general_kernel": general,
                       "value_general_kernel": general["value_general_kernel"] if general else None,
                       "samples_per_step": samples_per_step, "rays_shot_per_step": int(rays.item()),
                       "scene_build_s": build_s},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pkg, name, scene_desc, camera_params)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
