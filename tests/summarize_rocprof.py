"""Turns a gpurun_out/prof_<tag>/ directory (tests/run_rocprof.sh) into the committed evidence under
profiles/: <tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats) and <tag>_pmc.json (PMC counters of
the render kernel, summed per launch), and refreshes profiles/hbm_traffic.json for bench.py."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# summarize_rocprof.py <tag> [samples per launch] [key in hbm_traffic.json]
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
samples_per_launch = int(sys.argv[2]) if len(sys.argv) > 2 else 1920 * 1080 * 1024
traffic_key = sys.argv[3] if len(sys.argv) > 3 else "rtweekend1_1920x1080x1024_mis"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv"))[0]
rows = list(csv.DictReader(open(stats)))
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=rows[0].keys())
    w.writeheader()
    for r in rows:
        r["Name"] = r["Name"][:120]
        w.writerow(r)
render = [r for r in rows if "render_kernel" in r["Name"]][0]

counters = collections.OrderedDict()
launches = {}
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
        per = collections.defaultdict(float)
        ids = set()
        info = {}
        for r in csv.DictReader(open(f)):
            if "render_kernel" in r["Kernel_Name"]:
                per[r["Counter_Name"]] += float(r["Counter_Value"])
                ids.add(r["Dispatch_Id"])
                info = {k: r.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size",
                                               "Grid_Size", "Workgroup_Size")}
        for k, v in per.items():
            counters[k] = v / max(1, len(ids))
        launches.update(info)

c = counters
summary = {
    "tag": tag,
    "command": open(os.path.join(src, "command.txt")).read().strip() if os.path.exists(os.path.join(src, "command.txt")) else
               "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline (stats); "
               "rocprofv3 --pmc <set> --kernel-trace -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline (one run per set)",
    "samples_per_launch": samples_per_launch,
    "kernel": render["Name"],
    "kernel_calls": int(render["Calls"]),
    "kernel_avg_ms": float(render["AverageNs"]) / 1e6,
    "kernel_share_of_gpu_time_pct": float(render["Percentage"]),
    "dispatch": launches,
    "counters_per_launch": counters,
}
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    # MI355X_MICROARCH.md (HBM): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reads half the fetched bytes
    hbm = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
    summary["hbm_bytes_per_launch"] = hbm
    summary["hbm_note"] = "(2*FETCH_SIZE + WRITE_SIZE) KiB; the x2 is the guide's gfx950 correction, calibrated for wide streaming reads only"
if "TCC_HIT_sum" in c:
    summary["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
    summary["valu_lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0)
if "SQ_WAVE_CYCLES" in c and "SQ_WAIT_ANY" in c:
    summary["wave_cycle_shares"] = {"active_inst_any": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"],
                                    "wait_any": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
                                    "wait_inst_any": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]}
if "SQ_INSTS_VALU" in c:
    samples = samples_per_launch
    summary["valu_wave_instructions_per_sample"] = c["SQ_INSTS_VALU"] / samples
    summary["valu_wave_instructions_per_second"] = c["SQ_INSTS_VALU"] / (summary["kernel_avg_ms"] * 1e-3)
    summary["valu_issue_peak_per_second"] = 256 * 4 * 2.4e9 / 2  # 1024 SIMD-32 x one wave64 VALU op per 2 cycles
    summary["valu_issue_fraction_of_peak"] = summary["valu_wave_instructions_per_second"] / summary["valu_issue_peak_per_second"]
if "GRBM_GUI_ACTIVE" in c:
    summary["effective_clock_ghz"] = c["GRBM_GUI_ACTIVE"] / 8 / (summary["kernel_avg_ms"] * 1e-3) / 1e9
json.dump(summary, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1)

tp = os.path.join(dst, "hbm_traffic.json")
traffic = json.load(open(tp)) if os.path.exists(tp) else {}
if "hbm_bytes_per_launch" in summary:
    traffic[traffic_key] = {"hbm_bytes_per_launch": summary["hbm_bytes_per_launch"], "source": f"profiles/{tag}_pmc.json"}
    json.dump(traffic, open(tp, "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k not in ("counters_per_launch", "command")}, indent=1))
