"""Turns a gpurun_out/prof_<tag>/ directory (tests/probes/run_rocprof.sh) into the committed evidence under
profiles/: <tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats), <tag>_pmc.json (PMC counters of the
render kernel, per launch) and the workload's entry in profiles/kernel_counters.json, which bench.py reads
for the per-sample counters no kernel can count for itself (VALU instructions, L2 requests, fabric bytes).

    python tests/probes/summarize_rocprof.py <tag> [workload]
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (WORKLOADS: frame sizes)

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def read(name, default=None):
    p = os.path.join(src, name)
    return open(p).read().strip() if os.path.exists(p) else default


workload = sys.argv[2] if len(sys.argv) > 2 else read("workload.txt", "rtweekend1")
w = bench.WORKLOADS[workload]
samples_per_launch = (w["width"] * w["height"] // (w["shard"][1] if "shard" in w else 1)) * w["spp"]

# (a tag that was profiled twice keeps both runs' files under gpurun_out/: the newest run is the one that counts)
stats = max(glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv")), key=os.path.getmtime)
rows = list(csv.DictReader(open(stats)))
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    wr = csv.DictWriter(f, fieldnames=rows[0].keys())
    wr.writeheader()
    for r in rows:
        r["Name"] = r["Name"][:120]
        wr.writerow(r)
# the render kernel THE BENCH LINE names (bench.py also launches the general spheres-only kernel on the headline workload, untimed
# for `value`: its dispatches must not be mixed into the headline kernel's counters)
def norm(name):
    return name.replace("void ", "").split("(")[0].replace("> >", ">>").strip()


wanted = None
tb = read("trace_bench.json")
if tb:
    try:
        wanted = json.loads([l for l in tb.splitlines() if l.startswith("{")][-1])["roofline"]["kernel"]
    except Exception:
        wanted = None
render_rows = [r for r in rows if "render_kernel" in r["Name"]]
render = ([r for r in render_rows if wanted and norm(r["Name"]) == wanted] or render_rows)[0]
wanted = wanted or norm(render["Name"])

counters = collections.OrderedDict()
launches = {}
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in sorted(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)[-1:]:
        per = collections.defaultdict(float)
        ids = set()
        info = {}
        for r in csv.DictReader(open(f)):
            if "render_kernel" in r["Kernel_Name"] and norm(r["Kernel_Name"]) == wanted:
                per[r["Counter_Name"]] += float(r["Counter_Value"])
                ids.add(r["Dispatch_Id"])
                info = {k: r.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size",
                                               "Grid_Size", "Workgroup_Size")}
                info["kernel_name"] = r["Kernel_Name"]
        for k, v in per.items():
            counters[k] = v / max(1, len(ids))
        launches.update(info)

c = counters
kernel_ms = float(render["AverageNs"]) / 1e6
summary = {
    "tag": tag,
    "workload": workload,
    "command": read("command.txt"),
    "source_hash": read("source_hash.txt"),
    "samples_per_launch": samples_per_launch,
    "kernel": render["Name"],
    "kernel_calls": int(render["Calls"]),
    "kernel_avg_ms": kernel_ms,
    "kernel_share_of_gpu_time_pct": float(render["Percentage"]),
    "dispatch": launches,
    "counters_per_launch": counters,
}
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    # MI355X_MICROARCH.md (HBM): FETCH_SIZE / WRITE_SIZE are in KiB.  The guide's x2 on FETCH_SIZE is calibrated for
    # wide coalesced streaming reads (128-byte requests tallied at 64); whether it applies to THIS access pattern is
    # decided from the request-size split below, not assumed.
    summary["fetch_bytes_as_reported"] = c["FETCH_SIZE"] * 1024.0
    summary["write_bytes"] = c["WRITE_SIZE"] * 1024.0
if "TCC_EA0_RDREQ_sum" in c:
    rd, rd32, bub = c["TCC_EA0_RDREQ_sum"], c.get("TCC_EA0_RDREQ_32B_sum", 0.0), c.get("TCC_BUBBLE_sum", 0.0)
    # profiles/r02_fabric_counter_calibration.txt: on gfx950 every non-32B fabric read request is one 128-byte L2 line
    # (streaming control: 2 GiB / 16.78 M requests; random 64- and 128-byte records: one request per record either way);
    # counter_defs' FETCH_SIZE formula tallies them at 64 because TCC_BUBBLE stays 0
    summary["fabric_read_requests"] = {"all": rd, "32B": rd32, "128B_bubble": bub, "to_dram": c.get("TCC_EA0_RDREQ_DRAM_sum"),
                                       "bytes_at_128B_lines": (rd - rd32) * 128 + rd32 * 32,
                                       "bytes_formula_of_counter_defs": bub * 128 + (rd - bub - rd32) * 64 + rd32 * 32}
if "TCC_HIT_sum" in c:
    summary["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    summary["l2_miss_bytes_at_128B_lines"] = c["TCC_MISS_sum"] * 128.0
if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
    summary["valu_lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0)
if "SQ_WAVE_CYCLES" in c and "SQ_WAIT_ANY" in c:
    summary["wave_cycle_shares"] = {"active_inst_any": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"],
                                    "wait_any": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
                                    "wait_inst_any": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]}
if "SQ_INSTS_VALU" in c:
    summary["valu_wave_instructions_per_sample"] = c["SQ_INSTS_VALU"] / samples_per_launch
    summary["valu_wave_instructions_per_second"] = c["SQ_INSTS_VALU"] / (kernel_ms * 1e-3)
    summary["valu_issue_peak_per_second"] = bench.VALU_ISSUE_PEAK  # 1024 SIMD-32 x one wave64 VALU op per 2 cycles
    summary["valu_issue_fraction_of_peak"] = summary["valu_wave_instructions_per_second"] / bench.VALU_ISSUE_PEAK
if "SQ_INSTS_SALU" in c:  # scalar instructions take vector issue slots from their SIMD too (profiles/r04_valu_issue.txt)
    summary["salu_wave_instructions_per_sample"] = c["SQ_INSTS_SALU"] / samples_per_launch
    summary["smem_wave_instructions_per_sample"] = c.get("SQ_INSTS_SMEM", 0.0) / samples_per_launch
    summary["lds_wave_instructions_per_sample"] = c.get("SQ_INSTS_LDS", 0.0) / samples_per_launch
if "TCP_TCC_READ_REQ_sum" in c:
    summary["l2_read_requests_per_sample"] = c["TCP_TCC_READ_REQ_sum"] / samples_per_launch
    summary["l2_read_requests_per_second"] = c["TCP_TCC_READ_REQ_sum"] / (kernel_ms * 1e-3)
    if "TCP_TCC_WRITE_REQ_sum" in c:
        summary["l2_write_requests_per_sample"] = c["TCP_TCC_WRITE_REQ_sum"] / samples_per_launch
if "GRBM_GUI_ACTIVE" in c:
    summary["effective_clock_ghz"] = c["GRBM_GUI_ACTIVE"] / 8 / (kernel_ms * 1e-3) / 1e9
json.dump(summary, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1)

# what bench.py prints next to its live kernel time
kp = os.path.join(dst, "kernel_counters.json")
kc = json.load(open(kp)) if os.path.exists(kp) else {}
entry = {"source": f"profiles/{tag}_pmc.json", "source_hash": summary["source_hash"], "kernel_ms_when_measured": kernel_ms,
         "kernel": (launches.get("kernel_name") or render["Name"]).replace("void ", "").split("(")[0].replace("> >", ">>")}
for k in ("valu_wave_instructions_per_sample", "valu_lane_utilisation", "l2_read_requests_per_sample", "l2_hit_rate",
          "salu_wave_instructions_per_sample", "smem_wave_instructions_per_sample", "lds_wave_instructions_per_sample", "effective_clock_ghz",
          "wave_cycle_shares"):
    if k in summary:
        entry[k] = summary[k]
if "fetch_bytes_as_reported" in summary:
    # fabric bytes: every L2 miss moves one 128-byte line when requests are 128-byte (streaming), 64 when they are
    # 64-byte; the request-size counters decide (fabric_read_requests).  Default to the counter_defs formula.
    fr = summary.get("fabric_read_requests")
    rd_bytes = fr["bytes_at_128B_lines"] if fr else 2.0 * summary["fetch_bytes_as_reported"]
    entry["hbm_bytes_per_launch"] = rd_bytes + summary["write_bytes"]
    entry["hbm_bytes_note"] = ("fabric (Infinity Cache + HBM) bytes: TCC_EA0_RDREQ x 128-byte lines (= 2 x FETCH_SIZE, "
                               "profiles/r02_fabric_counter_calibration.txt) + WRITE_SIZE")
kc[workload] = entry
json.dump(kc, open(kp, "w"), indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in summary.items() if k not in ("counters_per_launch", "command")}, indent=1))
