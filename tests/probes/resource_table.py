#!/usr/bin/env python3
"""Register / spill / scratch / LDS table of every kernel in the SHIPPED librt_hip.so (gfx950 code object metadata).

The render kernels sit on a register-allocation knife edge (a dead pointer expression cost 12 spilled
registers and 3.5 %, an out-of-line claim loop 6 %), so the table of the default kernels is committed
(profiles/resource_table.json) and tests/test_resource_table.py fails when a build's numbers differ from it: a change of
register allocation has to be SEEN (and measured) in the commit that causes it.

  python tests/probes/resource_table.py            print the table of the built library
  python tests/probes/resource_table.py --write    ... and rewrite profiles/resource_table.json (after a rebuild)
  python tests/probes/resource_table.py --diff     differences between the built library and the committed table
"""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(ROOT, "raytracing-rust_amd", "librt_hip.so")
TABLE = os.path.join(ROOT, "profiles", "resource_table.json")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
CXXFILT = "c++filt"
KEYS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
        "group_segment_fixed_size", "kernarg_segment_size", "max_flat_workgroup_size")


def code_object(so_path):
    """the gfx950 ELF inside the library's clang offload bundle"""
    data = open(so_path, "rb").read()
    i = data.find(b"__CLANG_OFFLOAD_BUNDLE__")
    if i < 0:
        raise RuntimeError(f"{so_path}: no offload bundle")
    n = struct.unpack_from("<Q", data, i + 24)[0]
    off = i + 32
    for _ in range(n):
        o, sz, tl = struct.unpack_from("<QQQ", data, off)
        off += 24
        triple = data[off:off + tl].decode()
        off += tl
        if "gfx950" in triple:
            return data[i + o:i + o + sz]
    raise RuntimeError(f"{so_path}: no gfx950 code object")


def waves_per_simd(vgprs, agprs=0):
    """occupancy the unified 512-register file allows (allocation granule 8)"""
    total = max(1, vgprs + agprs)
    total = (total + 7) // 8 * 8
    return min(8, 512 // total)


def extract(so_path=LIB):
    with tempfile.NamedTemporaryFile(suffix=".elf") as f:
        f.write(code_object(so_path))
        f.flush()
        notes = subprocess.run([READELF, "--notes", f.name], check=True, capture_output=True, text=True).stdout
    kernels = {}
    for blk in re.split(r"\n\s+- \.agpr_count:|\n\s+- \.args:", notes):
        nm = re.search(r"\.name:\s+(\S+)", blk)
        if not nm or not nm.group(1).startswith("_Z"):
            continue
        d = {}
        for key in KEYS:
            m = re.search(r"\.%s:\s+(\d+)" % key, "\n.agpr_count:" + blk if key == "agpr_count" else blk)
            if m:
                d[key] = int(m.group(1))
        kernels[nm.group(1)] = d
    names = list(kernels)
    dem = subprocess.run([CXXFILT] + names, check=True, capture_output=True, text=True).stdout.splitlines()
    out = {}
    for mangled, pretty in zip(names, dem):
        d = kernels[mangled]
        pretty = re.sub(r"\(.*$", "", pretty)  # drop the parameter list
        d["waves_per_simd_by_registers"] = waves_per_simd(d.get("vgpr_count", 0), d.get("agpr_count", 0))
        # the exchange variants (RT_TUNE_EXCHANGE, off by default) ride along but are not guarded
        d["default"] = not pretty.endswith(", true>")
        out[pretty] = d
    return out


def source_hash():
    sys.path.insert(0, ROOT)
    import bench
    return bench.source_hash()


def diff(built, committed):
    lines = []
    for name in sorted(set(built) | set(committed)):
        a, b = committed.get(name), built.get(name)
        if a is None or b is None:
            lines.append(f"{'only in the build' if a is None else 'only in the table'}: {name}")
            continue
        ch = {k: (a.get(k), b.get(k)) for k in set(a) | set(b) if a.get(k) != b.get(k)}
        if ch:
            lines.append(f"{name}: " + ", ".join(f"{k} {x} -> {y}" for k, (x, y) in sorted(ch.items())))
    return lines


if __name__ == "__main__":
    table = extract()
    if "--diff" in sys.argv:
        old = json.load(open(TABLE))
        print("\n".join(diff(table, old["kernels"])) or "no differences")
    elif "--write" in sys.argv:
        json.dump({"source_hash": source_hash(), "library": "raytracing-rust_amd/librt_hip.so", "arch": "gfx950",
                   "how": "python tests/probes/resource_table.py --write (after make -C raytracing-rust_amd/csrc)",
                   "kernels": table}, open(TABLE, "w"), indent=1, sort_keys=True)
        print(f"wrote {TABLE}: {len(table)} kernels")
    else:
        for name, d in sorted(table.items()):
            print(f"{d['vgpr_count']:4d} v {d['sgpr_count']:4d} s  spill v{d['vgpr_spill_count']:3d} s{d['sgpr_spill_count']:3d}  scratch {d['private_segment_fixed_size']:4d}  "
                  f"{d['waves_per_simd_by_registers']} w/SIMD  {'' if d['default'] else '(opt-in) '}{name}")
