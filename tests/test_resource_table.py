"""The kernels' register / spill / scratch / LDS numbers are part of the product's performance (DESIGN.md section 4 records
3.5 % and 6 % losses from edits that only perturbed register allocation), so the table of the shipped library is committed as
profiles/resource_table.json and a build whose DEFAULT kernels differ from it fails here: whoever changes a kernel has to
rebuild, look at the difference (`python tests/probes/resource_table.py --diff`), measure, and commit the new table with the change."""
import importlib.util
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("resource_table", os.path.join(ROOT, "tests", "probes", "resource_table.py"))
rtab = importlib.util.module_from_spec(spec)
spec.loader.exec_module(rtab)


@pytest.fixture(scope="module")
def built_table():
    if not os.path.exists(rtab.READELF):
        pytest.skip("llvm-readelf not available")
    csrc = os.path.join(ROOT, "raytracing-rust_amd", "csrc")
    srcs = [os.path.join(csrc, n) for n in os.listdir(csrc) if n.endswith((".hip", ".h", ".cpp")) or n == "Makefile"]
    srcs.append(os.path.join(ROOT, "include", "rt_detmath.h"))
    if not os.path.exists(rtab.LIB) or os.path.getmtime(rtab.LIB) < max(os.path.getmtime(p) for p in srcs):
        subprocess.run(["make", "-C", csrc, "-s", "../librt_hip.so"], check=True)  # the table must describe THESE sources
    return rtab.extract(rtab.LIB)


def test_committed_table_describes_the_current_sources():
    table = json.load(open(rtab.TABLE))
    assert table["source_hash"] == rtab.source_hash(), (
        "kernel sources changed since profiles/resource_table.json was written: rebuild (make -C raytracing-rust_amd/csrc), "
        "review `python tests/probes/resource_table.py --diff`, then `--write` and commit the table with the change")


def test_default_kernels_match_the_committed_table(built_table):
    committed = json.load(open(rtab.TABLE))["kernels"]
    default_built = {k: v for k, v in built_table.items() if v["default"]}
    default_committed = {k: v for k, v in committed.items() if v["default"]}
    assert len(default_built) >= 30 + 4  # 30 render instantiations the library selects from + the batch-query kernels
    differences = rtab.diff(default_built, default_committed)
    assert not differences, "\n".join(differences)


def test_headline_kernels_hold_their_budgets(built_table):
    """the kernels of BASELINE configs 2 and 3 (coarse schedule, MIS, sky tables in LDS): 4 waves per SIMD and, since the
    kernel arguments are read where they are used (RenderArgs in rt_render.hip), no SGPR spills to speak of and next to no
    scratch in the spheres-only kernel"""
    spheres = built_table["void rt::render_kernel<1, false, false, true, rt::Feat<false, false, false, false>, false>"]
    simple = built_table["void rt::render_kernel<1, false, false, true, rt::Feat<true, true, false, false>, false>"]
    assert spheres["waves_per_simd_by_registers"] >= 4 and simple["waves_per_simd_by_registers"] >= 4
    # (running BOUNCE in the iteration whose PRIMARY filled it parks 13 loop-invariant / rarely used registers in 40 bytes of
    # scratch and is still 1 % faster than the spill-free loop, same-box A/B gpurun_out/r03r: that much is allowed, no more)
    # (15 with the material handles of rt_types.h, same 40 bytes of scratch)
    assert spheres["private_segment_fixed_size"] <= 48 and spheres["vgpr_spill_count"] <= 16
    assert spheres["sgpr_spill_count"] <= 24
    # the kernel BASELINE config 2 itself runs (rtweekend1's tree is one node over two single-sphere leaves: rt_types.h FeatPair,
    # the general walk not compiled in, material types known from what was hit, the scene itself read from the kernel
    # arguments once per super-phase): no spilled register of either kind, no scratch
    # Round 4: at most 80 VGPRs = SIX waves per SIMD (two workgroups of 768 threads per CU, rt_render.hip RT_PAIR_BLOCK) since the
    # persistent loop stopped keeping the lane state in two register sets (101 VGPRs before); one loop-invariant 64-bit scalar
    # may sit in two lanes of a VGPR (two v_readlane per iteration), nothing else.
    pair = built_table["void rt::render_kernel<1, false, false, true, rt::FeatPair, false>"]
    assert pair["waves_per_simd_by_registers"] >= 6
    assert pair["private_segment_fixed_size"] == 0 and pair["vgpr_spill_count"] == 0 and pair["sgpr_spill_count"] <= 2
    # config 3's kernel: no spilled VGPR (round 4)
    assert simple["vgpr_spill_count"] == 0 and simple["private_segment_fixed_size"] == 0
