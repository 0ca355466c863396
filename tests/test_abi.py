"""The C-ABI boundary without a GPU: the library loads, exports every symbol include/rt_hip.h
declares, the ctypes mirror has the C compiler's struct sizes, host-only entry points work, and
compute entry points fail loudly (there is no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

import scenes

abi = scenes.abi
ROOT = scenes.ROOT


def test_header_symbols_are_listed_and_exported(hb):
    header = open(os.path.join(ROOT, "include", "rt_hip.h")).read()
    declared = set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", header))
    declared -= {"rt_scene"}  # type name
    assert declared == set(abi.EXPORTED_SYMBOLS), declared ^ set(abi.EXPORTED_SYMBOLS)
    lib = hb.lib()
    for sym in abi.EXPORTED_SYMBOLS:
        assert hasattr(lib, sym), sym
    assert lib.rt_abi_version() == abi.RT_ABI_VERSION


def test_struct_sizes_match_the_c_compiler():
    names = list(abi.EXPECTED_SIZES)
    src = '#include <stdio.h>\n#include "rt_hip.h"\nint main(void){' + "".join(
        f'printf("{n} %zu\\n", sizeof({n}));' for n in names) + "return 0;}"
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "s.c"); exe = os.path.join(td, "s")
        open(c, "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    sizes = dict(line.split() for line in out.strip().splitlines())
    for n, (cls, expected) in abi.EXPECTED_SIZES.items():
        assert int(sizes[n]) == expected == C.sizeof(cls), (n, sizes[n], expected, C.sizeof(cls))


def test_header_compiles_as_c_and_cpp():
    for compiler, flags in (("gcc", ["-std=c99", "-x", "c"]), ("g++", ["-std=c++17", "-x", "c++"])):
        for header in ("rt_hip.h", "rt_detmath.h"):
            subprocess.run([compiler, *flags, "-fsyntax-only", "-Wall", "-Werror", "-Wno-unused-function",
                            os.path.join(ROOT, "include", header)], check=True)


def test_camera_new_matches_oracle(hb, O):
    for name in ("rtweekend1", "overshadowed"):
        p = scenes.load_ssml(name).camera_params
        assert bytes(hb.camera_new(**p)) == bytes(O.camera_new(**p))
    assert bytes(hb.camera_new(**scenes.MESH_CAMERA)) == bytes(O.camera_new(**scenes.MESH_CAMERA))


def test_output_sizes_and_shard_order(hb):
    o = abi.default_render_opts(50, 30, 1)
    assert hb.output_floats(o) == 50 * 30 * 3
    o.shard_index, o.shard_count = 1, 3
    o.output_layout = abi.RT_LAYOUT_SHARD
    tiles = 7 * 4  # ceil(50/8) x ceil(30/8)
    owned = len(range(1, tiles, 3))
    assert hb.output_floats(o) == owned * 64 * 3
    order = hb.shard_pixel_order(o)
    valid = order[order != np.uint64(abi.NO_INDEX)]
    assert len(np.unique(valid)) == len(valid)
    # all three shards together cover every pixel exactly once
    seen = []
    for k in range(3):
        o.shard_index = k
        od = hb.shard_pixel_order(o)
        seen.append(od[od != np.uint64(abi.NO_INDEX)])
    allp = np.sort(np.concatenate(seen))
    assert np.array_equal(allp, np.arange(50 * 30, dtype=np.uint64))
    # first tile of shard 0 is the 8x8 block at the origin, row-major
    o.shard_index = 0
    first = hb.shard_pixel_order(o)[:64].reshape(8, 8)
    assert np.array_equal(first, (np.arange(8)[:, None] * 50 + np.arange(8)[None, :]).astype(np.uint64))


def test_option_validation(hb):
    o = abi.default_render_opts(1, 30, 1)
    with pytest.raises(hb.RtHipError) as e:
        hb.output_floats(o)
    assert e.value.code == abi.RT_ERR_INVALID_ARGUMENT
    o = abi.default_render_opts(50, 30, 1); o.shard_index, o.shard_count = 3, 3
    with pytest.raises(hb.RtHipError):
        hb.output_floats(o)


def test_no_cpu_fallback(hb):
    """Without a HIP device the product must fail loudly, never compute on the CPU."""
    if hb.device_count() > 0:
        pytest.skip("a GPU is present; this check is for the CPU-only container")
    with pytest.raises(hb.RtHipError) as e:
        hb.HipScene(scenes.load_ssml("rtweekend1").scene)
    assert e.value.code == abi.RT_ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value)


def test_product_does_not_reference_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/"""
    pkg_dir = os.path.join(ROOT, "raytracing-rust_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".h", ".hpp", ".cpp", ".hip", ".c")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower().replace("oracle/ (the cpu checker)", ""), os.path.join(dirpath, f)
    out = subprocess.run(["ldd", os.path.join(pkg_dir, "librt_hip.so")], capture_output=True, text=True).stdout
    assert "liboracle" not in out


def test_cpp_host_header_compiles_and_fails_loudly_without_a_gpu(hb, tmp_path):
    """include/rt_hip.hpp + tests/cpp/host_demo.cpp: a compiled host above the C ABI"""
    exe = str(tmp_path / "host_demo")
    lib_dir = os.path.join(ROOT, "raytracing-rust_amd")
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "host_demo.cpp"), "-o", exe, "-L", lib_dir, "-lrt_hip",
                    f"-Wl,-rpath,{lib_dir}"], check=True)
    if hb.device_count() > 0:
        pytest.skip("a GPU is present; the GPU-side check is tests/test_gpu_parity.py")
    r = subprocess.run([exe, str(tmp_path / "o.f32"), str(tmp_path / "o.png"), "8", "8", "1", "0"], capture_output=True, text=True)
    assert r.returncode == 1 and "no CPU fallback" in r.stderr
