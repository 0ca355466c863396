"""raytracing-rust_amd/csrc/rt_lean.h: the branch-free sin+cos / acos / atan2 the kernels use return the bits of
include/rt_detmath.h (the arithmetic contract the oracle calls), checked by enumeration on the host.

The division / square-root short forms of the same header exist only on the device (they are built from v_rcp_f32 /
v_sqrt_f32); tests/test_gpu_parity.py::test_lean_arithmetic_matches_the_ieee_operators runs those against the plain
operators on the GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def lean_check(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    exe = str(tmp_path_factory.mktemp("lean") / "lean_check")
    subprocess.run([HIPCC, "-x", "hip", "--cuda-host-only", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-mfma",
                    "-Wno-unused-function", "-o", exe, os.path.join(ROOT, "tests", "cpp", "lean_check.cpp"), "-lpthread"], check=True)
    return exe


def _run(exe, stride):
    out = subprocess.run([exe, str(stride)], check=True, capture_output=True, text=True, timeout=900).stdout
    res = {}
    for line in out.splitlines():
        name, n, bad = line.split()
        res[name] = (int(n), int(bad))
    return res


def test_lean_elementary_functions_equal_detmath(lean_check):
    # stride 13: every 13th float of [0, 2^22] (both signs) for sin + cos, every 13th of ALL bit patterns for acos,
    # the atan2 grids: ~0.6 G evaluations, about 20 s on 8 cores.  RT_LEAN_EXHAUSTIVE=1 runs stride 1 (every float).
    stride = 1 if os.environ.get("RT_LEAN_EXHAUSTIVE") == "1" else 13
    res = _run(lean_check, stride)
    assert set(res) == {"sincos", "acos", "atan2_grid", "atan2_ratio"}
    for name, (n, bad) in res.items():
        assert n > 1000 and bad == 0, (name, n, bad)


def test_division_by_launch_constants_is_verified(hb):
    """csrc/rt_lean.h div_by_verified: x / c as q0 = x * rc, q = fma(fma(-c, q0, x), rc, q0) with rc = RN(1 / c) -- used by the
    kernels only for divisors the HOST has verified over all 2^23 significands (csrc/rt_build.cpp verified_reciprocal; the C ABI
    exposes that check as rt_selftest_division).  Here: pi and 2 pi (the kernels' compile-time reciprocals must be the verified
    ones), the image sizes and sky resolutions of the BASELINE configs; divisors outside [2^-20, 2^32] are refused; and an
    independent numpy emulation of the three operations agrees with the division on a million random numerators."""
    import ctypes as C
    import numpy as np
    lib = hb.lib()

    def check(c):
        rc, ok = C.c_float(), C.c_int()
        assert lib.rt_selftest_division(C.c_float(c), C.byref(rc), C.byref(ok)) == 0
        return bool(ok.value), np.float32(rc.value)

    pi, tau = np.float32(3.14159274101257324219), np.float32(6.28318548202514648438)
    for c in (pi, tau):
        ok, rc = check(float(c))
        assert ok and rc == np.float32(1.0) / c
    rng = np.random.default_rng(3)
    # (... and the sphere radii of the reference's scenes: the hit record's (p - c) / r goes through the radius' reciprocal,
    # rt_intersect.h make_sphere_hit_by_reciprocal; rtweekend1's two-sphere kernels require both to pass)
    for c in (1919.0, 1079.0, 399.0, 224.0, 4095.0, 100.0, 1.0, 2.0, 63.0, 35.0, float(pi), 0.5, 1000.0, 0.45, 0.4):
        ok, rc = check(c)
        assert ok, c  # (every divisor tried so far passes; a failing one would only cost speed: the kernels keep the plain division)
        c32 = np.float32(c)
        x = (rng.uniform(0.5, 2.0, 1 << 20) * 2.0 ** rng.integers(-40, 40, 1 << 20)).astype(np.float32)
        x[:3] = [0.0, np.float32(2.0 ** -60), np.float32(2.0 ** 60)]  # (-0 comes back as +0: no caller feeds one whose sign matters)
        q0 = (x * rc).astype(np.float32)
        e = (np.float64(x) - np.float64(c32) * np.float64(q0)).astype(np.float32)   # fma(-c, q0, x): the exact remainder fits an f32
        q = (np.float64(e) * np.float64(rc) + np.float64(q0)).astype(np.float32)
        assert np.array_equal(q.view(np.uint32), (x / c32).astype(np.float32).view(np.uint32)), c
    for c in (0.0, 1.0e-10, 1.0e12, float("nan"), float("inf")):
        assert not check(c)[0], c
