"""Which RCCL a multi-device scene (rt_scene_create_multi) would bind, and how it degrades -- the part of the multi-GPU path that
can run without a GPU (rt_rccl_probe touches none).  The look-up order (csrc/rt_api.cpp load_rccl): RT_HIP_RCCL_LIB, a copy the
process already carries (RTLD_NOLOAD: PyTorch's), the system's; a library that cannot be loaded or lacks one of the six entry
points is unusable and the gather falls back to peer copies (the GPU side of that: tests/test_gpu_parity.py
::test_multi_device_gather_transports)."""
import os
import subprocess

import pytest

STUB = """
int ncclCommInitAll(void **c, int n, const int *d) { return 5; }
int ncclCommDestroy(void *c) { return 0; }
int ncclGroupStart(void) { return 0; }
int ncclGroupEnd(void) { return 0; }
int ncclSend(const void *b, unsigned long n, int t, int p, void *c, void *s) { return 0; }
#ifndef NO_RECV
int ncclRecv(void *b, unsigned long n, int t, int p, void *c, void *s) { return 0; }
#endif
"""


@pytest.fixture
def clean_env():
    saved = {k: os.environ.pop(k, None) for k in ("RT_HIP_RCCL_LIB", "RT_HIP_NO_RCCL")}
    yield
    for k, v in saved.items():
        os.environ.pop(k, None)
        if v is not None:
            os.environ[k] = v


def _stub(tmp_path, name, flags=()):
    src = tmp_path / (name + ".c")
    src.write_text(STUB)
    out = str(tmp_path / (name + ".so"))
    subprocess.run(["gcc", "-shared", "-fPIC", *flags, str(src), "-o", out], check=True)
    return out


def test_rccl_lookup_and_degradation(hb, tmp_path, clean_env):
    usable, note = hb.rccl_probe()  # this image ships ROCm's librccl; a process that imported torch already carries it
    if usable:
        assert "librccl" in note and ("already loaded" in note or "on demand" in note), note
    else:
        assert "not found" in note or "lacks" in note, note
    os.environ["RT_HIP_RCCL_LIB"] = str(tmp_path / "absent.so")
    usable, note = hb.rccl_probe()
    assert not usable and "cannot be loaded" in note, note
    os.environ["RT_HIP_RCCL_LIB"] = _stub(tmp_path, "no_recv", ("-DNO_RECV",))
    usable, note = hb.rccl_probe()
    assert not usable and "lacks" in note and "ncclRecv" in note, note
    os.environ["RT_HIP_RCCL_LIB"] = _stub(tmp_path, "whole")
    usable, note = hb.rccl_probe()
    assert usable and "RT_HIP_RCCL_LIB" in note, note
    os.environ["RT_HIP_NO_RCCL"] = "1"
    usable, note = hb.rccl_probe()
    assert not usable and "RT_HIP_NO_RCCL" in note, note
