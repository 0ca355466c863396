"""the output stage right after the path (crates/output/src/lib.rs:74-113)."""
import os
import struct
import zlib

import numpy as np
import pytest

import scenes

abi = scenes.abi


def test_rgb8_conversion(hb):
    x = np.array([0.0, 1.0, 0.5, 0.25, 2.0, -0.1, np.nan, np.inf, 1e-9, 0.999999], dtype=np.float32)
    got = hb.output_rgb8(x, 2.2)
    want = [0, 255, 186, 136, 255, 0, 0, 255, 0, 255]  # (v^(1/2.2) * 255.999) as u8, saturating, NaN -> 0
    assert list(got) == want
    rng = np.random.default_rng(0)
    img = rng.uniform(0, 1.2, (37, 53, 3)).astype(np.float32)
    ref = np.clip(np.floor(img.astype(np.float64) ** (1 / 2.2) * 255.999), 0, 255).astype(np.uint8)
    out = hb.output_rgb8(img, 2.2)
    assert (out != ref).mean() < 1e-3 and np.abs(out.astype(int) - ref.astype(int)).max() <= 1  # f32 powf vs f64
    assert np.array_equal(hb.output_rgb8(img, 1.0), np.clip(np.floor(img * np.float32(255.999)), 0, 255).astype(np.uint8))


def _read_png(path):
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(data):
        n, = struct.unpack(">I", data[pos:pos + 4])
        typ = data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        crc, = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        assert crc == zlib.crc32(typ + body) & 0xFFFFFFFF
        chunks.append((typ, body))
        pos += 12 + n
    w, h, depth, colour = struct.unpack(">IIBB", chunks[0][1][:10])
    assert (depth, colour) == (8, 2) and chunks[-1][0] == b"IEND"
    raw = zlib.decompress(b"".join(b for t, b in chunks if t == b"IDAT"))
    rows = np.frombuffer(raw, dtype=np.uint8).reshape(h, w * 3 + 1)
    assert np.all(rows[:, 0] == 0)
    return rows[:, 1:].reshape(h, w, 3)


def test_png_and_ppm_round_trip(hb, tmp_path):
    rng = np.random.default_rng(1)
    for (h, w) in ((36, 64), (300, 250)):  # the second spans several 64 KiB stored-deflate blocks
        img = rng.uniform(0, 1, (h, w, 3)).astype(np.float32)
        want = hb.output_rgb8(img, 2.2)
        hb.save_image(str(tmp_path / "a.png"), img, 2.2)
        assert np.array_equal(_read_png(str(tmp_path / "a.png")), want)
        hb.save_image(str(tmp_path / "a.ppm"), img, 2.2)
        data = open(tmp_path / "a.ppm", "rb").read()
        header = f"P6\n{w} {h}\n255\n".encode()
        assert data.startswith(header) and np.array_equal(np.frombuffer(data[len(header):], np.uint8).reshape(h, w, 3), want)


def test_filename_rules(hb, tmp_path):
    img = np.zeros((2, 2, 3), dtype=np.float32)
    for bad, code in (("noext", abi.RT_ERR_INVALID_ARGUMENT), ("a.b.png", abi.RT_ERR_INVALID_ARGUMENT), ("x.exr", abi.RT_ERR_UNSUPPORTED)):
        with pytest.raises(hb.RtHipError) as e:
            hb.save_image(bad, img)  # like the reference: exactly one '.', dispatch on the extension
        assert e.value.code == code
