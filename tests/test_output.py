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


def test_rgb8_host_conversion_equals_the_oracle_twin(hb, O):
    """f3: the product's host conversion and the oracle's restatement of lib.rs:92-95 share the contract's rt_powf, so
    they must agree byte for byte (special values included); the GPU twin is checked in test_gpu_parity.py."""
    rng = np.random.default_rng(3)
    img = np.concatenate([rng.uniform(-0.2, 1.5, 300000), 10.0 ** rng.uniform(-30, 30, 5000),
                          [np.nan, np.inf, -np.inf, 0.0, -0.0, 1.0, 255.0 / 255.999]]).astype(np.float32)
    for gamma in (2.2, 1.0, 2.4, 0.8):
        assert np.array_equal(hb.output_rgb8(img, gamma), O.output_rgb8(img, gamma))


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


def _read_exr(path):
    """minimal OpenEXR reader: single-part scanline file, no compression, FLOAT channels"""
    d = open(path, "rb").read()
    assert d[:4] == b"\x76\x2f\x31\x01" and struct.unpack("<I", d[4:8])[0] == 2
    pos, attrs = 8, {}
    while d[pos] != 0:
        e = d.index(b"\0", pos); name = d[pos:e].decode(); pos = e + 1
        e = d.index(b"\0", pos); typ = d[pos:e].decode(); pos = e + 1
        size, = struct.unpack("<i", d[pos:pos + 4]); pos += 4
        attrs[name] = (typ, d[pos:pos + size]); pos += size
    pos += 1
    for required in ("channels", "compression", "dataWindow", "displayWindow", "lineOrder", "pixelAspectRatio",
                     "screenWindowCenter", "screenWindowWidth"):
        assert required in attrs, required
    assert attrs["compression"] == ("compression", b"\0") and attrs["lineOrder"] == ("lineOrder", b"\0")
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    names, c, p = [], attrs["channels"][1], 0
    while c[p] != 0:
        e = c.index(b"\0", p); names.append(c[p:e].decode()); p = e + 1
        assert struct.unpack("<iB3xii", c[p:p + 16]) == (2, 0, 1, 1); p += 16
    assert names == sorted(names) == ["B", "G", "R"]
    offsets = struct.unpack(f"<{h}Q", d[pos:pos + 8 * h])
    img = np.zeros((h, w, 3), dtype=np.float32)
    for y, off in enumerate(offsets):
        yy, size = struct.unpack("<ii", d[off:off + 8])
        assert yy == y and size == w * 12
        planes = np.frombuffer(d[off + 8:off + 8 + size], dtype="<f4").reshape(3, w)
        img[y, :, 2], img[y, :, 1], img[y, :, 0] = planes[0], planes[1], planes[2]
    assert offsets[-1] + 8 + w * 12 == len(d)
    return img


def test_exr_bmp_tiff(hb, tmp_path):
    rng = np.random.default_rng(2)
    img = rng.uniform(0, 4, (37, 53, 3)).astype(np.float32)  # odd width: BMP row padding
    img[3, 5] = (np.inf, 1e-30, 0.0)
    hb.save_image(str(tmp_path / "a.exr"), img, 2.2)
    assert _read_exr(str(tmp_path / "a.exr")).tobytes() == img.tobytes()  # gamma ignored, floats untouched
    want = hb.output_rgb8(img, 2.2)
    Image = pytest.importorskip("PIL.Image")
    for ext in ("bmp", "tiff", "png"):
        hb.save_image(str(tmp_path / f"a.{ext}"), img, 2.2)
        with Image.open(tmp_path / f"a.{ext}") as im:
            assert im.mode == "RGB" and im.size == (53, 37)
            assert np.array_equal(np.asarray(im), want), ext


def test_filename_rules(hb, tmp_path):
    img = np.zeros((2, 2, 3), dtype=np.float32)
    for bad, code in (("noext", abi.RT_ERR_INVALID_ARGUMENT), ("a.b.png", abi.RT_ERR_INVALID_ARGUMENT), ("x.jpg", abi.RT_ERR_UNSUPPORTED), ("x.gif", abi.RT_ERR_UNSUPPORTED)):
        with pytest.raises(hb.RtHipError) as e:
            hb.save_image(bad, img)  # like the reference: exactly one '.', dispatch on the extension
        assert e.value.code == code


def test_final_statistics_text(hb):
    """output::get_readable_duration / print_final_statistics (crates/output/src/lib.rs:33-63,115-124)"""
    d = hb.get_readable_duration
    assert d(0.4) == "~0 seconds" and d(1) == "1 second" and d(59.9) == "59 seconds"
    assert d(60) == "1 minute, ~0 seconds" and d(3600 + 2 * 60 + 1) == "1 hour, 2 minutes, 1 second"
    assert d(2 * 86400 + 3 * 3600 + 7) == "2 days, 3 hours, 7 seconds"
    text = hb.final_statistics(2.0, 3_000_000, 1024)
    assert text == "Finished rendering:\n\tSamples:\t1024\n\tTime taken:\t2 seconds\n\tRays shot:\t3000000 @ 1.50 Mray/s"
