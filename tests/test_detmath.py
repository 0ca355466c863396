"""The arithmetic contract (include/rt_detmath.h): random stream known answers and the accuracy of
the deterministic elementary functions against double-precision libm."""
import numpy as np


def ulp_error(got, x64_exact):
    got = got.astype(np.float64)
    ref32 = x64_exact.astype(np.float32)
    ulp = np.spacing(np.abs(ref32)).astype(np.float64)
    ulp = np.maximum(ulp, np.finfo(np.float32).tiny)
    return np.abs(got - x64_exact) / ulp


def test_philox_known_answers(O):
    # Random123 kat_vectors, philox4x32 10 rounds
    assert O.philox([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert O.philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert O.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def _xoshiro128pp(s, n):
    """Independent restatement of xoshiro128++ 1.0 (Blackman & Vigna)."""
    M = 0xFFFFFFFF
    rotl = lambda x, k: ((x << k) | (x >> (32 - k))) & M
    out = []
    for _ in range(n):
        out.append((rotl((s[0] + s[3]) & M, 7) + s[0]) & M)
        t = (s[1] << 9) & M
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]
        s[2] ^= t
        s[3] = rotl(s[3], 11)
    return out


def test_stream_is_philox_seeded_xoshiro(O):
    seed, pixel, sample = 0x1234567890ABCDEF, 77777, 5
    state = O.philox([pixel & 0xFFFFFFFF, pixel >> 32, sample & 0xFFFFFFFF, sample >> 32],
                     [seed & 0xFFFFFFFF, seed >> 32])
    want = _xoshiro128pp(list(state), 64)
    got = O.rng_u32(seed, pixel, sample, 64)
    assert [int(x) for x in got] == want


def test_float_conversions(O):
    u = O.rng_u32(9, 1, 2, 1000).astype(np.uint64)
    f = O.rng_f32(9, 1, 2, 1000)
    # rand 0.8 Standard f32: (u >> 8) * 2^-24
    assert np.array_equal(f, ((u >> 8).astype(np.float32) * np.float32(2.0 ** -24)))
    assert f.min() >= 0.0 and f.max() < 1.0


def test_streams_differ_by_pixel_and_sample(O):
    a = O.rng_u32(1, 10, 0, 16); b = O.rng_u32(1, 11, 0, 16); c = O.rng_u32(1, 10, 1, 16); d = O.rng_u32(2, 10, 0, 16)
    assert len({a.tobytes(), b.tobytes(), c.tobytes(), d.tobytes()}) == 4


def test_uniformity_of_stream(O):
    from scipy import stats
    f = np.concatenate([O.rng_f32(3, p, 0, 2000) for p in range(200)])
    hist, _ = np.histogram(f, bins=64, range=(0, 1))
    assert stats.chisquare(hist).pvalue > 1e-4


def test_rng_below_is_uniform_and_in_range(O):
    from scipy import stats
    for bound in (1, 2, 3, 7, 100):
        x = O.rng_below(5, bound, 70000)
        assert x.max() < bound
        if bound > 1:
            assert stats.chisquare(np.bincount(x, minlength=bound)).pvalue > 1e-4


def test_sin_cos_accuracy(O):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-2 * np.pi, 4 * np.pi, 400000), rng.uniform(-200, 200, 100000),
                        np.linspace(0, 2 * np.pi, 10001)]).astype(np.float32)
    x64 = x.astype(np.float64)
    es = ulp_error(O.detmath(0, x), np.sin(x64))
    ec = ulp_error(O.detmath(1, x), np.cos(x64))
    # away from the zeros of the result the error is ~1 ulp; near zeros it is bounded absolutely
    big_s = np.abs(np.sin(x64)) > 1e-3
    big_c = np.abs(np.cos(x64)) > 1e-3
    assert es[big_s].max() <= 2.0, es[big_s].max()
    assert ec[big_c].max() <= 2.0, ec[big_c].max()
    assert np.abs(O.detmath(0, x) - np.sin(x64)).max() < 2e-7
    assert np.abs(O.detmath(1, x) - np.cos(x64)).max() < 2e-7


def test_sin_cos_special_values(O):
    assert O.detmath(0, [0.0])[0] == 0.0 and O.detmath(1, [0.0])[0] == 1.0
    assert np.isnan(O.detmath(0, [np.inf])[0]) and np.isnan(O.detmath(1, [np.nan])[0])
    # large arguments are folded in double arithmetic: still in range and close to libm
    big = np.array([5e6, -1e8, 1e10, -3e12], dtype=np.float32)
    assert np.abs(O.detmath(0, big) - np.sin(big.astype(np.float64))).max() < 1e-3
    assert np.abs(O.detmath(1, big) - np.cos(big.astype(np.float64))).max() < 1e-3
    # from 2^50 on the contract says NaN (no Payne-Hanek reduction)
    assert np.all(np.isnan(O.detmath(0, np.array([2e15, -3e20, 3e38], dtype=np.float32))))


def test_acos_accuracy(O):
    x = np.concatenate([np.linspace(-1, 1, 200001), np.random.default_rng(1).uniform(-1, 1, 200000)]).astype(np.float32)
    e = ulp_error(O.detmath(2, x), np.arccos(x.astype(np.float64)))
    assert e.max() <= 2.5, e.max()
    assert O.detmath(2, [1.0])[0] == 0.0
    assert np.isnan(O.detmath(2, [1.5])[0]) and np.isnan(O.detmath(2, [-1.5])[0])


def test_atan2_accuracy_and_quadrants(O):
    rng = np.random.default_rng(2)
    y = rng.normal(size=400000).astype(np.float32)
    x = rng.normal(size=400000).astype(np.float32)
    got = O.detmath(3, y, x)
    ref = np.arctan2(y.astype(np.float64), x.astype(np.float64))
    assert ulp_error(got, ref).max() <= 4.0
    assert O.detmath(3, [0.0], [1.0])[0] == 0.0
    assert abs(O.detmath(3, [0.0], [-1.0])[0] - np.pi) < 1e-6
    assert abs(O.detmath(3, [1.0], [0.0])[0] - np.pi / 2) < 1e-6
    assert abs(O.detmath(3, [-1.0], [0.0])[0] + np.pi / 2) < 1e-6
    assert abs(O.detmath(3, [np.inf], [np.inf])[0] - np.pi / 4) < 1e-6


def test_tan_pow5(O):
    x = np.linspace(-1.4, 1.4, 10001).astype(np.float32)
    assert ulp_error(O.detmath(4, x), np.tan(x.astype(np.float64))).max() <= 4.0
    x = np.linspace(0, 1, 10001).astype(np.float32)
    assert ulp_error(O.detmath(5, x), x.astype(np.float64) ** 5).max() <= 3.0


def test_powf_against_float64(O):
    """rt_powf (the output stage's val.powf(1/gamma), crates/output/src/lib.rs:92-95): within 1 ulp of the correctly
    rounded result over colour-like bases and display gammas, exact on the special cases of powf."""
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(0, 4, 200000), 10.0 ** rng.uniform(-38, 38, 50000), rng.uniform(0, 1e-40, 1000)]).astype(np.float32)
    for y in (1 / 2.2, 1 / 2.4, 0.5, 1.0, 2.2, 3.0, -0.75):
        yy = np.full_like(x, np.float32(y))
        got = O.detmath(6, x, yy)
        with np.errstate(over="ignore", divide="ignore"):
            exact = x.astype(np.float64) ** np.float64(np.float32(y))
        fin = np.isfinite(exact) & (exact < 3e38) & (exact > 1e-37)
        assert ulp_error(got[fin], exact[fin]).max() <= 1.0
    sp_x = np.array([0.0, 0.0, 1.0, np.inf, np.inf, -1.0, np.nan, 2.0, 0.5, 2.0, 0.5, 5.0], dtype=np.float32)
    sp_y = np.array([0.5, -0.5, np.nan, 0.5, -0.5, 0.5, 0.5, np.inf, np.inf, -np.inf, -np.inf, 0.0], dtype=np.float32)
    want = np.array([0.0, np.inf, 1.0, np.inf, 0.0, np.nan, np.nan, np.inf, 0.0, 0.0, np.inf, 1.0], dtype=np.float32)
    got = O.detmath(6, sp_x, sp_y)
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(np.nan_to_num(got, nan=7.0), np.nan_to_num(want, nan=7.0))


def test_powf_negative_bases_follow_powf(O):
    """f32::powf is defined for a negative base when the exponent is an integer (the sign is the exponent's parity) and NaN
    otherwise; 1 / gamma IS an integer for gamma = 1, 0.5, 0.25.  rt_powf against glibc's powf (through numpy float32)."""
    rng = np.random.default_rng(6)
    x = -np.concatenate([rng.uniform(0, 4, 20000), 10.0 ** rng.uniform(-10, 10, 5000)]).astype(np.float32)
    for y in (1.0, 2.0, 3.0, 4.0, -1.0, -2.0, 7.0, 16777216.0, 0.5, 1 / 2.2, 2.5):
        got = O.detmath(6, x, np.full_like(x, np.float32(y)))
        with np.errstate(all="ignore"):
            exact = np.power(x.astype(np.float64), np.float64(np.float32(y)))
        if float(y).is_integer():
            fin = np.isfinite(exact) & (np.abs(exact) < 3e38) & (np.abs(exact) > 1e-37)
            if fin.any():
                assert ulp_error(got[fin], exact[fin]).max() <= 1.0, y
            assert not np.isnan(got).any() and np.array_equal(np.signbit(got), np.signbit(exact)), y
        else:
            assert np.all(np.isnan(got)), y
    sp_x = np.array([-0.0, -0.0, -0.0, -0.0, -np.inf, -np.inf, -np.inf, -1.0, -1.0, -2.0, -0.5], dtype=np.float32)
    sp_y = np.array([3.0, 2.0, -3.0, -2.0, 3.0, 2.0, -3.0, np.inf, -np.inf, np.inf, np.inf], dtype=np.float32)
    with np.errstate(all="ignore"):
        want = np.power(sp_x, sp_y)  # glibc powf
    got = O.detmath(6, sp_x, sp_y)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (got, want)


def test_quantise_u8_against_the_platform_powf(O):
    """The reference's output stage calls the PLATFORM's powf (Rust std -> libm); the contract's rt_powf is within 1 ulp of it,
    so an 8-bit value can differ by one at a bucket edge.  Measured here, not defined away: over a dense sweep of the colour
    range and the usual gammas the contract's byte equals glibc's byte except on a stated, tiny fraction of inputs, and never by
    more than one."""
    x = np.concatenate([np.linspace(0.0, 1.0, 2000001), np.linspace(1.0, 4.0, 300001), -np.linspace(0.0, 2.0, 100001)]).astype(np.float32)
    worst = 0.0
    for gamma in (2.2, 2.4, 1.8, 1.0, 0.5):
        got = O.output_rgb8(x, gamma).astype(np.int32)
        with np.errstate(all="ignore"):
            v = np.power(x, np.float32(1.0) / np.float32(gamma)) * np.float32(255.999)   # glibc powf in f32
        want = np.where(np.isnan(v) | (v <= 0), 0, np.where(v >= 255, 255, v.astype(np.int32))).astype(np.int32)
        diff = np.abs(got - want)
        assert diff.max() <= 1, gamma
        worst = max(worst, float((diff != 0).mean()))
    assert worst < 2e-5, worst  # about one input in 10^5 sits on a bucket edge where 1 ulp of powf decides the byte
