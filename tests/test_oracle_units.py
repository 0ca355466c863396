"""Deterministic unit tests of the oracle's building blocks -- the reference's own live unit tests
restated (utility/mod.rs:136-150, utility/coord.rs:33-50) plus hand-derivable values."""
import numpy as np


def test_sort_by_indices_reference_vector(O):
    # utility/mod.rs:140-149: indices [0,4,2,1,3] over [a,b,c,d,e] -> [a,e,c,b,d]
    got = O.sort_by_indices([10, 11, 12, 13, 14], [0, 4, 2, 1, 3])
    assert list(got) == [10, 14, 12, 11, 13]


def test_sort_by_indices_is_a_gather(O):
    rng = np.random.default_rng(0)
    for n in (1, 2, 7, 64, 1000):
        perm = rng.permutation(n)
        vals = rng.integers(0, 1 << 40, n)
        assert np.array_equal(O.sort_by_indices(vals, perm), vals[perm].astype(np.uint64))


def test_coordinate_inverse_round_trip(O):
    # utility/coord.rs:37-49, tolerance 1e-6 on the squared error
    rng = np.random.default_rng(1)
    for _ in range(200):
        z = rng.normal(size=3); z /= np.linalg.norm(z)
        v = rng.normal(size=3); v /= np.linalg.norm(v)
        fwd = O.coord_apply(z, v)
        back = O.coord_apply(z, fwd, inverse=True)
        assert np.sum((v - back) ** 2) < 1e-6
        inv_first = O.coord_apply(z, O.coord_apply(z, v, inverse=True))
        assert np.sum((v - inv_first) ** 2) < 1e-6


def test_next_previous_float(O):
    x = np.array([0.0, -0.0, 1.0, -1.0, 3.5e-45, np.inf, -np.inf, 1e30], dtype=np.float32)
    nxt = O.utility(0, x)
    prv = O.utility(1, x)
    assert nxt[0] == np.float32(1.4e-45) and nxt[1] == np.float32(1.4e-45)  # -0.0 is bumped as +0.0
    assert prv[0] == np.float32(-1.4e-45) and prv[1] == np.float32(-1.4e-45)
    assert nxt[2] == np.nextafter(np.float32(1), np.float32(2)) and prv[2] == np.nextafter(np.float32(1), np.float32(0))
    assert nxt[3] == np.nextafter(np.float32(-1), np.float32(0)) and prv[3] == np.nextafter(np.float32(-1), np.float32(-2))
    assert nxt[5] == np.inf and prv[6] == -np.inf
    finite = np.random.default_rng(2).normal(size=1000).astype(np.float32)
    assert np.array_equal(O.utility(0, finite), np.nextafter(finite, np.float32(np.inf)))
    assert np.array_equal(O.utility(1, finite), np.nextafter(finite, np.float32(-np.inf)))


def test_gamma(O):
    eps = np.float32(np.finfo(np.float32).eps)
    for n in (2, 3, 5, 6, 7):
        nm = np.float32(n) * np.float32(0.5) * eps
        assert O.utility(2, [float(n)])[0] == nm / (np.float32(1) - nm)


def test_offset_ray_moves_away_from_surface(O):
    o = O.offset_ray((1.0, 2.0, 3.0), (0.0, 0.0, 1.0), (3e-4, 3e-4, 3e-4), True)
    assert o[2] > 3.0 + 2.9e-4 and o[0] < 1.0 and o[1] < 2.0  # x,y: offset 0 -> previous_float
    o = O.offset_ray((1.0, 2.0, 3.0), (0.0, 0.0, 1.0), (3e-4, 3e-4, 3e-4), False)
    assert o[2] < 3.0 - 2.9e-4


def test_distribution1d_tables(O):
    vals = np.array([1.0, 3.0, 0.0, 4.0], dtype=np.float32)
    idx, pdf, cdf = O.dist1d(vals, 100000, seed=3)
    assert np.allclose(cdf, [0, 0.125, 0.5, 0.5, 1.0]) and np.allclose(pdf, [0.125, 0.375, 0.0, 0.5])
    assert np.array_equal(pdf, cdf[1:] - cdf[:-1])  # what the device recomputes instead of storing
    freq = np.bincount(idx, minlength=4) / idx.size
    assert freq[2] == 0.0 and np.abs(freq - pdf).max() < 5e-3


def test_distribution1d_all_zero_picks_last_cell(O):
    idx, pdf, cdf = O.dist1d(np.zeros(10, dtype=np.float32), 100)
    assert np.all(cdf == 0) and np.all(idx == 9)  # (first-1).clamp(0, len-2)


def test_camera_matches_hand_derivation(O):
    # rtweekend1.ssml: origin 0, lookat +y, vup +z, fov 121.28449..., focus 1 -> viewport 3.5556 x 2.0
    cam = O.camera_new((0, 0, 0), (0, 1, 0), (0, 0, 1), 121.28449291441745, 16 / 9, 0.0, 1.0)
    assert np.allclose(list(cam.horizontal), [-3.5555556, 0, 0], atol=1e-5)
    assert np.allclose(list(cam.vertical), [0, 0, 2.0], atol=1e-5)
    assert np.allclose(list(cam.lower_left), [1.7777778, 1.0, -1.0], atol=1e-5)
