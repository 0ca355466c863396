"""Pins of the oracle's integrators and sampler.

The analytic targets come from the reference's own (commented-out) integration tests:
furnace = (0.25, 0.25, 0.25) +- 1e-3 for naive, MIS and MIS+sky sampling
(crates/implementations/tests/sampling.rs:239-297) and MIS mean == naive mean (:181-207).
The rest are hand-derivable values and the committed golden fixtures."""
import json
import os

import numpy as np
import pytest

import scenes

abi = scenes.abi


@pytest.mark.parametrize("method", [abi.RT_METHOD_NAIVE, abi.RT_METHOD_MIS])
@pytest.mark.parametrize("sampler_res", [(0, 0), (10, 10)])
def test_furnace(O, method, sampler_res):
    s = O.Scene(scenes.furnace(sampler_res))
    mean = s.integrate_ray((0, 0, 3), (0, 0, -1), method, 1_500_000, seed=7)
    assert np.linalg.norm(mean - 0.25) < 1e-3, mean


@pytest.mark.parametrize("sampler_res", [(0, 0), (20, 10)])
def test_mis_equals_naive(O, sampler_res):
    s = O.Scene(scenes.lit_plane(sampler_res))
    naive = s.integrate_ray((0, 0, 3), (0, 0, -1), abi.RT_METHOD_NAIVE, 6_000_000, seed=3)
    mis = s.integrate_ray((0, 0, 3), (0, 0, -1), abi.RT_METHOD_MIS, 1_500_000, seed=4)
    assert np.linalg.norm(naive - mis) < 4e-3 * max(1.0, np.linalg.norm(naive)), (naive, mis)


@pytest.mark.parametrize("method", [abi.RT_METHOD_NAIVE, abi.RT_METHOD_MIS])
@pytest.mark.parametrize("sampler_res", [(0, 0), (12, 6)])
def test_sphere_light_over_a_floor_closed_form(O, method, sampler_res):
    """Irradiance of a Lambertian floor point straight below a spherical source of radiance L, radius R, centre
    height d is pi L (R/d)^2, so the point's radiance is rho L (R/d)^2 -- one bounce, no interreflection (an
    emitter ends the path).  Checks cone sampling of the light, its pdf, the MIS weights and the naive estimator
    against a number neither of them was tuned to."""
    rho, albedo, colour, strength, d, r = np.array([0.8, 0.6, 0.4]), 0.9, np.array([1.0, 0.7, 0.4]), 6.0, 3.0, 1.0
    s = O.Scene(scenes.floor_under(light=(d, r, tuple(colour), strength), sampler_res=sampler_res, rho=tuple(rho), albedo=albedo))
    want = rho * albedo * colour * strength * (r / d) ** 2
    n = 4_000_000 if method == abi.RT_METHOD_NAIVE else 1_000_000
    got = s.integrate_ray((2.0, 0.0, 1.5), (-2.0, 0.0, -1.5), method, n, seed=11)
    assert np.abs(got - want).max() < 4e-3, (got, want)


@pytest.mark.parametrize("method,sampler_res", [(abi.RT_METHOD_NAIVE, (0, 0)), (abi.RT_METHOD_MIS, (0, 0)),
                                                (abi.RT_METHOD_MIS, (50, 25)), (abi.RT_METHOD_MIS, (7, 3))])
def test_lerp_sky_over_a_floor_closed_form(O, method, sampler_res):
    """Under the Lerp sky L(w) = c1 t + c2 (1 - t), t = (w.z + 1)/2 (textures/mod.rs:283-291) a floor with normal
    +z receives E = pi (c1 + c2)/2 + pi (c1 - c2)/3, so its radiance is rho [(c1 + c2)/2 + (c1 - c2)/3]: checks
    the sky's importance sampling (table build, sample, pdf) and MIS against calculus."""
    rho, albedo = np.array([0.8, 0.6, 0.4]), 0.9
    c1, c2 = np.array([0.5, 0.7, 1.0]), np.array([1.0, 0.9, 0.2])
    s = O.Scene(scenes.floor_under(sky=(tuple(c1), tuple(c2)), sampler_res=sampler_res, rho=tuple(rho), albedo=albedo))
    want = rho * albedo * ((c1 + c2) / 2 + (c1 - c2) / 3)
    got = s.integrate_ray((2.0, 0.0, 1.5), (-2.0, 0.0, -1.5), method, 1_500_000, seed=12)
    assert np.abs(got - want).max() < 2e-3, (got, want)


def test_mirror_floor_shows_the_sky(O):
    """Reflect with fuzz 0 (materials/reflect.rs:22-43) under the naive integrator: the pixel is tex * Lerp(reflected
    direction), a deterministic value -- every sample identical, equal to the formula to float rounding."""
    sc = scenes.SceneDescription()
    tex = np.array([0.9, 0.8, 0.7])
    mirror = sc.reflect(sc.solid(tuple(tex)), 0.0)
    n = (0.0, 0.0, 1.0)
    a, b, c, d = (-500.0, -500.0, 0.0), (500.0, 500.0, 0.0), (-500.0, 500.0, 0.0), (500.0, -500.0, 0.0)
    sc.triangle([a, b, c], [n, n, n], mirror)
    sc.triangle([a, b, d], [n, n, n], mirror)
    c1, c2 = np.array([0.5, 0.7, 1.0]), np.array([1.0, 0.9, 0.2])
    sc.set_sky(sc.lerp(tuple(c1), tuple(c2)), (0, 0))
    s = O.Scene(sc)
    got = s.integrate_ray((2.0, 0.0, 1.5), (-2.0, 0.0, -1.5), abi.RT_METHOD_NAIVE, 1000, seed=5)
    t = 0.6 * 0.5 + 0.5  # reflected direction (-0.8, 0, 0.6)
    want = tex * (c1 * t + c2 * (1 - t))
    assert np.abs(got - want).max() < 2e-6, (got, want)


def test_glass_sphere_in_a_white_sky_is_white(O):
    """Refract (materials/refract.rs:23-61, white texture) neither absorbs nor emits, so under a uniform sky of
    radiance 1 every naive path that escapes carries exactly 1: the mean is 1 up to the rare path cut at
    MAX_DEPTH inside the sphere -- Schlick reflection vs refraction, both branches, conserve energy."""
    sc = scenes.SceneDescription()
    sc.sphere((0.0, 0.0, 0.0), 1.0, sc.refract(sc.solid((1.0, 1.0, 1.0)), 1.5))
    sc.set_sky(sc.solid((1.0, 1.0, 1.0)), (0, 0))
    s = O.Scene(sc)
    for origin, direction in (((0.0, 0.0, 4.0), (0.0, 0.0, -1.0)), ((0.7, 0.2, 4.0), (0.0, 0.0, -1.0)), ((0.0, 0.95, 4.0), (0.0, 0.0, -1.0))):
        got = s.integrate_ray(origin, direction, abi.RT_METHOD_NAIVE, 200_000, seed=6)
        assert np.abs(got - 1.0).max() < 2e-3, (origin, got)


def _f32(x):
    return np.float32(x)


def test_primary_sky_pixels_are_the_lerp_colour(O):
    """rtweekend1 at 1 spp: a camera ray that misses everything returns Emit(1.0) * Lerp(direction)
    (mis.rs:23-31, textures/mod.rs:284-287).  Recompute those pixels in numpy f32 from the stream."""
    ls = scenes.load_ssml("rtweekend1")
    s = O.Scene(ls.scene)
    cam = O.camera_new(**ls.camera_params)
    W, H = 64, 36
    opts = abi.default_render_opts(W, H, 1, seed=5)
    img, rays = s.render(cam, opts, n_threads=2)
    o = np.float32(list(cam.origin)); ll = np.float32(list(cam.lower_left))
    hz = np.float32(list(cam.horizontal)); vt = np.float32(list(cam.vertical))
    c1, c2 = np.float32([0.5, 0.7, 1.0]), np.float32([1.0, 1.0, 1.0])
    checked = 0
    for y in range(0, 6):  # rows above the small sphere (its top is at row ~7): every ray misses
        for x in range(0, W, 7):
            pix = y * W + x
            u32 = O.rng_u32(5, pix, 0, 2)
            r = ((u32 >> 9) | 0x3F800000).astype(np.uint32).view(np.float32) - _f32(1.0)
            u = (r[0] + _f32(x)) / _f32(W - 1)
            v = _f32(1.0) - (r[1] + _f32(y)) / _f32(H - 1)
            d = ll + hz * u + vt * v - o
            d = d / np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2], dtype=np.float32)
            t = d[2] * _f32(0.5) + _f32(0.5)
            want = c1 * t + c2 * (_f32(1.0) - t)
            assert np.array_equal(img[y, x], want), (x, y, img[y, x], want)
            checked += 1
    assert checked >= 60


def test_golden_images(O, golden_dir):
    meta = json.load(open(os.path.join(golden_dir, "golden_meta.json")))
    cases = {
        "rtweekend1": (scenes.load_ssml("rtweekend1").scene, scenes.load_ssml("rtweekend1").camera_params, 16),
        "overshadowed": (scenes.load_ssml("overshadowed").scene, scenes.load_ssml("overshadowed").camera_params, 16),
        "pyramid": (scenes.load_ssml("pyramid").scene, scenes.load_ssml("pyramid").camera_params, 16),
        "all_materials": (scenes.all_materials(), scenes.ALL_MATERIALS_CAMERA, 8),
        "mesh2000": (scenes.random_triangle_mesh(2000, seed=42, extent=3.0, edge=0.5, emissive_every=100, sampler_res=(20, 10)),
                     scenes.MESH_CAMERA | {"origin": (0.0, -9.0, 0.0)}, 8),
        "structured": (scenes.structured_meshes(3, 32, (20, 10)), scenes.STRUCTURED_CAMERA, 8),
    }
    for name, (sc, cam_params, spp) in cases.items():
        s = O.Scene(sc)
        cam = O.camera_new(**cam_params)
        for method, mname in ((abi.RT_METHOD_NAIVE, "naive"), (abi.RT_METHOD_MIS, "mis")):
            img, rays = s.render(cam, abi.default_render_opts(64, 36, spp, method=method, seed=1), n_threads=3)
            want = np.load(os.path.join(golden_dir, f"{name}_64x36_s{spp}_{mname}.npy"))
            assert np.array_equal(img, want), (name, mname)
            assert rays == meta[f"{name}_{mname}"]["rays_shot"]
        rays_in = np.load(os.path.join(golden_dir, f"{name}_rays.npy"))
        hits = s.check_hit(rays_in[:, :3], rays_in[:, 3:])
        assert hits.tobytes() == np.load(os.path.join(golden_dir, f"{name}_hits.npy")).tobytes()


def test_thread_count_does_not_change_pixels(O):
    ls = scenes.load_ssml("overshadowed")
    s = O.Scene(ls.scene); cam = O.camera_new(**ls.camera_params)
    opts = abi.default_render_opts(48, 27, 4)
    a, ra = s.render(cam, opts, n_threads=1)
    b, rb = s.render(cam, opts, n_threads=7)
    assert np.array_equal(a, b) and ra == rb


def test_overshadowed_mis_zeroes_nan_samples(O):
    """P-hazard 3: black sky with the default 100x100 sampler => sky pdf 0 => 0/0 in the
    light-sample term whenever the 50 % sky pick is unoccluded => the whole sample is zeroed by
    the NaN filter (mis.rs:39-43,88-90).  At 1 spp about half of the floor pixels are therefore
    exactly black; with sampler_res (0,0) the sky is never picked and they are lit.  The light pick
    is weighted 1/0.5, so the MEAN stays that of the naive estimator on this (nearly convex) scene."""
    ls = scenes.load_ssml("overshadowed")
    s = O.Scene(ls.scene); cam = O.camera_new(**ls.camera_params)
    opts = abi.default_render_opts(96, 54, 1, method=abi.RT_METHOD_MIS)
    img, _ = s.render(cam, opts)
    assert np.isfinite(img).all()
    floor = img[40:54, :, 0]
    zero_frac_default = (floor == 0).mean()
    origin, direction = (-5.0, 3.0, -3.0), (6.5, -3.0, 2.0)  # hits the floor at (1.5, 0, -1), next to the light
    naive = s.integrate_ray(origin, direction, abi.RT_METHOD_NAIVE, 3_000_000, seed=1)[0]
    mis = s.integrate_ray(origin, direction, abi.RT_METHOD_MIS, 1_000_000, seed=2)[0]
    assert abs(mis - naive) < 0.015 * naive, (mis, naive)

    ls.scene.set_sky(ls.scene.sky[0], (0, 0), ls.scene.sky[1])
    s2 = O.Scene(ls.scene)
    img2, _ = s2.render(cam, opts)
    zero_frac_unsampled = (img2[40:54, :, 0] == 0).mean()
    assert 0.35 < zero_frac_default < 0.65 and zero_frac_unsampled < 0.05, (zero_frac_default, zero_frac_unsampled)


def test_shards_tile_the_frame(O):
    ls = scenes.load_ssml("rtweekend1")
    s = O.Scene(ls.scene); cam = O.camera_new(**ls.camera_params)
    full, rays_full = s.render(cam, abi.default_render_opts(50, 30, 3))
    acc = np.zeros_like(full); total = 0
    for k in range(3):
        o = abi.default_render_opts(50, 30, 3)
        o.shard_index, o.shard_count = k, 3
        part, r = s.render(cam, o)
        assert np.all((part == 0) | (acc == 0))  # shards are disjoint
        acc += part; total += r
    assert np.array_equal(acc, full) and total == rays_full


def test_sample_window_continues_the_stream(O):
    """sample_begin lets a caller render in batches (progressive display / resume): two half
    batches average to the full render up to the rounding of the running mean."""
    ls = scenes.load_ssml("rtweekend1")
    s = O.Scene(ls.scene); cam = O.camera_new(**ls.camera_params)
    full, _ = s.render(cam, abi.default_render_opts(40, 24, 8))
    o1 = abi.default_render_opts(40, 24, 4)
    o2 = abi.default_render_opts(40, 24, 4); o2.sample_begin = 4
    a, _ = s.render(cam, o1); b, _ = s.render(cam, o2)
    assert np.abs((a + b) / 2 - full).max() < 1e-6
    assert not np.array_equal(a, b)


def test_max_depth_is_a_parameter(O):
    """MAX_DEPTH is a const in the reference (integrators/mod.rs:7); the boundary exposes it."""
    ls = scenes.load_ssml("rtweekend1")
    s = O.Scene(ls.scene); cam = O.camera_new(**ls.camera_params)
    o2 = abi.default_render_opts(40, 24, 8); o2.max_depth = 2
    a, ra = s.render(cam, o2)
    b, rb = s.render(cam, abi.default_render_opts(40, 24, 8))
    assert ra < rb and not np.array_equal(a, b)
    assert a.mean() < b.mean()  # the truncated paths lose the multiply-scattered sky light
    o50 = abi.default_render_opts(40, 24, 8); o50.max_depth = 50
    c, rc = s.render(cam, o50)
    assert np.array_equal(b, c) and rb == rc  # 50 is the default


def test_sample_split_is_a_reordering_of_the_same_samples(O):
    """rt_render_opts.sample_split: S chunks per pixel, each summed on its own in pass order, the sums added in chunk order
    and divided by spp once.  S = 1 is the reference's strictly sequential running mean (golden images); any S
    uses the same samples and streams, so the image moves only by float rounding."""
    ls = scenes.load_ssml("overshadowed")
    s = O.Scene(ls.scene); cam = O.camera_new(**ls.camera_params)
    base, rays = s.render(cam, abi.default_render_opts(48, 27, 10))
    for S in (1, 2, 3, 10):
        o = abi.default_render_opts(48, 27, 10); o.sample_split = S
        img, r = s.render(cam, o, n_threads=3)
        assert r == rays and np.abs(img - base).max() < 2e-6
        assert np.array_equal(img, base) == (S == 1)
    # hand check of the definition on one pixel: chunks [0,3) [3,6) [6,10) for S = 3, spp = 10
    per_pass = []
    for k in range(10):
        o = abi.default_render_opts(48, 27, 1); o.sample_begin = k
        per_pass.append(s.render(cam, o)[0][20, 30].astype(np.float32))
    want = np.zeros(3, np.float32)
    for (b, e) in ((0, 3), (3, 6), (6, 10)):
        m = np.zeros(3, np.float32)
        for k in range(b, e):
            m = m + per_pass[k]   # a chunk's passes are summed in pass order ...
        want = want + m           # ... the chunk sums in chunk order ...
    want = want / np.float32(10)  # ... and the total is divided by spp once
    o = abi.default_render_opts(48, 27, 10); o.sample_split = 3
    assert np.array_equal(s.render(cam, o)[0][20, 30], want)
    o.sample_split = 11
    with pytest.raises(O.OracleError):
        s.render(cam, o)
