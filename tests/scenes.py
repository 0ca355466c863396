"""Scene builders shared by the tests (all through the public SceneDescription builder)."""
import importlib
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pkg = importlib.import_module("raytracing-rust_amd")
abi = pkg.abi
SceneDescription = pkg.SceneDescription


def load_ssml(name):
    return pkg.ssml.load_file(os.path.join(ROOT, "tests", "golden", "scenes", name + ".ssml"))


def furnace(sampler_res):
    """furnace_test() of the reference's (commented-out) crates/implementations/tests/sampling.rs:31-63."""
    sc = SceneDescription()
    light = sc.emissive(sc.solid((1, 1, 1)), 1.0)
    mat = sc.lambertian(sc.solid((0.5, 0.5, 0.5)), 0.5)
    hidden = sc.emissive(sc.solid((1, 0, 1)), 15.0)
    sc.sphere((0, 0, 0), 0.5, mat)
    sc.sphere((0, 0, 0), 1000.0, light)
    sc.sphere((0, 0, -5), 0.45, hidden)  # hidden light: must contribute nothing
    sc.set_sky(sc.lerp((0, 0, 0), (0.5, 1.0, 0.2)), sampler_res)
    return sc


def lit_plane(sampler_res=(0, 0), sky_colour=(0.3, 0.4, 0.6)):
    """A diffuse floor of two big triangles under two emissive spheres: the shape of bxdf_testing()
    (tests/sampling.rs:95-178) with a constant sky in place of its (unshipped) image sky."""
    sc = SceneDescription()
    diffuse = sc.lambertian(sc.solid((0.5, 0.5, 0.5)), 0.5)
    a, b = (-500.0, -500.0, -40.0), (500.0, 500.0, -40.0)
    c, d = (-500.0, 500.0, -40.0), (500.0, -500.0, -40.0)
    n = (0.0, 0.0, 1.0)
    sc.triangle([a, b, c], [n, n, n], diffuse)
    sc.triangle([a, b, d], [n, n, n], diffuse)
    sc.sphere((0, -100.5, 0), 50.0, sc.emissive(sc.solid((0, 1, 0)), 5.5))
    sc.sphere((0, 0, 300), 50.0, sc.emissive(sc.solid((1, 1, 1)), 10.5))
    sc.set_sky(sc.solid(sky_colour), sampler_res)
    return sc


def random_spheres(n, seed=0, split_type=abi.RT_SPLIT_SAH, emissive_every=0, sampler_res=(16, 8)):
    rng = np.random.default_rng(seed)
    sc = SceneDescription(split_type)
    mats = [sc.lambertian(sc.solid(rng.uniform(0.2, 0.9, 3)), 0.8) for _ in range(4)]
    light = sc.emissive(sc.solid((1.0, 0.9, 0.8)), 4.0)
    for i in range(n):
        c = rng.uniform(-10, 10, 3)
        r = rng.uniform(0.2, 1.2)
        m = light if (emissive_every and i % emissive_every == 0) else mats[i % 4]
        sc.sphere(c, r, m)
    sc.set_sky(sc.lerp((0.5, 0.7, 1.0), (1.0, 1.0, 1.0)), sampler_res)
    return sc


def skewed_chain_of_spheres(n=64, ratio=1.35, split_type=abi.RT_SPLIT_MIDDLE):
    """collinear spheres whose positions and radii grow geometrically: every split peels one sphere off the small end, so the
    reference tree is a chain about n levels deep -- the deepest kind of tree a scene of n primitives can have (the
    wide-tree stack bound of such a tree is ~1.5 x its two-child depth: csrc/rt_api.cpp falls back to the two-child walk when that
    does not fit the LDS of a CU)"""
    sc = SceneDescription(split_type)
    mats = [sc.lambertian(sc.solid((0.8, 0.3, 0.3)), 0.8), sc.lambertian(sc.solid((0.3, 0.8, 0.3)), 0.8), sc.emissive(sc.solid((1.0, 0.9, 0.8)), 3.0)]
    x = 1.0
    for i in range(n):
        sc.sphere((x, 0.0, 0.0), 0.2 * x, mats[2] if i % 9 == 4 else mats[i % 2])
        x *= ratio
    sc.set_sky(sc.lerp((0.5, 0.7, 1.0), (1.0, 1.0, 1.0)), (16, 8))
    return sc


def random_triangle_mesh(n, seed=42, extent=10.0, edge=0.05, emissive_every=1000, split_type=abi.RT_SPLIT_SAH,
                         sampler_res=(100, 100)):
    """The synthetic mesh of BASELINE configs 4/5: centres uniform in [-extent,extent]^3,
    edge vectors uniform in [-edge,edge]^3, flat normals, Lambertian(0.5, 0.5) except every
    `emissive_every`-th triangle = Emit strength 5, Lerp sky as rtweekend1."""
    rng = np.random.default_rng(seed)
    centres = rng.uniform(-extent, extent, (n, 3)).astype(np.float32)
    e1 = rng.uniform(-edge, edge, (n, 3)).astype(np.float32)
    e2 = rng.uniform(-edge, edge, (n, 3)).astype(np.float32)
    v0 = centres
    v1 = centres + e1
    v2 = centres + e2
    vertices = np.stack([v0, v1, v2], axis=1).reshape(-1, 3).astype(np.float32)
    nrm = np.cross(e1.astype(np.float64), e2.astype(np.float64))
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30)
    normals = nrm.astype(np.float32)
    sc = SceneDescription(split_type)
    grey = sc.lambertian(sc.solid((0.5, 0.5, 0.5)), 0.5)
    light = sc.emissive(sc.solid((1.0, 1.0, 1.0)), 5.0)
    m = sc.mesh(vertices, normals)
    pi = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    ni = np.repeat(np.arange(n, dtype=np.uint32)[:, None], 3, axis=1)
    mats = np.full(n, grey, dtype=np.uint32)
    if emissive_every:
        mats[::emissive_every] = light
    sc.mesh_triangles_bulk(m, pi, ni, mats)
    sc.set_sky(sc.lerp((0.5, 0.7, 1.0), (1.0, 1.0, 1.0)), sampler_res)
    return sc


MESH_CAMERA = dict(origin=(0.0, -30.0, 0.0), lookat=(0.0, 0.0, 0.0), vup=(0.0, 0.0, 1.0), fov=60.0,
                   aspect_ratio=float(np.float32(16.0) / np.float32(9.0)), aperture=0.0, focus_dist=10.0)


def all_materials(seed=3, sampler_res=(32, 16)):
    """Every material and texture on the trait surface in one scene."""
    rng = np.random.default_rng(seed)
    sc = SceneDescription()
    ran_vecs, perm = pkg.ssml.perlin_tables(seed)
    img = rng.uniform(0.0, 1.0, (8, 16, 3)).astype(np.float32)
    t_solid = sc.solid((0.7, 0.6, 0.5))
    t_check = sc.checkered((0.9, 0.9, 0.9), (0.1, 0.1, 0.4))
    t_perlin = sc.perlin(ran_vecs, perm)
    t_image = sc.image(img)
    t_sky = sc.lerp((0.5, 0.7, 1.0), (1.0, 1.0, 1.0))
    ground = sc.lambertian(t_check, 0.9)
    sc.sphere((0, -1000, 0), 1000.0, ground)
    sc.sphere((-2.2, 0.5, 0), 0.5, sc.lambertian(t_perlin, 0.8))
    sc.sphere((-1.1, 0.5, 0), 0.5, sc.reflect(t_solid, 0.1))
    sc.sphere((0.0, 0.5, 0), 0.5, sc.refract(sc.solid((1, 1, 1)), 1.5))
    sc.sphere((1.1, 0.5, 0), 0.5, sc.trowbridge_reitz(t_solid, 0.5, (1.5, 1.5, 1.5), 0.0))
    sc.sphere((2.2, 0.5, 0), 0.5, sc.trowbridge_reitz(t_image, 0.3, (1.0, 1.0, 1.0), 1.0))
    sc.sphere((0.0, 2.5, 1.0), 0.4, sc.emissive(sc.solid((1.0, 0.8, 0.6)), 6.0))
    sc.aacuboid((-0.4, 0.0, 1.2), (0.4, 0.5, 1.8), sc.lambertian(t_image, 0.7))
    n = (0.0, 0.0, -1.0)
    sc.triangle([(-3, 0, -2), (3, 0, -2), (0, 3, -2)], [n, n, n], sc.emissive(sc.solid((0.4, 0.6, 1.0)), 2.0))
    sc.set_sky(t_sky, sampler_res)
    return sc


ALL_MATERIALS_CAMERA = dict(origin=(0.0, 1.5, 6.0), lookat=(0.0, 0.6, 0.0), vup=(0.0, 1.0, 0.0), fov=40.0,
                            aspect_ratio=float(np.float32(16.0) / np.float32(9.0)), aperture=0.0, focus_dist=10.0)


def icosphere(subdivisions):
    """unit icosphere: vertices [n,3] f32 (also its smooth normals), triangles [m,3] u32 -- a closed mesh whose
    triangles share every edge and vertex (the watertight / tie-breaking cases random soup never produces)"""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.array(p, dtype=np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(subdivisions):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]
        for (a, b, c) in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.array(v, dtype=np.float32), np.array(f, dtype=np.uint32)


def structured_meshes(subdivisions=4, grid=96, sampler_res=(50, 25)):
    """a smooth-shaded icosphere (emissive) over a rolling heightfield of grid x grid quads (Lambertian) plus a
    diffuse icosphere: closed meshes with shared edges, per-vertex normals, and a mesh light"""
    sc = SceneDescription()
    grey = sc.lambertian(sc.solid((0.6, 0.6, 0.55)), 0.8)
    red = sc.lambertian(sc.solid((0.8, 0.3, 0.25)), 0.9)
    glow = sc.emissive(sc.solid((1.0, 0.9, 0.7)), 4.0)
    # heightfield on the xz plane
    xs = np.linspace(-6, 6, grid + 1, dtype=np.float32)
    X, Z = np.meshgrid(xs, xs, indexing="ij")
    Y = (0.35 * np.sin(1.3 * X) * np.cos(0.9 * Z) - 0.2).astype(np.float32)
    verts = np.stack([X, Y, Z], axis=-1).reshape(-1, 3).astype(np.float32)
    dydx = 0.35 * 1.3 * np.cos(1.3 * X) * np.cos(0.9 * Z)
    dydz = -0.35 * 0.9 * np.sin(1.3 * X) * np.sin(0.9 * Z)
    nrm = np.stack([-dydx, np.ones_like(dydx), -dydz], axis=-1).reshape(-1, 3)
    nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
    idx = np.arange((grid + 1) * (grid + 1), dtype=np.uint32).reshape(grid + 1, grid + 1)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()
    tris = np.concatenate([np.stack([a, b, c], axis=1), np.stack([a, c, d], axis=1)])
    m = sc.mesh(verts, nrm)
    sc.mesh_triangles_bulk(m, tris, tris, np.full(len(tris), grey, np.uint32))
    v, f = icosphere(subdivisions)
    for (centre, radius, mat) in (((0.0, 0.9, 0.0), 0.8, red), ((2.2, 2.4, -1.0), 0.5, glow)):
        mv = (v * np.float32(radius) + np.float32(centre)).astype(np.float32)
        mm = sc.mesh(mv, v)
        sc.mesh_triangles_bulk(mm, f, f, np.full(len(f), mat, np.uint32))
    sc.set_sky(sc.lerp((0.5, 0.7, 1.0), (1.0, 1.0, 1.0)), sampler_res)
    return sc


STRUCTURED_CAMERA = dict(origin=(5.0, 3.0, 6.0), lookat=(0.0, 0.6, 0.0), vup=(0.0, 1.0, 0.0), fov=45.0,
                         aspect_ratio=float(np.float32(16.0) / np.float32(9.0)), aperture=0.0, focus_dist=10.0)


def random_everything(seed, sampler_res=(16, 8)):
    """A random small scene for differential testing: every primitive kind (spheres, free triangles, mesh
    triangles with their own normals, an axis-aligned cuboid) paired at random with every material kind over
    every texture kind, one to three lights (or none), any split type, samplable or unsamplable sky."""
    rng = np.random.default_rng(seed)
    sc = SceneDescription([abi.RT_SPLIT_SAH, abi.RT_SPLIT_MIDDLE, abi.RT_SPLIT_EQUAL_COUNTS][seed % 3])
    ran_vecs, perm = pkg.ssml.perlin_tables(seed)
    textures = [
        sc.solid(rng.uniform(0.1, 0.95, 3)),
        sc.lerp(rng.uniform(0.2, 1.0, 3), rng.uniform(0.0, 0.8, 3)),
        sc.checkered(rng.uniform(0.5, 1.0, 3), rng.uniform(0.0, 0.4, 3)),
        sc.perlin(ran_vecs, perm),
        sc.image(rng.uniform(0.0, 1.0, (int(rng.integers(2, 9)), int(rng.integers(2, 17)), 3)).astype(np.float32)),
    ]

    def tex():
        return textures[int(rng.integers(0, len(textures)))]

    def material():
        k = int(rng.integers(0, 6))
        if k == 0:
            return sc.lambertian(tex(), float(rng.uniform(0.3, 0.95)))
        if k == 1:
            return sc.reflect(tex(), float(rng.uniform(0.0, 0.5)))
        if k == 2:
            return sc.refract(tex(), float(rng.uniform(1.1, 2.0)))
        if k == 3:
            return sc.trowbridge_reitz(tex(), float(rng.uniform(0.1, 0.9)), tuple(rng.uniform(1.0, 2.5, 3)), float(rng.integers(0, 2)))
        if k == 4:
            return sc.lambertian(textures[0], 0.5)
        return sc.trowbridge_reitz(textures[0], 0.4, (1.5, 1.5, 1.5), 0.0)

    sc.sphere((0.0, -1000.0, 0.0), 1000.0, sc.lambertian(tex(), 0.8))
    for _ in range(int(rng.integers(2, 7))):
        sc.sphere(rng.uniform(-3, 3, 3) * (1, 0.3, 1) + (0, 0.7, 0), float(rng.uniform(0.2, 0.8)), material())
    for _ in range(int(rng.integers(0, 5))):
        p = rng.uniform(-3, 3, (3, 3)) * (1, 0.5, 1) + (0, 1.0, 0)
        n = np.cross(p[1] - p[0], p[2] - p[0])
        n = n / max(np.linalg.norm(n), 1e-9)
        sc.triangle([tuple(q) for q in p], [tuple(n)] * 3, material())
    if rng.integers(0, 2):
        lo = rng.uniform(-2.5, 1.5, 3) * (1, 0.0, 1)
        sc.aacuboid(tuple(lo), tuple(lo + rng.uniform(0.3, 1.0, 3)), material())
    if rng.integers(0, 2):
        v, f = icosphere(int(rng.integers(0, 2)))
        centre, radius = rng.uniform(-2, 2, 3) * (1, 0.2, 1) + (0, 1.2, 0), float(rng.uniform(0.3, 0.7))
        m = sc.mesh((v * np.float32(radius) + np.float32(centre)).astype(np.float32), v)
        mat = material()
        sc.mesh_triangles_bulk(m, f, f, np.full(len(f), mat, np.uint32))
    for _ in range(int(rng.integers(0, 4))):
        light = sc.emissive(sc.solid(rng.uniform(0.5, 1.0, 3)), float(rng.uniform(1.0, 8.0)))
        if rng.integers(0, 2):
            sc.sphere(rng.uniform(-3, 3, 3) * (1, 0.3, 1) + (0, 3.0, 0), float(rng.uniform(0.2, 0.6)), light)
        else:
            p = rng.uniform(-2, 2, (3, 3)) + (0, 3.5, 0)
            n = np.cross(p[1] - p[0], p[2] - p[0])
            n = n / max(np.linalg.norm(n), 1e-9)
            sc.triangle([tuple(q) for q in p], [tuple(n)] * 3, light)
    sky = [textures[1], textures[0], sc.solid((0.0, 0.0, 0.0))][int(rng.integers(0, 3))]
    sc.set_sky(sky, sampler_res if rng.integers(0, 4) else (0, 0))
    cam = dict(origin=tuple(rng.uniform(-1, 1, 3) * (3, 0.5, 1) + (0, 1.5, 7.0)), lookat=(0.0, 0.8, 0.0), vup=(0.0, 1.0, 0.0),
               fov=float(rng.uniform(35, 75)), aspect_ratio=float(np.float32(16.0) / np.float32(9.0)), aperture=0.0, focus_dist=10.0)
    return sc, cam


def floor_under(light=None, sky=None, sampler_res=(0, 0), rho=(0.8, 0.6, 0.4), albedo=0.9):
    """A Lambertian floor z = 0 (two big triangles, normal +z) under either one emissive sphere
    `light = (centre_height, radius, colour, strength)` or a Lerp sky `sky = (c1, c2)`: single-bounce scenes
    whose outgoing radiance at the origin has a closed form (tests/test_oracle_render.py)."""
    sc = SceneDescription()
    diffuse = sc.lambertian(sc.solid(rho), albedo)
    a, b = (-5000.0, -5000.0, 0.0), (5000.0, 5000.0, 0.0)
    c, d = (-5000.0, 5000.0, 0.0), (5000.0, -5000.0, 0.0)
    n = (0.0, 0.0, 1.0)
    sc.triangle([a, b, c], [n, n, n], diffuse)
    sc.triangle([a, b, d], [n, n, n], diffuse)
    if light is not None:
        h, r, colour, strength = light
        sc.sphere((0.0, 0.0, h), r, sc.emissive(sc.solid(colour), strength))
    if sky is not None:
        sc.set_sky(sc.lerp(sky[0], sky[1]), sampler_res)
    else:
        sc.set_sky(sc.solid((0.0, 0.0, 0.0)), sampler_res)
    return sc


def small_far_scenes():
    """[(name, scene, point to aim at)]: wide nodes with ABSENT children whose extent is far below 2e-5 of the distance rays
    come from -- where the padded interval test of the wide walk (rt_intersect.h descend4) lets an absent child's inverted
    interval through, so only the node's present mask keeps the walk out of it."""
    out = []
    # three tiny spheres in a clump: one all-leaf root node of extent 4e-5 (a phantom child decodes to node 0, the root itself).
    # (Centroids closer than 100 * f32::EPSILON = 1.2e-5 on the widest axis would make ONE leaf: acceleration/mod.rs:129-134.)
    sc = SceneDescription()
    m = sc.lambertian(sc.solid(0.5), 0.5)
    for k in range(3):
        sc.sphere((2.0e-5 * k, 1.0e-5 * (k % 2), -1.5e-5 * k), 5.0e-6, m)
    sc.set_sky(sc.solid(1.0), (0, 0))
    out.append(("three tiny spheres", sc, np.zeros(3, np.float32)))
    # a floor, a few ordinary spheres and, far from everything, a clump of three small triangles: a 3-child node deep in the tree
    sc = SceneDescription()
    m = sc.lambertian(sc.solid(0.5), 0.5)
    sc.sphere((0.0, 0.0, -1000.0), 999.0, m)
    rng = np.random.default_rng(4)
    for k in range(12):
        sc.sphere(tuple(rng.uniform(-3, 3, 3)), 0.4, m)
    c = np.array([40.0, 35.0, 30.0])
    up = [(0.0, 0.0, 1.0)] * 3
    for k in range(3):
        p = c + np.array([1.0e-4 * k, 0.0, 0.0]) + rng.uniform(-1, 1, 3) * 2.0e-5
        sc.triangle([tuple(p), tuple(p + np.array([3.0e-5, 0, 0])), tuple(p + np.array([0, 3.0e-5, 1.0e-5]))], up, m)
    sc.set_sky(sc.solid(1.0), (0, 0))
    out.append(("a far clump of triangles", sc, c.astype(np.float32)))
    return out
