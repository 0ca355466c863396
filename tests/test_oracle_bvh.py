"""BVH build + hit queries of the oracle.  The reference has no BVH / intersector tests at all
(SURVEY section 4), so these pin structure (invariants of acceleration/mod.rs:97-160) and the
independence of the closest hit from the split strategy."""
import numpy as np
import pytest

import scenes


def check_tree(nodes, n_prims, prim_boxes_in_slot_order):
    """children partition the parent's primitive range; bounds contain the primitives; preorder ids."""
    seen = np.zeros(n_prims, dtype=int)

    def visit(i):
        nd = nodes[i]
        off, cnt = int(nd["primitive_offset"]), int(nd["number_primitives"])
        lo = prim_boxes_in_slot_order[off:off + cnt, 0].min(axis=0)
        hi = prim_boxes_in_slot_order[off:off + cnt, 1].max(axis=0)
        assert np.array_equal(nd["min"], lo) and np.array_equal(nd["max"], hi)
        c0, c1 = int(nd["children"][0]), int(nd["children"][1])
        if c0 < 0:
            assert c1 < 0
            seen[off:off + cnt] += 1
            return
        assert c0 == i + 1  # preorder numbering: left child is pushed right after its parent
        l, r = nodes[c0], nodes[c1]
        assert int(l["primitive_offset"]) == off
        assert int(r["primitive_offset"]) == off + int(l["number_primitives"])
        assert int(l["number_primitives"]) + int(r["number_primitives"]) == cnt
        visit(c0)
        visit(c1)

    visit(0)
    assert np.all(seen == 1)


@pytest.mark.parametrize("split", [0, 1, 2])
def test_sphere_bvh_invariants(O, split):
    sc = scenes.random_spheres(300, seed=split, split_type=split, emissive_every=50)
    s = O.Scene(sc)
    n_nodes, n_prims, n_lights = s.counts()
    assert n_prims == 300 and n_lights == 6
    order = s.primitive_order()
    assert sorted(order.tolist()) == list(range(300))
    rng = np.random.default_rng(split)  # regenerate the geometry the builder used
    rng_cols = [rng.uniform(0.2, 0.9, 3) for _ in range(4)]
    centres, radii = [], []
    for _ in range(300):
        centres.append(rng.uniform(-10, 10, 3)); radii.append(rng.uniform(0.2, 1.2))
    c = np.array(centres, dtype=np.float32)[order]
    r = np.array(radii, dtype=np.float32)[order][:, None]
    boxes = np.stack([c - r, c + r], axis=1)
    check_tree(s.nodes(), 300, boxes)
    # lights are the slots whose original index is a multiple of 50
    assert sorted(order[s.lights()].tolist()) == [0, 50, 100, 150, 200, 250]


def test_rtweekend1_tree_is_the_hand_derived_one(O):
    s = O.Scene(scenes.load_ssml("rtweekend1").scene)
    nodes = s.nodes()
    assert len(nodes) == 3 and list(s.primitive_order()) == [0, 1] and len(s.lights()) == 0
    assert list(nodes[0]["children"]) == [1, 2]
    assert np.array_equal(nodes[1]["min"], np.float32([-100, -99, -200.5]))
    assert np.array_equal(nodes[2]["max"], np.float32([0.5, 1.5, 0.5]))


def test_overshadowed_has_14_primitives_one_light(O):
    s = O.Scene(scenes.load_ssml("overshadowed").scene)
    n_nodes, n_prims, n_lights = s.counts()
    assert n_prims == 14 and n_lights == 1
    order = s.primitive_order()
    assert order[s.lights()[0]] == 1  # the emissive sphere is the 2nd primitive of the file


@pytest.mark.parametrize("builder", ["spheres", "mesh"])
def test_closest_hit_does_not_depend_on_split_type(O, builder):
    rng = np.random.default_rng(7)
    n = 4000
    org = rng.uniform(-12, 12, (n, 3)).astype(np.float32)
    dirs = rng.normal(size=(n, 3)).astype(np.float32)
    recs = []
    for split in (0, 1, 2):
        sc = scenes.random_spheres(200, seed=5, split_type=split) if builder == "spheres" else \
            scenes.random_triangle_mesh(3000, seed=5, extent=4.0, edge=0.6, split_type=split, sampler_res=(8, 8))
        s = O.Scene(sc)
        h = s.check_hit(org, dirs)
        orig_index = np.where(h["index"] == np.uint64(0xFFFFFFFFFFFFFFFF), -1,
                              s.primitive_order()[np.minimum(h["index"], len(s.primitive_order()) - 1).astype(np.int64)].astype(np.int64))
        recs.append((h, orig_index))
    (h0, i0) = recs[0]
    assert (i0 >= 0).sum() > 200
    for h, i in recs[1:]:
        assert np.array_equal(i, i0)  # same primitive wins (ties are measure-zero for random geometry)
        for f in ("t", "point", "normal", "error", "out", "material"):
            assert np.array_equal(h[f], h0[f])


def test_check_hit_matches_brute_force_single_leaf(O):
    """A BVH walk must return what testing every primitive returns: compare against the same
    scene forced into ONE leaf (all centroids equal on every axis => degenerate split, mod.rs:129-134)."""
    rng = np.random.default_rng(3)
    n = 2000
    org = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    dirs = rng.normal(size=(n, 3)).astype(np.float32)
    sc_tree = scenes.random_spheres(64, seed=9)
    s = O.Scene(sc_tree)
    h = s.check_hit(org, dirs)
    # brute force in numpy/f64 over the original spheres: nearest positive root
    rng2 = np.random.default_rng(9)
    _ = [rng2.uniform(0.2, 0.9, 3) for _ in range(4)]
    C_, R_ = [], []
    for _i in range(64):
        C_.append(rng2.uniform(-10, 10, 3)); R_.append(rng2.uniform(0.2, 1.2))
    C_ = np.array(C_, dtype=np.float32).astype(np.float64); R_ = np.array(R_, dtype=np.float32).astype(np.float64)
    d = dirs.astype(np.float64); d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = org.astype(np.float64)
    oc = C_[None, :, :] - o[:, None, :]
    b = np.einsum("nij,nj->ni", oc, d)
    disc = b * b - (np.einsum("nij,nij->ni", oc, oc) - R_[None, :] ** 2)
    sq = np.sqrt(np.maximum(disc, 0))
    t0, t1 = b - sq, b + sq
    t = np.where(t0 > 1e-6, t0, np.where(t1 > 1e-6, t1, np.inf))
    t = np.where(disc > 0, t, np.inf)
    best = t.min(axis=1)
    hit = np.isfinite(best)
    got_hit = h["index"] != np.uint64(0xFFFFFFFFFFFFFFFF)
    agree = hit == got_hit
    assert agree.mean() > 0.999  # grazing rays may flip in f32
    both = hit & got_hit
    assert np.allclose(h["t"][both], best[both], rtol=2e-4, atol=2e-4)
    orig = s.primitive_order()[h["index"][both].astype(np.int64)]
    assert (orig == t.argmin(axis=1)[both]).mean() > 0.999


def test_check_hit_index_visibility(O):
    s = O.Scene(scenes.load_ssml("overshadowed").scene)
    light = int(s.lights()[0])
    # from above the light, straight down: visible; from inside the ground sphere: blocked
    h = s.check_hit_index([(0, 5, 0), (-3, -1, 0), (3, 5, 0)], [(0, -1, 0), (3, 1.5, 0), (0, -1, 0)], [light, light, light])
    assert h["found"][0] == 1 and abs(h["t"][0] - 4.0) < 1e-5 and h["out"][0] == 1
    assert h["found"][1] == 0  # origin inside the ground sphere: its wall is crossed before the light
    assert h["found"][2] == 0  # misses the light entirely


def test_triangle_hits_match_moeller_trumbore_in_f64(O):
    """The watertight triangle test (primitives/triangle.rs:105-216: shear, edge functions, f64 fallback,
    conservative t bound) against an independent Moeller-Trumbore in numpy float64 over every triangle: same
    closest triangle, same t, hit point on the triangle's plane, barycentric uv as triangle.rs:179 defines it."""
    rng = np.random.default_rng(21)
    n_tri, n_rays = 300, 20000
    sc = scenes.random_triangle_mesh(n_tri, seed=5, extent=3.0, edge=1.5, emissive_every=0, sampler_res=(0, 0))
    s = O.Scene(sc)
    rng_scene = np.random.default_rng(5)  # the generator of scenes.random_triangle_mesh, replayed
    centres = rng_scene.uniform(-3.0, 3.0, (n_tri, 3)).astype(np.float32)
    e1 = rng_scene.uniform(-1.5, 1.5, (n_tri, 3)).astype(np.float32)
    e2 = rng_scene.uniform(-1.5, 1.5, (n_tri, 3)).astype(np.float32)
    v0 = centres.astype(np.float64)
    v1 = (centres + e1).astype(np.float64)
    v2 = (centres + e2).astype(np.float64)
    org = rng.uniform(-6, 6, (n_rays, 3)).astype(np.float32)
    dirs = (rng.uniform(-2, 2, (n_rays, 3)).astype(np.float32) - org)
    h = s.check_hit(org, dirs)
    o = org.astype(np.float64)
    d = dirs.astype(np.float64)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    E1, E2 = v1 - v0, v2 - v0
    best_t = np.full(n_rays, np.inf)
    best_k = np.full(n_rays, -1)
    best_uv = np.zeros((n_rays, 2))
    for k in range(n_tri):
        p = np.cross(d, E2[k])
        det = p @ E1[k]
        ok = np.abs(det) > 1e-14
        inv = np.where(ok, 1.0 / np.where(ok, det, 1.0), 0.0)
        tv = o - v0[k]
        u = np.einsum("ij,ij->i", tv, p) * inv
        q = np.cross(tv, E1[k])
        v = np.einsum("ij,ij->i", d, q) * inv
        t = (q @ E2[k]) * inv
        hit = ok & (u >= 0) & (v >= 0) & (u + v <= 1) & (t > 1e-9) & (t < best_t)
        best_t = np.where(hit, t, best_t)
        best_k = np.where(hit, k, best_k)
        best_uv[hit] = np.stack([u[hit], v[hit]], axis=1)
    got_hit = h["index"] != np.uint64(0xFFFFFFFFFFFFFFFF)
    want_hit = best_k >= 0
    assert want_hit.mean() > 0.2
    # P-hazard 4: Ray::new / Axis::swap_z swap x<->z for Y-dominant rays too, so such rays are sheared
    # along x instead of their dominant axis; when |d.x| is small the conservative `t < delta_t` bound
    # (triangle.rs:160-177) blows up and the reference REJECTS genuine hits.  The oracle reproduces that: every
    # disagreement is a Y-dominant ray with a small x component, and it is always a miss where f64 sees a hit.
    ad = np.abs(d)
    y_dominant = (ad[:, 1] > ad[:, 2]) & ~((ad[:, 0] > ad[:, 1]) & (ad[:, 0] > ad[:, 2]))
    assert (got_hit == want_hit)[~y_dominant].mean() > 0.9995  # rays through an edge may flip between f32 and f64
    lost = want_hit & ~got_hit & y_dominant
    assert not (got_hit & ~want_hit & y_dominant).any()
    assert 0 < lost.sum() < 0.03 * (want_hit & y_dominant).sum() and ad[lost, 0].max() < 0.1
    both = got_hit & want_hit
    slot_to_original = s.primitive_order()
    same = slot_to_original[h["index"][both].astype(np.int64)] == best_k[both]
    assert same[~y_dominant[both]].mean() > 0.9995
    # a Y-dominant ray that lost its nearest triangle reports a farther one, never a nearer one
    assert (h["t"][both][~same] > best_t[both][~same] - 1e-4).all()
    sel = np.nonzero(both)[0][same]
    assert np.allclose(h["t"][sel], best_t[sel], rtol=2e-5, atol=2e-5)
    # uv = b1 (1,0) + b2 (1,1) with b1, b2 the weights of v1, v2 (triangle.rs:179)
    b1, b2 = best_uv[sel, 0], best_uv[sel, 1]
    assert np.allclose(h["uv"][sel, 0], b1 + b2, atol=2e-4) and np.allclose(h["uv"][sel, 1], b2, atol=2e-4)
    point = o[sel] + d[sel] * best_t[sel, None]
    assert np.abs(h["point"][sel] - point).max() < 1e-4
