"""Bvh::new as the product builds it on the host (csrc/rt_build.cpp, reached without a GPU through
rt_scene_create(..., RT_DEVICE_NONE)) against the oracle's independent restatement (oracle/ora_bvh.c):
nodes byte for byte, primitive order, lights -- acceleration/mod.rs:58-160, split.rs:78-210."""
import numpy as np
import pytest

import scenes

abi = scenes.abi


def _same_tree(hb, O, sc):
    g = hb.HipScene(sc, device=abi.RT_DEVICE_NONE)
    c = O.Scene(sc)
    assert g.counts() == c.counts()
    gn, cn = g.nodes(), c.nodes()
    if gn.tobytes() != cn.tobytes():
        for i in range(len(gn)):
            assert gn[i].tobytes() == cn[i].tobytes(), (i, gn[i], cn[i])
    assert np.array_equal(g.primitive_order(), c.primitive_order())
    assert np.array_equal(g.lights(), c.lights())
    return g


@pytest.mark.parametrize("seed", list(range(120)))
def test_random_scenes(hb, O, seed):
    sc, _ = scenes.random_everything(seed)
    _same_tree(hb, O, sc)


@pytest.mark.parametrize("split", [abi.RT_SPLIT_SAH, abi.RT_SPLIT_MIDDLE, abi.RT_SPLIT_EQUAL_COUNTS])
def test_sizes_and_split_types(hb, O, split):
    for n in (1, 2, 3, 4, 5, 8, 13, 64, 255, 256, 257, 1000, 20000):
        _same_tree(hb, O, scenes.random_spheres(n, seed=n + split, split_type=split, emissive_every=7))
    for n in (1, 2, 5, 12, 300, 50000, 250000):  # >= 32768 primitives: subtrees are built concurrently (rt_build.cpp)
        _same_tree(hb, O, scenes.random_triangle_mesh(n, seed=n + split, extent=4.0, edge=0.4, split_type=split,
                                                      emissive_every=11, sampler_res=(4, 4)))


def test_degenerate_inputs(hb, O):
    """coincident centroids (the 100*EPSILON leaf rule, mod.rs:129-134), identical primitives, zero-size boxes,
    signed zeros in the bounds, more than 255 coincident primitives (a leaf larger than MAX_IN_NODE)"""
    for split in (abi.RT_SPLIT_SAH, abi.RT_SPLIT_MIDDLE, abi.RT_SPLIT_EQUAL_COUNTS):
        sc = scenes.SceneDescription(split)
        m = sc.lambertian(sc.solid(0.5), 0.5)
        for k in range(40):
            sc.sphere((1.0, 2.0, 3.0), 0.5 + 0.01 * (k % 3), m)      # same centre
        for k in range(300):
            sc.sphere((-4.0, 0.0, -0.0), 0.25, m)                       # identical, > MAX_IN_NODE of them
        sc.sphere((0.0, -0.0, 0.0), 0.0, m)                             # a point
        sc.sphere((0.0, -1000.0, 0.0), 1000.0, m)                       # bounds touch +0.0 / -0.0
        n = (0.0, 1.0, 0.0)
        sc.triangle([(-1, -0.0, -1), (1, -0.0, -1), (0, -0.0, 1)], [n, n, n], m)
        sc.triangle([(2, 0, 2), (2, 0, 2), (2, 0, 2)], [n, n, n], m)    # zero area
        sc.set_sky(sc.solid(0.5), (0, 0))
        _same_tree(hb, O, sc)


def test_host_only_scene_cannot_render(hb):
    sc = scenes.random_spheres(3)
    g = hb.HipScene(sc, device=abi.RT_DEVICE_NONE)
    cam = hb.camera_new(origin=(0, 0, 5), lookat=(0, 0, 0), vup=(0, 1, 0), fov=40.0, aspect_ratio=1.0, aperture=0.0, focus_dist=1.0)
    with pytest.raises(hb.RtHipError) as e:
        g.render(cam, abi.default_render_opts(4, 4, 1))
    assert e.value.code == abi.RT_ERR_NO_DEVICE  # no CPU fallback
    with pytest.raises(hb.RtHipError) as e:
        g.check_hit(np.zeros((1, 3), np.float32), np.ones((1, 3), np.float32))
    assert e.value.code == abi.RT_ERR_NO_DEVICE


def test_malformed_descriptors_are_refused_not_crashed(hb):
    """rt_scene_create validates what it is handed (indices into textures / materials / meshes / triangle data,
    primitive and material types, sky) and returns RT_ERR_INVALID_ARGUMENT; NaN or infinite geometry is data,
    not an error (the reference would build a tree from it too).  None of it may crash."""
    import ctypes as C
    rng = np.random.default_rng(99)
    n_refused = n_accepted = 0
    for trial in range(300):
        sc, _ = scenes.random_everything(int(rng.integers(0, 40)))
        desc = sc.desc()
        kind = int(rng.integers(0, 10))
        prims = C.cast(desc.primitives, C.POINTER(C.c_uint32))  # 10 dwords per rt_primitive_desc
        k = int(rng.integers(0, desc.n_primitives))
        if kind == 0:
            prims[10 * k + 0] = int(rng.integers(3, 1000))                  # primitive type
        elif kind == 1:
            prims[10 * k + 1] = desc.n_materials + int(rng.integers(0, 5))  # material index
        elif kind == 2:
            desc.materials[int(rng.integers(0, desc.n_materials))].texture = desc.n_textures + 3
        elif kind == 3:
            desc.materials[int(rng.integers(0, desc.n_materials))].type = int(rng.integers(5, 100))
        elif kind == 4:
            desc.textures[int(rng.integers(0, desc.n_textures))].type = int(rng.integers(5, 100))
        elif kind == 5:
            desc.sky.material = desc.n_materials
        elif kind == 6:
            desc.sky.sampler_res_x, desc.sky.sampler_res_y = 0, 7           # Distribution2D::new would panic
        elif kind == 7:
            desc.split_type = int(rng.integers(3, 50))
        elif kind == 9:  # a mesh / triangle-data index past the end of its array (falls back to a bad type if none exists)
            hit = [i for i in range(desc.n_primitives) if prims[10 * i] in (abi.RT_PRIM_MESH_TRIANGLE, abi.RT_PRIM_TRIANGLE)]
            if hit:
                i = hit[int(rng.integers(0, len(hit)))]
                if prims[10 * i] == abi.RT_PRIM_MESH_TRIANGLE:
                    prims[10 * i + 3 + int(rng.integers(0, 6))] = 0x7FFFFFF0   # point / normal index
                else:
                    prims[10 * i + 2], prims[10 * i + 3] = 0xFFFFFFF0, 0        # triangle-data index (u64)
            else:
                prims[10 * k + 0] = 77
        else:  # NaN / inf in the geometry words of one primitive: accepted, builds, must not hang or crash
            prims[10 * k + 2 + int(rng.integers(0, 4))] = int(rng.choice([0x7FC00000, 0x7F800000, 0xFF800000]))
        h = C.c_void_p()
        rc = hb.lib().rt_scene_create(C.byref(desc), C.c_int(abi.RT_DEVICE_NONE), C.byref(h))
        if rc == 0:
            n_accepted += 1
            hb.lib().rt_scene_destroy(h)
        else:
            assert rc == abi.RT_ERR_INVALID_ARGUMENT, (kind, rc, hb.lib().rt_last_error())
            n_refused += 1
    assert n_refused > 150 and n_accepted > 10


# ---- the wide tree (DevNodeQ4, csrc/rt_build.cpp): an independent check in float64 of what the exactness argument of
# rt_intersect.h ("the wide walk") needs from the host: every reference leaf reachable exactly once, every stored child
# box a SUPERSET of the reference box of everything below it, exact leaf boxes, and a stack bound that covers every path ----
LEAF = 0x80000000
NONE = 0x7FFFFFFE


def _check_wide_tree(g):
    ref_nodes = g.nodes()
    wide, root, depth, leaf_boxes = g.wide_tree()
    leaves = {int(n["primitive_offset"]): n for n in ref_nodes if n["children"][0] < 0}
    # leaves of more than 31 primitives are referenced through a table filled in node order (rt_build.cpp)
    big_leaves = [n for n in ref_nodes if n["children"][0] < 0 and n["number_primitives"] > 31]
    if len(wide) == 0:
        return 0
    seen = []

    def grid_box(node, k):
        lo, hi = np.zeros(3), np.zeros(3)
        for a in range(3):
            step = 2.0 ** (int((node["exps"] >> (8 * a)) & 0xFF) - 127)
            lo[a] = float(node["origin"][a]) + float((int(node["qlo"][a]) >> (8 * k)) & 0xFF) * step
            hi[a] = float(node["origin"][a]) + float((int(node["qhi"][a]) >> (8 * k)) & 0xFF) * step
        return lo, hi

    def below(ref, stack_need):
        """exact bounds (float64) of everything below `ref`; checks containment on the way"""
        if ref & LEAF:
            count, first = (ref >> 26) & 31, ref & 0x03FFFFFF
            if count == 0:
                n = big_leaves[first]
                first = int(n["primitive_offset"])
            else:
                n = leaves[first]
                assert int(n["number_primitives"]) == count
            seen.append(first)
            # the exact leaf box the walk tests is the reference node's box, bit for bit
            assert leaf_boxes[first, 0:3].tobytes() == n["min"].tobytes() and leaf_boxes[first, 4:7].tobytes() == n["max"].tobytes()
            return n["min"].astype(np.float64), n["max"].astype(np.float64)
        node = wide[ref]
        kids = [int(c) for c in node["child"] if int(c) != NONE]
        assert 2 <= len(kids) <= 4 and all(int(c) == NONE for c in node["child"][len(kids):])
        need = stack_need + len(kids) - 1
        assert need + 1 <= depth, (need, depth)
        lo_all, hi_all = np.full(3, np.inf), np.full(3, -np.inf)
        for k, c in enumerate(kids):
            b = below(c, need)
            lo, hi = grid_box(node, k)
            assert (lo <= b[0]).all() and (hi >= b[1]).all(), (ref, k, lo, b[0], hi, b[1])  # conservative: a superset
            # ... and not wildly so: within one grid step of the exact box
            for a in range(3):
                step = 2.0 ** (int((node["exps"] >> (8 * a)) & 0xFF) - 127)
                assert b[0][a] - lo[a] < step + 1e-30 and hi[a] - b[1][a] < step + 1e-30
            lo_all, hi_all = np.minimum(lo_all, b[0]), np.maximum(hi_all, b[1])
        return lo_all, hi_all

    import sys
    sys.setrecursionlimit(10000)
    below(root, 0)
    assert sorted(seen) == sorted(leaves)  # every reference leaf exactly once
    return len(wide)


@pytest.mark.parametrize("seed", list(range(0, 120, 3)))
def test_wide_tree_of_random_scenes(hb, seed):
    sc, _ = scenes.random_everything(seed)
    _check_wide_tree(hb.HipScene(sc, device=abi.RT_DEVICE_NONE))


def test_wide_tree_sizes_and_degenerate_boxes(hb):
    built = 0
    for n in (2, 3, 5, 64, 1000, 20000):
        built += _check_wide_tree(hb.HipScene(scenes.random_spheres(n, seed=n, emissive_every=7), device=abi.RT_DEVICE_NONE))
    for n in (2, 12, 300, 50000):
        built += _check_wide_tree(hb.HipScene(scenes.random_triangle_mesh(n, seed=n, extent=4.0, edge=0.4, sampler_res=(4, 4)),
                                              device=abi.RT_DEVICE_NONE))
    assert built > 1000
    # flat boxes (axis-aligned triangles), huge and tiny coordinates side by side
    sc = scenes.SceneDescription()
    m = sc.lambertian(sc.solid(0.5), 0.5)
    rng = np.random.default_rng(3)
    for i in range(200):
        a = rng.uniform(-1, 1, 3) * 10.0 ** rng.integers(-6, 7)
        sc.aacuboid(tuple(a), tuple(a + rng.uniform(0.0, 1.0, 3) * 10.0 ** rng.integers(-8, 5)), m)
    sc.set_sky(sc.solid(0.5), (0, 0))
    assert _check_wide_tree(hb.HipScene(sc, device=abi.RT_DEVICE_NONE)) > 100
    # bounds beyond 2^60 or non-finite: no wide tree (every ray takes the two-child walk)
    for bad in (3.0e18, float("inf")):
        sc = scenes.SceneDescription()
        m = sc.lambertian(sc.solid(0.5), 0.5)
        for i in range(10):
            sc.sphere((float(i), 0.0, 0.0), 0.3, m)
        sc.sphere((bad, 0.0, 0.0), 1.0, m)
        sc.set_sky(sc.solid(0.5), (0, 0))
        assert len(hb.HipScene(sc, device=abi.RT_DEVICE_NONE).wide_tree()[0]) == 0


# ---- the COMPACT wide node (what the kernels fetch) and its absent children.  rt_intersect.h descend4 accepts a child when its
# padded interval test passes; an absent child's interval is inverted (qlo = 255, qhi = 0 on every axis), which the test rejects
# only while 255 * max|a| > 2 * pad, pad = 1e-5 * (max|b| + 255 * max|a|): a node whose extent is below ~2e-5 of its distance lets
# the phantom through.  Decoded, a phantom child of an all-leaf node is wide node 0 -- the root: the walk would never end.  So the
# node carries a present mask (bits 26-29 of child[1]) and the walk looks slots 2 and 3 up in it.  This walks the compact bytes in
# numpy f32 the way descend4 does: with the mask, every walk ends and reaches no absent child; WITHOUT it the scenes below do
# produce phantoms (the hazard is real, the test would have caught it). ----
def _f32(x):
    return np.float32(x)


def _fma(a, b, c):
    return np.float32(np.float64(a) * np.float64(b) + np.float64(c))


def _descend4_children(node, o, inv, use_present_mask):
    """child slots descend4 would visit (nearest first), emulated in f32; no pruning (limit_valid = false)"""
    exps = int(node["exps"])
    s = [np.uint32(((exps >> (8 * a)) & 0xFF) << 23).view(np.float32) for a in range(3)]
    a_ = [_f32(s[k] * inv[k]) for k in range(3)]
    b_ = [_f32(_f32(node["origin"][k] - o[k]) * inv[k]) for k in range(3)]
    near = [int(node["qhi"][k]) if inv[k] < 0 else int(node["qlo"][k]) for k in range(3)]
    far = [int(node["qlo"][k]) if inv[k] < 0 else int(node["qhi"][k]) for k in range(3)]
    big = _f32(max(abs(b_[0]), abs(b_[1]), abs(b_[2])) + _f32(255.0) * max(abs(a_[0]), abs(a_[1]), abs(a_[2])))
    pad = _f32(_f32(1.0e-5) * big)
    present = int(node["child"][1]) >> 26
    out = []
    for c in range(4):
        tn = max(_fma(_f32((near[k] >> (8 * c)) & 0xFF), a_[k], b_[k]) for k in range(3))
        tf = min(_fma(_f32((far[k] >> (8 * c)) & 0xFF), a_[k], b_[k]) for k in range(3))
        te = max(_f32(tn - pad), _f32(0.0))
        h = (_f32(tf - tn) >= _f32(-2.0) * pad) and (tf >= -pad)
        if use_present_mask and c >= 2:
            h = h and ((present >> c) & 1) != 0
        if h:
            out.append((float(te), c))
    return [c for _, c in sorted(out)]


def _compact_ref(node, c):
    delta = (int(node["exps"]) >> (24 + 2 * c)) & 3
    if (int(node["child"][0]) >> (26 + c)) & 1:
        return LEAF | ((int(node["child"][1]) & 0x03FFFFFF) + delta)
    return (int(node["child"][0]) & 0x03FFFFFF) + delta


def _walk_compact(nodes, root, o, d, use_present_mask, max_steps):
    """(leaf indices reached, absent slots visited, node steps); gives up after max_steps node steps"""
    with np.errstate(all="ignore"):
        d = (d / np.sqrt((d * d).sum(dtype=np.float32))).astype(np.float32)
        inv = (np.float32(1.0) / d).astype(np.float32)
        stack, leaves, phantoms, steps = [root], [], 0, 0
        while stack and steps < max_steps:
            ref = stack.pop()
            if ref & LEAF:
                leaves.append(ref & 0x03FFFFFF)
                continue
            steps += 1
            node = nodes[ref]
            kids = _descend4_children(node, o, inv, use_present_mask)
            present = int(node["child"][1]) >> 26
            phantoms += sum(1 for c in kids if not (present >> c) & 1)
            for c in reversed(kids):
                stack.append(_compact_ref(node, c))
    return leaves, phantoms, steps


def test_compact_wide_walk_never_visits_an_absent_child(hb):
    rng = np.random.default_rng(8)
    hazard_seen = False
    for what, sc, target in scenes.small_far_scenes():
        g = hb.HipScene(sc, device=abi.RT_DEVICE_NONE)
        nodes, leaf_boxes = g.wide_tree_compact()
        explicit, root, depth, _ = g.wide_tree()
        assert len(nodes) == len(explicit) and len(nodes) >= 1, what
        for w in range(len(nodes)):  # the compact node decodes to the explicit one, absent children included
            present = int(nodes[w]["child"][1]) >> 26
            assert present in (0x3, 0x7, 0xF), (what, w, present)
            for c in range(4):
                ref = int(explicit[w]["child"][c])
                assert ((present >> c) & 1) == (ref != NONE), (what, w, c)
                if ref == NONE:
                    continue
                got = _compact_ref(nodes[w], c)
                if got & LEAF:
                    assert int(leaf_boxes[got & 0x03FFFFFF, 3:4].view(np.uint32)[0]) == ref, (what, w, c)
                else:
                    assert got == ref, (what, w, c)
        assert any(int(n["child"][1]) >> 26 != 0xF for n in nodes), what  # the scene does hold nodes with absent children
        for i in range(300):
            o = (target + rng.normal(size=3) * 20.0).astype(np.float32)
            d = (target + rng.normal(size=3).astype(np.float32) * np.float32(2.0e-5) - o).astype(np.float32)
            leaves, phantoms, steps = _walk_compact(nodes, root, o, d, True, 50 * len(nodes) + 50)
            assert phantoms == 0 and steps <= len(nodes), (what, i, phantoms, steps)  # every node at most once: the walk ends
            assert len(leaves) == len(set(leaves)), (what, i)
            _, old_phantoms, _ = _walk_compact(nodes, root, o, d, False, 50 * len(nodes) + 50)
            hazard_seen = hazard_seen or old_phantoms > 0
    assert hazard_seen  # without the mask these very rays do walk into absent children
