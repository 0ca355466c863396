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
