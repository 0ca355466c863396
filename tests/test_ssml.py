"""The .ssml reader against the grammar and defaults of crates/loader and the
loader's own smoke-test inputs (loader/src/{lib,materials,primitives,textures}.rs tests)."""
import ctypes as C

import numpy as np
import pytest

import scenes

ssml = scenes.pkg.ssml
abi = scenes.abi


def test_rtweekend1_resolved():
    ls = scenes.load_ssml("rtweekend1")
    sc = ls.scene
    assert ls.camera_params["origin"] == (0, 0, 0) and ls.camera_params["lookat"] == (0, 1, 0)
    assert ls.camera_params["vup"] == (0, 0, 1) and ls.camera_params["focus_dist"] == 1.0
    assert ls.camera_params["fov"] == np.float32(121.28449291441745)
    assert sc.n_primitives == 2
    d = sc.desc()
    assert d.n_textures == 3  # sky, grey, __DEFAULT_TEX
    assert d.n_materials == 3  # ground, __DEFAULT_MAT, the sky's Emit(1.0)
    assert list(d.textures[0].colour_one) == pytest.approx([0.5, 0.7, 1.0]) and d.textures[0].type == abi.RT_TEX_LERP
    assert list(d.textures[0].colour_two) == [1.0, 1.0, 1.0]  # a lone float auto-casts to a Vec3
    assert (d.sky.sampler_res_x, d.sky.sampler_res_y) == (100, 100)
    assert d.materials[d.sky.material].type == abi.RT_MAT_EMIT and d.materials[d.sky.material].param == 1.0
    p0 = d.primitives[0]
    assert p0.type == abi.RT_PRIM_SPHERE and list(p0.u.sphere.centre) == [0.0, 1.0, -100.5] and p0.u.sphere.radius == 100.0
    assert d.materials[p0.material].type == abi.RT_MAT_LAMBERTIAN and d.materials[p0.material].param == 1.0


def test_overshadowed_resolved():
    sc = scenes.load_ssml("overshadowed").scene
    d = sc.desc()
    assert d.n_primitives == 14 and d.n_meshes == 1
    assert d.meshes[0].n_vertices == 8 and d.meshes[0].n_normals == 6
    tri = d.primitives[2]
    assert tri.type == abi.RT_PRIM_MESH_TRIANGLE
    assert list(tri.u.mesh_triangle.point_indices) == [0, 1, 2] and list(tri.u.mesh_triangle.normal_indices) == [5, 5, 5]
    last = d.primitives[13]
    assert list(last.u.mesh_triangle.point_indices) == [4, 6, 7] and list(last.u.mesh_triangle.normal_indices) == [4, 4, 4]
    light = d.materials[d.primitives[1].material]
    assert light.type == abi.RT_MAT_EMIT and light.param == 1.5


def test_defaults_and_default_material():
    src = """camera (
)
primitive (
	type sphere
	centre 1 2 3
)
"""
    ls = ssml.load_str(src)
    assert ls.camera_params["origin"] == (3.0, 0.0, 0.0) and ls.camera_params["fov"] == 40.0
    assert ls.camera_params["focus_dist"] == 10.0
    d = ls.scene.desc()
    p = d.primitives[0]
    assert p.u.sphere.radius == 1.0
    m = d.materials[p.material]
    assert m.type == abi.RT_MAT_LAMBERTIAN and m.param == 0.25  # __DEFAULT_MAT
    assert list(d.textures[m.texture].colour_one) == [1.0, 1.0, 1.0]  # __DEFAULT_TEX
    assert (d.sky.sampler_res_x, d.sky.sampler_res_y) == (100, 100)  # no sky object: defaults


def test_loader_smoke_inputs_parse():
    # the inline strings of the reference's loader tests
    for src in ("texture grey (\n\ttype solid\n\tcolour 0.5\n)",
                "texture checkered (\n\ttype checkered\n\tprimary 0.5 0.5 0.0\n\tsecondary 0.0\n)",
                "\ntexture grey (\n\ttype solid\n\tcolour 0.5\n)\nmaterial ground (\n\ttype lambertian\n\ttexture grey\n\talbedo 0.5\n)"):
        objs = ssml.parse(src)
        assert objs and objs[0][0] == "texture"


def test_value_forms():
    objs = ssml.parse("#ver1\nsky (\n\ttexture sky\n\tsampler_res 60 30\n\tfoo 1e-3\n\tbar -1 2.5 .5\n)\n")
    kind, name, v = objs[0]
    assert kind == "sky" and name is None
    assert v["texture"] == "sky" and v["sampler_res"] == (60.0, 30.0) and v["foo"] == (np.float32(1e-3),)
    assert v["bar"] == (-1.0, 2.5, 0.5)


def test_materials_and_trowbridge_alpha_is_squared():
    src = """camera (
)
texture t (
	type solid
	colour 0.2 0.4 0.6
)
material a (
	type trowbridge_reitz
	texture t
	alpha 0.5
	ior 1.5
	metallic 1.0
)
material b (
	type refract
)
material c (
	type reflect
	fuzz 0.3
)
material d (
	type emissive
)
primitive (
	type sphere
	material a
	centre 0 0 0
)
"""
    d = ssml.load_str(src).scene.desc()
    a = d.materials[0]
    assert a.type == abi.RT_MAT_TROWBRIDGE_REITZ and a.param == 0.25 and list(a.ior) == [1.5, 1.5, 1.5] and a.metallic == 1.0
    assert d.materials[1].type == abi.RT_MAT_REFRACT and d.materials[1].param == 1.5
    assert d.materials[2].type == abi.RT_MAT_REFLECT and d.materials[2].param == np.float32(0.3)
    assert d.materials[3].type == abi.RT_MAT_EMIT and d.materials[3].param == 1.5


@pytest.mark.parametrize("src,msg", [
    ("primitive (\n\ttype sphere\n\tcentre 0 0 0\n)\n", "camera"),
    ("camera (\n)\nprimitive (\n\ttype sphere\n)\n", "centre"),
    ("camera (\n)\nprimitive (\n\ttype cube\n\tcentre 0 0 0\n)\n", "primitive type"),
    ("camera (\n)\ntexture x (\n\tcolour 1\n)\n", "type"),
    ("camera (\n", "')'"),
    ("banana (\n)\n", "kind"),
])
def test_errors(src, msg):
    with pytest.raises(ssml.SsmlError) as e:
        ssml.load_str(src)
    assert msg in str(e.value)


def test_primitive_record_layout_of_bulk_builder():
    sc = scenes.random_triangle_mesh(10, seed=1, sampler_res=(4, 4))
    sc.sphere((1, 2, 3), 4.0, 0)
    d = sc.desc()
    assert d.n_primitives == 11
    assert d.primitives[3].type == abi.RT_PRIM_MESH_TRIANGLE and list(d.primitives[3].u.mesh_triangle.point_indices) == [9, 10, 11]
    assert list(d.primitives[3].u.mesh_triangle.normal_indices) == [3, 3, 3]
    assert d.primitives[10].type == abi.RT_PRIM_SPHERE and d.primitives[10].u.sphere.radius == 4.0
    assert C.sizeof(abi.PrimitiveDesc) == 40


def test_obj_mesh_loader():
    """`mesh ( type mesh, obj <file> )` -> loader/src/obj.rs:11-61 semantics"""
    ls = scenes.load_ssml("pyramid")
    d = ls.scene.desc()
    assert d.n_meshes == 2  # one MeshData per `o` object
    assert d.meshes[0].n_vertices == 5 and d.meshes[0].n_normals == 5
    assert d.meshes[1].n_vertices == 4 and d.meshes[1].n_normals == 1  # indices become object-relative
    assert d.n_primitives == 2 + 2 + 2 + 2  # quad fan (2) + 2 glow + 2 stone + slab (2)
    stone = [i for i in range(d.n_materials) if d.materials[i].type == abi.RT_MAT_LAMBERTIAN and d.materials[i].param == np.float32(0.8)][0]
    glow = [i for i in range(d.n_materials) if d.materials[i].type == abi.RT_MAT_EMIT and d.materials[i].param == 3.0][0]
    mats = [d.primitives[i].material for i in range(d.n_primitives)]
    assert mats[:6] == [stone, stone, glow, glow, stone, stone]
    default_mat = [i for i in range(d.n_materials) if d.materials[i].type == abi.RT_MAT_LAMBERTIAN and d.materials[i].param == 0.25][0]
    assert mats[6:] == [default_mat, default_mat]  # no usemtl -> "default" -> not found -> __DEFAULT_MAT
    q0, q1 = d.primitives[0], d.primitives[1]
    assert list(q0.u.mesh_triangle.point_indices) == [0, 1, 2] and list(q1.u.mesh_triangle.point_indices) == [0, 2, 3]
    s0, s1 = d.primitives[6], d.primitives[7]
    assert s0.u.mesh_triangle.mesh == 1 and list(s0.u.mesh_triangle.point_indices) == [0, 2, 1]
    assert list(s1.u.mesh_triangle.point_indices) == [0, 3, 2] and list(s1.u.mesh_triangle.normal_indices) == [0, 0, 0]  # negative indices


def test_obj_without_normals_is_rejected(tmp_path):
    (tmp_path / "a.obj").write_text("o a\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    (tmp_path / "a.ssml").write_text("camera (\n)\nmesh (\n\ttype mesh\n\tobj a.obj\n)\n")
    with pytest.raises(ssml.SsmlError) as e:
        ssml.load_file(str(tmp_path / "a.ssml"))
    assert "vertex normals" in str(e.value)


def test_image_texture_is_decoded_like_to_rgb32f(pkg, hb, tmp_path):
    """`texture … ( type image filename … )` (loader/src/textures.rs:51-60): 8-bit pixels become value / 255 in f32,
    RGB order, no gamma -- what image::open(..).to_rgb32f() yields (textures/mod.rs:236-246)"""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(4)
    px = rng.integers(0, 256, (5, 9, 3), dtype=np.uint8)
    Image.fromarray(px, "RGB").save(tmp_path / "tex.png")
    src = """
camera (
    origin 0 0 3
    lookat 0 0 0
    vup 0 1 0
    fov 40
)
texture pic (
    type image
    filename tex.png
)
material m (
    type lambertian
    texture pic
    albedo 0.8
)
primitive (
    type sphere
    material m
    centre 0 0 0
    radius 1
)
sky (
    texture pic
)
"""
    ls = pkg.ssml.load_str(src, base_dir=str(tmp_path))
    t = [t for t in ls.scene.textures if t.type == abi.RT_TEX_IMAGE][0]
    assert (t.image_width, t.image_height) == (9, 5)
    got = np.ctypeslib.as_array(t.image_rgb, shape=(5, 9, 3))
    assert np.array_equal(got, px.astype(np.float32) / np.float32(255.0))
    with pytest.raises(pkg.ssml.SsmlError):
        pkg.ssml.load_str("texture t (\n    type image\n)\nsky (\n    texture t\n)\n")
