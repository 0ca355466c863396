"""Minimal `.ssml` scene reader: feeds the reference's own scene files to the back end.

Grammar restated from crates/loader/src/parser.rs:82-197 and the per-type defaults from
crates/loader/src/{lib,misc,textures,materials,primitives,meshes}.rs:

    file    = ["#ver1"] object*
    object  = kind [name] "(" (key value NEWLINE)* ")"
    kind    = camera | material | primitive | sky | texture | mesh
    value   = 3 floats | 2 floats | 1 float | free text to end of line

Load order (loader/src/lib.rs:218-240): textures (+ `__DEFAULT_TEX`), materials
(+ `__DEFAULT_MAT`), camera, sky, primitives, meshes.
"""
import re

import numpy as np

from . import abi
from .scene import SceneDescription

_KINDS = ("camera", "material", "primitive", "sky", "texture", "mesh")
_IDENT = r"[A-Za-z_][A-Za-z0-9_]*"
_FLOAT = r"[+-]?(?:(?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?|inf(?:inity)?|nan)"


class SsmlError(ValueError):
    pass


def _parse_value(text):
    """parser.rs:119-131: try 3 floats, then 2, then 1 (each `preceded(space0, double)`), else text."""
    one = re.compile(r"[ \t]*(" + _FLOAT + r")", re.IGNORECASE)
    for n in (3, 2, 1):
        pos, got = 0, []
        for _ in range(n):  # nom's `double` is greedy and never backtracks: take floats one at a time
            m = one.match(text, pos)
            if not m:
                break
            got.append(np.float32(float(m.group(1))))
            pos = m.end()
        if len(got) == n and text[pos:] == "":
            return tuple(got)
    return text.lstrip(" \t")


def parse(src):
    """-> list of (kind, name|None, {key: value}) in file order."""
    pos = 0
    n = len(src)

    def skip_ws():
        nonlocal pos
        while pos < n and src[pos] in " \t\r\n":
            pos += 1

    skip_ws()
    if src.startswith("#ver1", pos):
        pos += 5
    objects = []
    while True:
        skip_ws()
        if pos >= n:
            break
        m = re.compile(r"(" + "|".join(_KINDS) + r")").match(src, pos)
        if not m:
            raise SsmlError(f"expected object kind at offset {pos}")
        kind = m.group(1)
        pos = m.end()
        name = None
        m = re.compile(r"[ \t]+(" + _IDENT + r")").match(src, pos)
        if m:
            name = m.group(1)
            pos = m.end()
        skip_ws()
        if pos >= n or src[pos] != "(":
            raise SsmlError(f"expected '(' after {kind}")
        pos += 1
        skip_ws()
        values = {}
        while True:
            m = re.compile(r"[ \t]*(" + _IDENT + r")[ \t]+([^\r\n]*)(\r?\n)").match(src, pos)
            if not m:
                break
            values[m.group(1)] = _parse_value(m.group(2))  # HashMap: a repeated key keeps the last value
            pos = m.end()
        skip_ws()
        if pos >= n or src[pos] != ")":
            raise SsmlError(f"expected ')' closing {kind} at offset {pos}")
        pos += 1
        objects.append((kind, name, values))
    return objects


class _Props:
    """Properties accessors with the loader's auto-cast (loader/src/lib.rs:103-178)."""

    def __init__(self, values):
        self.v = values

    def vec3(self, key):
        x = self.v.get(key)
        if isinstance(x, tuple) and len(x) == 3:
            return x
        if isinstance(x, tuple) and len(x) == 1:
            return (x[0], x[0], x[0])
        return None

    def vec2(self, key):
        x = self.v.get(key)
        if isinstance(x, tuple) and len(x) == 2:
            return x
        if isinstance(x, tuple) and len(x) == 1:
            return (x[0], x[0])
        return None

    def float(self, key):
        x = self.v.get(key)
        if isinstance(x, tuple) and len(x) == 1:
            return x[0]
        return None

    def text(self, key):
        x = self.v.get(key)
        return x if isinstance(x, str) else None


def _or(value, default):
    return default if value is None else value


def _f32_as_usize(x):
    x = float(x)
    if not x > 0:
        return 0
    return int(x)


class LoadedScene:
    def __init__(self, scene, camera_params):
        self.scene = scene  # SceneDescription
        self.camera_params = camera_params  # dict for rt_camera_new


def resolve_path(path, base_dir):
    """The reference opens the path as given (relative to the process cwd, loader/src/obj.rs:12); as a
    convenience a relative path that does not exist there is also tried next to the scene file."""
    import os
    if os.path.isabs(path) or os.path.exists(path) or base_dir is None:
        return path
    return os.path.join(base_dir, path)


def decode_image_rgb32f(path):
    """What `image::open(path).to_rgb32f()` yields for 8-bit images (textures/mod.rs:212-246): RGB, each channel
    `value as f32 / 255.0`, no gamma.  Decoding is host-side I/O outside the hot path; it uses Pillow when it is
    installed and refuses other bit depths rather than guess their scaling."""
    try:
        from PIL import Image
    except ImportError as e:  # pragma: no cover
        raise SsmlError("image textures need Pillow to decode; pass decoded pixels through SceneDescription.image") from e
    import numpy as np
    with Image.open(path) as im:
        if im.mode not in ("RGB", "RGBA", "L", "LA", "P", "1"):
            raise SsmlError(f"image texture {path}: unsupported pixel format {im.mode} (8-bit images only)")
        rgb = np.asarray(im.convert("RGB"), dtype=np.uint8)
    if rgb.shape[0] == 0 or rgb.shape[1] == 0:
        raise SsmlError(f"image texture {path} is empty")
    return rgb.astype(np.float32) / np.float32(255.0)


def parse_obj(text):
    """Wavefront OBJ as the `wavefront_obj` 10.0 crate presents it to loader/src/obj.rs:11-61: a list of
    objects, each with its OWN vertex / normal arrays (indices made object-relative) and its faces grouped
    by `usemtl`; polygons are fan-triangulated, points and lines are ignored by the loader.
    -> [ {"name", "vertices" [n,3] f32, "normals" [m,3] f32, "geometry": [(material_name|None, [(v,vn|None)x3, ...])]} ]"""
    objects = []
    cur = None
    v_base = n_base = t_base = 0
    v_total = n_total = t_total = 0

    def new_object(name):
        nonlocal cur, v_base, n_base, t_base
        cur = {"name": name, "vertices": [], "normals": [], "geometry": []}
        objects.append(cur)
        v_base, n_base, t_base = v_total, n_total, t_total

    def cur_geometry(material):
        if not cur["geometry"] or cur["geometry"][-1][0] != material:
            cur["geometry"].append((material, []))
        return cur["geometry"][-1][1]

    material = None
    for raw in text.splitlines():
        line = raw.split("#", 1)[0].strip()
        if not line:
            continue
        tok = line.split()
        key, args = tok[0], tok[1:]
        if key == "o":
            new_object(args[0] if args else "")
            material = None
        elif key == "v":
            if cur is None:
                new_object("")
            cur["vertices"].append([np.float32(float(a)) for a in args[:3]])
            v_total += 1
        elif key == "vn":
            if cur is None:
                new_object("")
            cur["normals"].append([np.float32(float(a)) for a in args[:3]])
            n_total += 1
        elif key == "vt":
            t_total += 1
        elif key == "usemtl":
            material = args[0] if args else None
        elif key == "f":
            if cur is None:
                new_object("")
            corners = []
            for a in args:
                parts = a.split("/")
                vi = int(parts[0])
                vi = vi - 1 - v_base if vi > 0 else len(cur["vertices"]) + vi
                ni = None
                if len(parts) >= 3 and parts[2] != "":
                    ni = int(parts[2])
                    ni = ni - 1 - n_base if ni > 0 else len(cur["normals"]) + ni
                corners.append((vi, ni))
            shapes = cur_geometry(material)
            for k in range(1, len(corners) - 1):  # fan triangulation
                shapes.append((corners[0], corners[k], corners[k + 1]))
    for o in objects:
        o["vertices"] = np.asarray(o["vertices"], dtype=np.float32).reshape(-1, 3)
        o["normals"] = np.asarray(o["normals"], dtype=np.float32).reshape(-1, 3)
    return objects


def add_obj(sc, path, lookup_material):
    """load_obj (loader/src/obj.rs:11-61): one MeshData per object, one MeshTriangle per triangle, material
    by `usemtl` name looked up among the SCENE's materials, name "default" when the group has none,
    falling back to the default material; triangles without vertex normals are an error."""
    with open(path, "r") as f:
        objects = parse_obj(f.read())
    for o in objects:
        mesh = sc.mesh(o["vertices"], o["normals"] if len(o["normals"]) else np.zeros((1, 3), np.float32))
        pi, ni, mats = [], [], []
        for material_name, shapes in o["geometry"]:
            mat = lookup_material(material_name if material_name is not None else "default")
            for tri in shapes:
                if any(c[1] is None for c in tri):
                    raise SsmlError("Please export obj file with vertex normals!")
                for c in tri:
                    if not (0 <= c[0] < len(o["vertices"])) or not (0 <= c[1] < len(o["normals"])):
                        raise SsmlError(f"obj index out of range in object '{o['name']}'")
                pi.append([c[0] for c in tri])
                ni.append([c[1] for c in tri])
                mats.append(mat)
        if pi:
            sc.mesh_triangles_bulk(mesh, np.asarray(pi, np.uint32), np.asarray(ni, np.uint32), np.asarray(mats, np.uint32))


def load_str(src, split_type=abi.RT_SPLIT_SAH, perlin_seed=0, base_dir=None):
    objects = parse(src)
    sc = SceneDescription(split_type)
    tex_by_name, mat_by_name = {}, {}

    # ---- textures (loader/src/textures.rs) ----
    for kind, name, values in objects:
        if kind != "texture":
            continue
        p = _Props(values)
        ttype = p.text("type")
        if ttype is None:
            raise SsmlError("missing required type for texture")
        if ttype == "solid":
            idx = sc.solid(_or(p.vec3("colour"), (0.5, 0.5, 0.5)))
        elif ttype == "lerp":
            idx = sc.lerp(_or(p.vec3("primary"), (1, 1, 1)), _or(p.vec3("secondary"), (0, 0, 0)))
        elif ttype == "checkered":
            idx = sc.checkered(_or(p.vec3("primary"), (1, 1, 1)), _or(p.vec3("secondary"), (0, 0, 0)))
        elif ttype == "perlin":
            ran_vecs, perm = perlin_tables(perlin_seed)
            idx = sc.perlin(ran_vecs, perm)
        elif ttype == "image":  # ImageTexture::load + ::new  loader/src/textures.rs:51-60, textures/mod.rs:208-246
            filename = p.text("filename")
            if filename is None:
                raise SsmlError("missing required filename")
            idx = sc.image(decode_image_rgb32f(resolve_path(filename, base_dir)))
        else:
            raise SsmlError(f"required a known value for texture type, found '{ttype}'")
        if name is not None:
            tex_by_name[name] = idx
    tex_by_name["__DEFAULT_TEX"] = sc.solid((1, 1, 1))  # loader/src/lib.rs:354-368

    def tex_of(p):
        t = p.text("texture")
        return tex_by_name.get(t, tex_by_name["__DEFAULT_TEX"]) if t is not None else tex_by_name["__DEFAULT_TEX"]

    # ---- materials (loader/src/materials.rs) ----
    def load_material(values):
        p = _Props(values)
        mtype = p.text("type")
        if mtype is None:
            raise SsmlError("missing required type for material")
        if mtype == "lambertian":
            return sc.lambertian(tex_of(p), _or(p.float("albedo"), 0.5))
        if mtype == "emissive":
            return sc.emissive(tex_of(p), _or(p.float("strength"), 1.5))
        if mtype == "reflect":
            return sc.reflect(tex_of(p), _or(p.float("fuzz"), 0.1))
        if mtype == "refract":
            return sc.refract(tex_of(p), _or(p.float("eta"), 1.5))
        if mtype == "trowbridge_reitz":
            return sc.trowbridge_reitz(tex_of(p), _or(p.float("alpha"), 0.5), _or(p.vec3("ior"), (1, 1, 1)),
                                       _or(p.float("metallic"), 0.0))
        raise SsmlError(f"required a known value for material type, found '{mtype}'")

    for kind, name, values in objects:
        if kind == "material":
            idx = load_material(values)
            if name is not None:
                mat_by_name[name] = idx
    mat_by_name["__DEFAULT_MAT"] = load_material(  # loader/src/lib.rs:382-397
        {"type": "lambertian", "texture": "__DEFAULT_TEX", "albedo": (np.float32(0.25),)})

    def mat_of(p):
        m = p.text("material")
        return mat_by_name.get(m, mat_by_name["__DEFAULT_MAT"]) if m is not None else mat_by_name["__DEFAULT_MAT"]

    # ---- camera (loader/src/misc.rs:6-18) ----
    cams = [v for k, _, v in objects if k == "camera"]
    if not cams:
        raise SsmlError("missing required camera object")
    p = _Props(cams[0])
    camera_params = dict(
        origin=_or(p.vec3("origin"), (3.0, 0.0, 0.0)),
        lookat=_or(p.vec3("lookat"), (0.0, 0.0, 0.0)),
        vup=_or(p.vec3("vup"), (0.0, 1.0, 0.0)),
        fov=_or(p.float("fov"), 40.0),
        aspect_ratio=float(np.float32(16.0) / np.float32(9.0)),
        aperture=_or(p.float("aperture"), 0.0),
        focus_dist=_or(p.float("focus_dis"), 10.0),
    )

    # ---- sky (loader/src/misc.rs:20-38) ----
    skies = [v for k, _, v in objects if k == "sky"]
    p = _Props(skies[0] if skies else {})
    res = _or(p.vec2("sampler_res"), (100.0, 100.0))
    sc.set_sky(tex_of(p), (_f32_as_usize(res[0]), _f32_as_usize(res[1])))

    # ---- primitives (loader/src/primitives.rs) ----
    for kind, _, values in objects:
        if kind != "primitive":
            continue
        p = _Props(values)
        ptype = p.text("type")
        if ptype is None:
            raise SsmlError("missing required type for primitive")
        if ptype != "sphere":
            raise SsmlError(f"required a known value for primitive type, found '{ptype}'")
        centre = p.vec3("centre")
        if centre is None:
            raise SsmlError("expected centre on sphere, found nothing")
        sc.sphere(centre, _or(p.float("radius"), 1.0), mat_of(p))

    # ---- meshes (loader/src/meshes.rs) ----
    for kind, _, values in objects:
        if kind != "mesh":
            continue
        p = _Props(values)
        mtype = p.text("type")
        if mtype is None:
            raise SsmlError("missing required type for mesh")
        if mtype == "aacuboid":
            p1, p2 = p.vec3("point_one"), p.vec3("point_two")
            if p1 is None or p2 is None:
                raise SsmlError("expected point_one and point_two on aacuboid")
            sc.aacuboid(p1, p2, mat_of(p))
        elif mtype == "mesh":
            path = p.text("obj")
            if path is None:
                raise SsmlError("expected obj on mesh, found nothing")
            add_obj(sc, resolve_path(path, base_dir), lambda name: mat_by_name.get(name, mat_by_name["__DEFAULT_MAT"]))
        else:
            raise SsmlError(f"required a known value for mesh type, found '{mtype}'")

    return LoadedScene(sc, camera_params)


def load_file(path, **kw):
    import os
    with open(path, "r") as f:
        return load_str(f.read(), base_dir=os.path.dirname(os.path.abspath(path)), **kw)


def perlin_tables(seed=0):
    """Perlin::new draws its tables from thread_rng (textures/mod.rs:91-152); here they come from
    a seeded numpy generator so a scene file maps to one fixed texture."""
    rng = np.random.default_rng(seed)
    r = rng.uniform(-1.0, 1.0, size=256).astype(np.float32)
    ran_vecs = np.repeat(r[:, None], 3, axis=1)  # gen_range(-1.0..1.0) * Vec3::one()
    perm = np.stack([rng.permutation(256) for _ in range(3)]).astype(np.uint32)
    return ran_vecs, perm
