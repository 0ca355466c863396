"""Host-side scene description: the POD the reference keeps in its Region arena
(crates/region) flattened into the arrays rt_scene_create() consumes.

The builder mirrors the constructors of the reference types it stands for
(SolidColour::new, Lambertian::new, Sphere::new, MeshTriangle::new, Sky::new ...), with the
loader's defaults where the reference has them (crates/loader/src/*.rs).
"""
import ctypes as C

import numpy as np

from . import abi


def _v3(v):
    if np.isscalar(v):
        v = (v, v, v)  # Properties::vec3 auto-cast: Float -> x * Vec3::one() (loader/src/lib.rs:138-144)
    a = np.asarray(v, dtype=np.float32).reshape(3)
    return (C.c_float * 3)(float(a[0]), float(a[1]), float(a[2]))


class SceneDescription:
    """Accumulates textures / materials / meshes / primitives / sky; `.desc()` yields rt_scene_desc."""

    def __init__(self, split_type=abi.RT_SPLIT_SAH):
        self.textures = []
        self.materials = []
        self.meshes = []  # (vertices float32 [n,3], normals float32 [m,3])
        self.prim_records = []  # small scenes: python list of PrimitiveDesc
        self.prim_array = None  # large scenes: a ready ctypes array
        self.triangles = []
        self.sky = None
        self.split_type = split_type
        self._keep = []

    # ---- textures (textures/mod.rs) ----
    def _tex(self, type_, c1=(0, 0, 0), c2=(0, 0, 0)):
        t = abi.TextureDesc()
        t.type = type_
        t.colour_one = _v3(c1)
        t.colour_two = _v3(c2)
        self.textures.append(t)
        return len(self.textures) - 1

    def solid(self, colour=0.5):  # loader default colour 0.5 (loader/src/textures.rs:78-84)
        return self._tex(abi.RT_TEX_SOLID, colour)

    def lerp(self, primary=1.0, secondary=0.0):  # loader/src/textures.rs:69-76
        return self._tex(abi.RT_TEX_LERP, primary, secondary)

    def checkered(self, primary=1.0, secondary=0.0):  # loader/src/textures.rs:42-49
        return self._tex(abi.RT_TEX_CHECKERED, primary, secondary)

    def image(self, rgb):
        """rgb: float32 [H, W, 3] as `to_rgb32f` yields (textures/mod.rs:236-246)."""
        rgb = np.ascontiguousarray(rgb, dtype=np.float32)
        h, w, _ = rgb.shape
        i = self._tex(abi.RT_TEX_IMAGE)
        self._keep.append(rgb)
        self.textures[i].image_rgb = rgb.ctypes.data_as(C.POINTER(C.c_float))
        self.textures[i].image_width = w
        self.textures[i].image_height = h
        return i

    def perlin(self, ran_vecs, perm):
        """ran_vecs float32 [256,3]; perm uint32 [3,256] = perm_x, perm_y, perm_z (textures/mod.rs:76-81)."""
        ran_vecs = np.ascontiguousarray(ran_vecs, dtype=np.float32).reshape(256, 3)
        perm = np.ascontiguousarray(perm, dtype=np.uint32).reshape(3, 256)
        i = self._tex(abi.RT_TEX_PERLIN)
        self._keep += [ran_vecs, perm]
        self.textures[i].perlin_ran_vecs = ran_vecs.ctypes.data_as(C.POINTER(C.c_float))
        self.textures[i].perlin_perm = perm.ctypes.data_as(C.POINTER(C.c_uint32))
        return i

    # ---- materials (materials/*.rs, defaults loader/src/materials.rs:43-111) ----
    def _mat(self, type_, texture, param, ior=(1, 1, 1), metallic=0.0):
        m = abi.MaterialDesc()
        m.type = type_
        m.texture = texture
        m.param = float(np.float32(param))
        m.ior = _v3(ior)
        m.metallic = float(np.float32(metallic))
        self.materials.append(m)
        return len(self.materials) - 1

    def emissive(self, texture, strength=1.5):
        return self._mat(abi.RT_MAT_EMIT, texture, strength)

    def lambertian(self, texture, albedo=0.5):
        return self._mat(abi.RT_MAT_LAMBERTIAN, texture, albedo)

    def reflect(self, texture, fuzz=0.1):
        return self._mat(abi.RT_MAT_REFLECT, texture, fuzz)

    def refract(self, texture, eta=1.5):
        return self._mat(abi.RT_MAT_REFRACT, texture, eta)

    def trowbridge_reitz(self, texture, roughness=0.5, ior=(1, 1, 1), metallic=0.0):
        # TrowbridgeReitz::new stores alpha = roughness^2 (materials/trowbridge_reitz.rs:17-24);
        # the loader passes the .ssml `alpha` key as that roughness (P-hazard 11)
        r = np.float32(roughness)
        return self._mat(abi.RT_MAT_TROWBRIDGE_REITZ, texture, r * r, ior, metallic)

    # ---- primitives ----
    def sphere(self, centre, radius, material):  # Sphere::new (primitives/sphere.rs:19-27)
        p = abi.PrimitiveDesc()
        p.type = abi.RT_PRIM_SPHERE
        p.material = material
        p.u.sphere.centre = _v3(centre)
        p.u.sphere.radius = float(np.float32(radius))
        self.prim_records.append(p)
        return len(self.prim_records) - 1

    def triangle(self, points, normals, material):  # Triangle::new (primitives/triangle.rs:21-28)
        t = abi.TriangleData()
        pts = np.asarray(points, dtype=np.float32).reshape(9)
        nrm = np.asarray(normals, dtype=np.float32).reshape(9)
        t.points = (C.c_float * 9)(*[float(x) for x in pts])
        t.normals = (C.c_float * 9)(*[float(x) for x in nrm])
        self.triangles.append(t)
        p = abi.PrimitiveDesc()
        p.type = abi.RT_PRIM_TRIANGLE
        p.material = material
        p.u.triangle.data = len(self.triangles) - 1
        self.prim_records.append(p)
        return len(self.prim_records) - 1

    def mesh(self, vertices, normals):  # MeshData::new (primitives/triangle.rs:58-67)
        v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
        n = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        self.meshes.append((v, n))
        return len(self.meshes) - 1

    def mesh_triangle(self, mesh, point_indices, normal_indices, material):  # MeshTriangle::new (:38-56)
        p = abi.PrimitiveDesc()
        p.type = abi.RT_PRIM_MESH_TRIANGLE
        p.material = material
        p.u.mesh_triangle.mesh = mesh
        p.u.mesh_triangle.point_indices = (C.c_uint32 * 3)(*[int(i) for i in point_indices])
        p.u.mesh_triangle.normal_indices = (C.c_uint32 * 3)(*[int(i) for i in normal_indices])
        self.prim_records.append(p)
        return len(self.prim_records) - 1

    def mesh_triangles_bulk(self, mesh, point_indices, normal_indices, materials):
        """Append many MeshTriangles at once (numpy [n,3] index arrays, [n] material ids)."""
        pi = np.ascontiguousarray(point_indices, dtype=np.uint32).reshape(-1, 3)
        ni = np.ascontiguousarray(normal_indices, dtype=np.uint32).reshape(-1, 3)
        mats = np.ascontiguousarray(materials, dtype=np.uint32).reshape(-1)
        n = pi.shape[0]
        assert C.sizeof(abi.PrimitiveDesc) == 40
        rec = np.zeros((n, 10), dtype=np.uint32)
        rec[:, 0] = abi.RT_PRIM_MESH_TRIANGLE
        rec[:, 1] = mats
        rec[:, 2] = mesh
        rec[:, 3:6] = pi
        rec[:, 6:9] = ni
        if self.prim_records:
            head = np.frombuffer(bytes(b"".join(bytes(p) for p in self.prim_records)), dtype=np.uint32).reshape(-1, 10)
            rec = np.concatenate([head, rec], axis=0)
            self.prim_records = []
        if self.prim_array is not None:
            rec = np.concatenate([self.prim_array, rec], axis=0)
        self.prim_array = np.ascontiguousarray(rec)

    def aacuboid(self, point_one, point_two, material):
        """`mesh ( type aacuboid ... )`: 8 vertices, 6 axis normals, 12 MeshTriangles
        (crates/loader/src/meshes.rs:26-103)."""
        p1 = np.asarray(point_one, dtype=np.float32)
        p2 = np.asarray(point_two, dtype=np.float32)
        mn, mx = np.minimum(p1, p2), np.maximum(p1, p2)
        points = [
            mn, (mx[0], mn[1], mn[2]), (mx[0], mx[1], mn[2]), (mn[0], mx[1], mn[2]),
            (mn[0], mn[1], mx[2]), (mx[0], mn[1], mx[2]), mx, (mn[0], mx[1], mx[2]),
        ]
        normals = [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]
        m = self.mesh(np.array(points, dtype=np.float32), np.array(normals, dtype=np.float32))
        tris = [
            ([0, 1, 2], 5), ([0, 2, 3], 5), ([0, 1, 5], 3), ([0, 5, 4], 3), ([1, 2, 5], 0), ([2, 5, 6], 0),
            ([2, 3, 7], 2), ([2, 6, 7], 2), ([0, 3, 4], 1), ([3, 4, 7], 1), ([4, 5, 6], 4), ([4, 6, 7], 4),
        ]
        for p, n in tris:
            self.mesh_triangle(m, p, [n, n, n], material)

    # ---- sky (sky.rs:13-39; loader/src/misc.rs:20-38) ----
    def set_sky(self, texture, sampler_res=(100, 100), material=None):
        if material is None:
            material = self.emissive(texture, 1.0)  # Emit::new(tex, 1.0): loader/src/misc.rs:27
        self.sky = (texture, material, int(sampler_res[0]), int(sampler_res[1]))

    @property
    def n_primitives(self):
        n = len(self.prim_records)
        if self.prim_array is not None:
            n += self.prim_array.shape[0]
        return n

    def desc(self):
        """Build the rt_scene_desc; the returned object keeps every buffer alive."""
        d = abi.SceneDesc()
        d.abi_version = abi.RT_ABI_VERSION
        keep = []

        tex = (abi.TextureDesc * max(1, len(self.textures)))(*self.textures)
        mats = (abi.MaterialDesc * max(1, len(self.materials)))(*self.materials)
        keep += [tex, mats]
        d.textures, d.n_textures = tex, len(self.textures)
        d.materials, d.n_materials = mats, len(self.materials)

        meshes = (abi.MeshDesc * max(1, len(self.meshes)))()
        for i, (v, n) in enumerate(self.meshes):
            meshes[i].vertices = v.ctypes.data_as(C.POINTER(C.c_float))
            meshes[i].n_vertices = v.shape[0]
            meshes[i].normals = n.ctypes.data_as(C.POINTER(C.c_float))
            meshes[i].n_normals = n.shape[0]
        keep.append(meshes)
        d.meshes, d.n_meshes = meshes, len(self.meshes)

        if self.prim_array is not None:
            rec = self.prim_array
            if self.prim_records:
                tail = np.frombuffer(b"".join(bytes(p) for p in self.prim_records), dtype=np.uint32).reshape(-1, 10)
                rec = np.ascontiguousarray(np.concatenate([rec, tail], axis=0))
            keep.append(rec)
            d.primitives = rec.ctypes.data_as(C.POINTER(abi.PrimitiveDesc))
            d.n_primitives = rec.shape[0]
        else:
            prims = (abi.PrimitiveDesc * max(1, len(self.prim_records)))(*self.prim_records)
            keep.append(prims)
            d.primitives, d.n_primitives = prims, len(self.prim_records)

        tris = (abi.TriangleData * max(1, len(self.triangles)))(*self.triangles)
        keep.append(tris)
        d.triangles, d.n_triangles = tris, len(self.triangles)

        if self.sky is None:
            raise ValueError("scene has no sky (call set_sky)")
        d.sky.texture, d.sky.material, d.sky.sampler_res_x, d.sky.sampler_res_y = self.sky
        d.split_type = self.split_type
        d._keep = (keep, self._keep, self.meshes)
        return d
