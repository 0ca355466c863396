"""ctypes mirror of include/rt_hip.h (the C ABI of librt_hip.so).

Field order and types must match the header exactly; tests/test_abi.py checks the struct
sizes against the values the C compiler reports.
"""
import ctypes as C

RT_ABI_VERSION = 2
RT_GATHER_NONE, RT_GATHER_RCCL, RT_GATHER_PEER, RT_GATHER_PEER_STAGED, RT_GATHER_SAME_DEVICE = 0, 1, 2, 3, 4

RT_OK = 0
RT_ERR_INVALID_ARGUMENT = -1
RT_ERR_NO_DEVICE = -2
RT_ERR_HIP = -3
RT_ERR_OUT_OF_MEMORY = -4
RT_ERR_UNSUPPORTED = -5

# enum AllTextures (textures/mod.rs:18-25)
RT_TEX_CHECKERED, RT_TEX_SOLID, RT_TEX_IMAGE, RT_TEX_LERP, RT_TEX_PERLIN = range(5)
# enum AllMaterials (materials/mod.rs:18-25)
RT_MAT_EMIT, RT_MAT_LAMBERTIAN, RT_MAT_TROWBRIDGE_REITZ, RT_MAT_REFLECT, RT_MAT_REFRACT = range(5)
# enum AllPrimitives (primitives/mod.rs:14-19)
RT_PRIM_SPHERE, RT_PRIM_TRIANGLE, RT_PRIM_MESH_TRIANGLE = range(3)
# enum SplitType (acceleration/split.rs:34-45)
RT_SPLIT_SAH, RT_SPLIT_MIDDLE, RT_SPLIT_EQUAL_COUNTS = range(3)
# enum RenderMethod (samplers/mod.rs:43-47)
RT_METHOD_NAIVE, RT_METHOD_MIS = range(2)
RT_LAYOUT_FRAME, RT_LAYOUT_SHARD = range(2)
RT_TUNE_TRAVERSAL, RT_TUNE_FEATURE_SET, RT_TUNE_SCENE_IN_LDS, RT_TUNE_SCHEDULE, RT_TUNE_WALK, RT_TUNE_STACK_CAP, RT_TUNE_EXCHANGE = range(7)

NO_INDEX = 0xFFFFFFFFFFFFFFFF  # usize::MAX
RT_DEVICE_NONE = -1  # rt_scene_create: Bvh::new on the host only (no GPU touched, nothing can be rendered)

f32x3 = C.c_float * 3


class TextureDesc(C.Structure):
    _fields_ = [
        ("type", C.c_int32),
        ("colour_one", f32x3),
        ("colour_two", f32x3),
        ("image_rgb", C.POINTER(C.c_float)),
        ("image_width", C.c_uint32),
        ("image_height", C.c_uint32),
        ("perlin_ran_vecs", C.POINTER(C.c_float)),
        ("perlin_perm", C.POINTER(C.c_uint32)),
    ]


class MaterialDesc(C.Structure):
    _fields_ = [
        ("type", C.c_int32),
        ("texture", C.c_uint32),
        ("param", C.c_float),
        ("ior", f32x3),
        ("metallic", C.c_float),
    ]


class _Sphere(C.Structure):
    _fields_ = [("centre", f32x3), ("radius", C.c_float)]


class _MeshTriangle(C.Structure):
    _fields_ = [("mesh", C.c_uint32), ("point_indices", C.c_uint32 * 3), ("normal_indices", C.c_uint32 * 3)]


class _Triangle(C.Structure):
    _fields_ = [("data", C.c_uint64)]


class _PrimUnion(C.Union):
    _fields_ = [("sphere", _Sphere), ("mesh_triangle", _MeshTriangle), ("triangle", _Triangle)]


class PrimitiveDesc(C.Structure):
    _fields_ = [("type", C.c_int32), ("material", C.c_uint32), ("u", _PrimUnion)]


class TriangleData(C.Structure):
    _fields_ = [("points", C.c_float * 9), ("normals", C.c_float * 9)]


class MeshDesc(C.Structure):
    _fields_ = [
        ("vertices", C.POINTER(C.c_float)),
        ("n_vertices", C.c_uint64),
        ("normals", C.POINTER(C.c_float)),
        ("n_normals", C.c_uint64),
    ]


class SkyDesc(C.Structure):
    _fields_ = [
        ("texture", C.c_uint32),
        ("material", C.c_uint32),
        ("sampler_res_x", C.c_uint32),
        ("sampler_res_y", C.c_uint32),
    ]


class SceneDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("n_textures", C.c_uint32),
        ("textures", C.POINTER(TextureDesc)),
        ("n_materials", C.c_uint32),
        ("n_meshes", C.c_uint32),
        ("materials", C.POINTER(MaterialDesc)),
        ("meshes", C.POINTER(MeshDesc)),
        ("n_primitives", C.c_uint64),
        ("primitives", C.POINTER(PrimitiveDesc)),
        ("n_triangles", C.c_uint64),
        ("triangles", C.POINTER(TriangleData)),
        ("sky", SkyDesc),
        ("split_type", C.c_int32),
    ]


class Camera(C.Structure):
    _fields_ = [("origin", f32x3), ("lower_left", f32x3), ("horizontal", f32x3), ("vertical", f32x3)]


class RenderOpts(C.Structure):
    _fields_ = [
        ("width", C.c_uint64),
        ("height", C.c_uint64),
        ("samples_per_pixel", C.c_uint64),
        ("sample_begin", C.c_uint64),
        ("seed", C.c_uint64),
        ("render_method", C.c_int32),
        ("max_depth", C.c_uint32),
        ("rr_threshold", C.c_uint32),
        ("shard_index", C.c_uint32),
        ("shard_count", C.c_uint32),
        ("tile_width", C.c_uint32),
        ("tile_height", C.c_uint32),
        ("output_layout", C.c_int32),
        ("sample_split", C.c_uint32),
        ("reserved0", C.c_uint32),
    ]


class HitRecord(C.Structure):
    _fields_ = [
        ("t", C.c_float),
        ("point", f32x3),
        ("error", f32x3),
        ("normal", f32x3),
        ("uv", C.c_float * 2),
        ("has_uv", C.c_int32),
        ("out", C.c_int32),
        ("material", C.c_uint32),
        ("found", C.c_uint32),
        ("index", C.c_uint64),
    ]


class RayDesc(C.Structure):
    _fields_ = [("origin", f32x3), ("direction", f32x3)]


class BvhNode(C.Structure):
    _fields_ = [
        ("min", f32x3),
        ("max", f32x3),
        ("children", C.c_int64 * 2),
        ("primitive_offset", C.c_uint64),
        ("number_primitives", C.c_uint64),
    ]


# sizes the C compiler must agree with (x86-64 SysV); checked by tests/test_abi.py
class SamplerProgressC(C.Structure):  # rt_sampler_progress
    _fields_ = [
        ("samples_completed", C.c_uint64),
        ("rays_shot", C.c_uint64),
        ("current_image", C.POINTER(C.c_float)),
        ("n_floats", C.c_uint64),
    ]


class LaunchInfo(C.Structure):  # rt_launch_info
    _fields_ = [
        ("method", C.c_int32),
        ("pruned", C.c_int32),
        ("fine", C.c_int32),
        ("sky_in_lds", C.c_int32),
        ("scene_in_lds", C.c_int32),
        ("feature_set", C.c_int32),
        ("block_threads", C.c_uint32),
        ("n_blocks", C.c_uint32),
        ("blocks_per_cu", C.c_uint32),
        ("waves_per_simd", C.c_uint32),
        ("lds_bytes", C.c_uint32),
        ("n_cus", C.c_uint32),
        ("sample_split", C.c_uint32),
        ("reserved", C.c_uint32),
        ("n_items", C.c_uint64),
        ("kernel", C.c_char * 160),
    ]


# rt_presentation_update: int (*)(void *data, const rt_sampler_progress *, uint64_t samples_done)
PresentationUpdate = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(SamplerProgressC), C.c_uint64)

EXPECTED_SIZES = {
    "rt_texture_desc": (TextureDesc, 64),
    "rt_material_desc": (MaterialDesc, 28),
    "rt_primitive_desc": (PrimitiveDesc, 40),
    "rt_triangle_data": (TriangleData, 72),
    "rt_mesh_desc": (MeshDesc, 32),
    "rt_sky_desc": (SkyDesc, 16),
    "rt_scene_desc": (SceneDesc, 96),
    "rt_camera": (Camera, 48),
    "rt_render_opts": (RenderOpts, 80),
    "rt_hit_record": (HitRecord, 72),
    "rt_ray_desc": (RayDesc, 24),
    "rt_bvh_node": (BvhNode, 56),
    "rt_sampler_progress": (SamplerProgressC, 32),
    "rt_launch_info": (LaunchInfo, 224),
}

# every symbol include/rt_hip.h declares
EXPORTED_SYMBOLS = [
    "rt_last_error",
    "rt_abi_version",
    "rt_device_count",
    "rt_render_opts_default",
    "rt_camera_new",
    "rt_scene_create",
    "rt_scene_create_multi",
    "rt_scene_device_count",
    "rt_scene_destroy",
    "rt_scene_counts",
    "rt_scene_get_nodes",
    "rt_scene_get_primitive_order",
    "rt_scene_get_lights",
    "rt_scene_wide_info",
    "rt_scene_get_wide_nodes",
    "rt_scene_get_leaf_boxes",
    "rt_scene_get_wide_nodes_compact",
    "rt_scene_auto_sample_split",
    "rt_scene_gather_info",
    "rt_rccl_probe",
    "rt_selftest_division",
    "rt_scene_get_leaf_boxes_compact",
    "rt_scene_set_traversal",
    "rt_scene_set_tuning",
    "rt_render",
    "rt_sample_image",
    "rt_render_device",
    "rt_render_output_floats",
    "rt_shard_pixel_order",
    "rt_last_kernel_ms",
    "rt_last_launch_info",
    "rt_output_rgb8",
    "rt_output_save",
    "rt_output_rgb8_device",
    "rt_render_rgb8",
    "rt_check_hit",
    "rt_check_hit_index",
    "rt_selftest_lean",
]


def default_render_opts(width=1920, height=1080, spp=128, method=RT_METHOD_MIS, seed=1):
    """RenderOptions::default() (samplers/mod.rs:31-41) + MAX_DEPTH/RR consts (integrators/mod.rs:7-8)."""
    o = RenderOpts()
    o.width, o.height = width, height
    o.samples_per_pixel = spp
    o.sample_begin = 0
    o.seed = seed
    o.render_method = method
    o.max_depth = 50
    o.rr_threshold = 3
    o.shard_index, o.shard_count = 0, 1
    o.tile_width = o.tile_height = 0
    o.output_layout = RT_LAYOUT_FRAME
    o.sample_split = 1
    o.reserved0 = 0
    return o
