/*
 * rt_hip.h -- C ABI of the MI355X path-tracing back end (librt_hip.so).
 *
 * The reference (nonl4331/raytracing-rust) has no FFI; the narrowest seam that carries
 * whole-image work is the trait method
 *     Sampler::sample_image(RenderOptions, &Camera, &AccelerationStructure, callback)
 *     crates/implementations/src/samplers/mod.rs:7-20, implemented by RandomSampler
 *     (samplers/random_sampler.rs:10-99) and called only from Scene::render (src/scene.rs:35-42).
 * This header is what a Rust `extern "C"` block would bind to put a GPU `impl Sampler`
 * behind that seam (binding text: INTEGRATION.md).  Plain pointers and sizes only.
 *
 * Every POD below mirrors a reference type; the citation next to it is the type it
 * replaces.  All floats are f32 (rt_core/src/lib.rs:23-32), vectors are 3 packed floats
 * (Vec3 is #[repr(C)] {x,y,z}: rt_core/src/vec.rs:108-114).
 *
 * The random stream and the elementary functions are fixed by include/rt_detmath.h, so
 * that rt_render(seed) is reproducible (the reference itself is unseeded).
 *
 * Error convention: functions return RT_OK (0) or a negative rt_status; the message is
 * available from rt_last_error() (thread local).  Nothing aborts, nothing throws.
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 4): rt_render_opts.sample_split = 0 means "the library picks" (rt_scene_auto_sample_split) and S > 1 adds chunk SUMS
 * (rounds 1 - 3: 0 was the sequential fold, chunks were running means); new entry points rt_scene_auto_sample_split,
 * rt_scene_gather_info, rt_rccl_probe, rt_selftest_division, rt_scene_get_wide_nodes_compact, rt_scene_get_leaf_boxes_compact */
#define RT_ABI_VERSION 2u

typedef enum rt_status {
	RT_OK = 0,
	RT_ERR_INVALID_ARGUMENT = -1,
	RT_ERR_NO_DEVICE = -2,
	RT_ERR_HIP = -3,
	RT_ERR_OUT_OF_MEMORY = -4,
	RT_ERR_UNSUPPORTED = -5
} rt_status;

/* ---- textures: enum AllTextures, crates/implementations/src/textures/mod.rs:18-25 ---- */
typedef enum rt_texture_type {
	RT_TEX_CHECKERED = 0, /* CheckeredTexture  textures/mod.rs:27-73  */
	RT_TEX_SOLID = 1,     /* SolidColour       textures/mod.rs:182-200 */
	RT_TEX_IMAGE = 2,     /* ImageTexture      textures/mod.rs:202-266 (decoded pixels only) */
	RT_TEX_LERP = 3,      /* Lerp              textures/mod.rs:268-291 */
	RT_TEX_PERLIN = 4     /* Perlin            textures/mod.rs:75-180 */
} rt_texture_type;

typedef struct rt_texture_desc {
	int32_t type;
	float colour_one[3]; /* solid: colour; lerp/checkered: colour_one (".ssml" `primary`) */
	float colour_two[3]; /* lerp/checkered: colour_two (`secondary`) */
	/* image: row-major RGB f32 as `to_rgb32f()` yields, true width/height in pixels
	 * (the reference stores width-1/height-1 in `dim`, textures/mod.rs:232) */
	const float *image_rgb;
	uint32_t image_width;
	uint32_t image_height;
	/* perlin: ran_vecs[256][3], then perm_x[256], perm_y[256], perm_z[256] */
	const float *perlin_ran_vecs;
	const uint32_t *perlin_perm;
} rt_texture_desc;

/* ---- materials: enum AllMaterials, crates/implementations/src/materials/mod.rs:18-25 ---- */
typedef enum rt_material_type {
	RT_MAT_EMIT = 0,             /* materials/emissive.rs:5-39         param = strength */
	RT_MAT_LAMBERTIAN = 1,       /* materials/lambertian.rs:5-51       param = albedo   */
	RT_MAT_TROWBRIDGE_REITZ = 2, /* materials/trowbridge_reitz.rs:5-92 param = the stored
	                                `alpha` field, i.e. roughness*roughness (:17-24) */
	RT_MAT_REFLECT = 3,          /* materials/reflect.rs:7-43          param = fuzz     */
	RT_MAT_REFRACT = 4           /* materials/refract.rs:8-56          param = eta      */
} rt_material_type;

typedef struct rt_material_desc {
	int32_t type;
	uint32_t texture; /* index into rt_scene_desc.textures */
	float param;
	float ior[3];   /* TrowbridgeReitz only */
	float metallic; /* TrowbridgeReitz only */
} rt_material_desc;

/* ---- primitives: enum AllPrimitives, crates/implementations/src/primitives/mod.rs:14-19 ---- */
typedef enum rt_primitive_type {
	RT_PRIM_SPHERE = 0,       /* primitives/sphere.rs:9-27   */
	RT_PRIM_TRIANGLE = 1,     /* primitives/triangle.rs:11-29 */
	RT_PRIM_MESH_TRIANGLE = 2 /* primitives/triangle.rs:31-56 */
} rt_primitive_type;

typedef struct rt_primitive_desc {
	int32_t type;
	uint32_t material; /* index into rt_scene_desc.materials */
	union {
		struct {
			float centre[3];
			float radius;
		} sphere;
		struct {
			uint32_t mesh; /* index into rt_scene_desc.meshes */
			uint32_t point_indices[3];
			uint32_t normal_indices[3];
		} mesh_triangle;
		struct {
			uint64_t data; /* index into rt_scene_desc.triangles */
		} triangle;
	} u;
} rt_primitive_desc;

/* Triangle { points: [Vec3;3], normals: [Vec3;3] }  primitives/triangle.rs:11-15 */
typedef struct rt_triangle_data {
	float points[9];
	float normals[9];
} rt_triangle_data;

/* MeshData { vertices: Vec<Vec3>, normals: Vec<Vec3> }  primitives/triangle.rs:58-67 */
typedef struct rt_mesh_desc {
	const float *vertices;
	uint64_t n_vertices;
	const float *normals;
	uint64_t n_normals;
} rt_mesh_desc;

/* Sky::new(texture, mat, sampler_res)  crates/implementations/src/sky.rs:13-39 */
typedef struct rt_sky_desc {
	uint32_t texture;
	uint32_t material; /* the loader makes Emit(texture, 1.0): loader/src/misc.rs:27 */
	uint32_t sampler_res_x;
	uint32_t sampler_res_y; /* (0,0) disables importance sampling: sky.rs:61-63 */
} rt_sky_desc;

/* enum SplitType  acceleration/split.rs:34-45 */
typedef enum rt_split_type { RT_SPLIT_SAH = 0, RT_SPLIT_MIDDLE = 1, RT_SPLIT_EQUAL_COUNTS = 2 } rt_split_type;

/* Everything Bvh::new(primitives, sky, split_type) consumes (acceleration/mod.rs:58-93),
 * flattened out of the Region arena (crates/region) the reference keeps it in. */
typedef struct rt_scene_desc {
	uint32_t abi_version; /* RT_ABI_VERSION */
	uint32_t n_textures;
	const rt_texture_desc *textures;
	uint32_t n_materials;
	uint32_t n_meshes;
	const rt_material_desc *materials;
	const rt_mesh_desc *meshes;
	uint64_t n_primitives;
	const rt_primitive_desc *primitives;
	uint64_t n_triangles;
	const rt_triangle_data *triangles;
	rt_sky_desc sky;
	int32_t split_type;
} rt_scene_desc;

/* SimpleCamera's four ray-generating fields  crates/implementations/src/camera.rs:6-17 */
typedef struct rt_camera {
	float origin[3];
	float lower_left[3];
	float horizontal[3];
	float vertical[3];
} rt_camera;

/* enum RenderMethod  samplers/mod.rs:43-47 */
typedef enum rt_render_method { RT_METHOD_NAIVE = 0, RT_METHOD_MIS = 1 } rt_render_method;

typedef enum rt_output_layout {
	RT_LAYOUT_FRAME = 0, /* width*height*3 floats, row-major, y down, RGB (SamplerProgress.current_image) */
	RT_LAYOUT_SHARD = 1  /* only this shard's pixels, packed in work order (see rt_shard_pixel_order) */
} rt_output_layout;

/* RenderOptions (samplers/mod.rs:22-41) plus what the reference hard-codes or lacks:
 * MAX_DEPTH / RUSSIAN_ROULETTE_THRESHOLD (integrators/mod.rs:7-8), a seed, a sample
 * window for resume/progressive batches, and the tile sharding used across GPUs. */
typedef struct rt_render_opts {
	uint64_t width;
	uint64_t height;
	uint64_t samples_per_pixel; /* passes rendered by THIS call */
	uint64_t sample_begin;      /* index of the first pass (0 for a fresh render) */
	uint64_t seed;
	int32_t render_method;  /* rt_render_method; reference default MIS (src/parameters.rs:37-38) */
	uint32_t max_depth;     /* reference value 50 */
	uint32_t rr_threshold;  /* reference value 3  */
	uint32_t shard_index;   /* this GPU's shard, 0 <= shard_index < shard_count */
	uint32_t shard_count;   /* 1 = whole image */
	uint32_t tile_width;    /* shard granularity in pixels; 0 = default (8) */
	uint32_t tile_height;   /* 0 = default (8) */
	int32_t output_layout;  /* rt_output_layout */
	/* 1 (default): a pixel's passes are folded strictly in pass order, `mean += (pass-mean)/i`,
	 * the reference's accumulation (src/main.rs:179-185).  0 = automatic: the power of two <= 64 that gives this device
	 * >= 64 work items per resident lane (16 for one GPU at 1080p x 1024 passes; rt_scene_auto_sample_split states the rule,
	 * rt_last_launch_info reports the choice).  The CPU checker (oracle/) takes explicit splits only: 0 is refused there.
	 * S > 1: the passes of a pixel are split
	 * into S contiguous chunks [floor(c*spp/S), floor((c+1)*spp/S)); each chunk's passes are SUMMED in pass order (f32, from +0),
	 * the chunk sums are added in chunk order and the total is divided by spp once: (sum_c sum_c) / spp.  Same samples, same
	 * streams, a different association than the running mean: the image changes at the 1e-7 level (tests hold the whole 1080p
	 * frame of the BASELINE configs to < 1e-4 of the sequential fold).  It exists for parallelism and for balance: without it a
	 * render cannot use more lanes than it has pixels (8 GPUs at 1080p have one pixel per lane), and even one GPU ends a 1080p
	 * frame with most of its lanes idle while the last whole pixels finish (7 - 29 % of the launch on the BASELINE workloads).
	 * (Rounds 1 - 3 folded each chunk as a running mean and combined sum_c mean_c * n_c: three IEEE divisions per sample that the
	 * chunked form has no use for -- the reference's own fold is the S = 1 case, which is unchanged.) */
	uint32_t sample_split;
	uint32_t reserved0;
} rt_render_opts;

void rt_render_opts_default(rt_render_opts *opts);

/* One hit record = Hit + material + primitive index
 * (rt_core/src/primitive.rs:3-16, the tuple Bvh::check_hit returns acceleration/mod.rs:265-298) */
typedef struct rt_hit_record {
	float t;
	float point[3];
	float error[3];
	float normal[3];
	float uv[2];
	int32_t has_uv;
	int32_t out;
	uint32_t material;
	uint32_t found;  /* check_hit: always 1; check_hit_index: 0 when the call returns None */
	uint64_t index;  /* primitive index in BVH order; UINT64_MAX = sky (usize::MAX) */
} rt_hit_record;

/* A ray as handed to Ray::new(origin, direction, time)  rt_core/src/ray.rs:13-46 */
typedef struct rt_ray_desc {
	float origin[3];
	float direction[3];
} rt_ray_desc;

/* Node { bounds, children, primitive_offset, number_primitives }  acceleration/mod.rs:331-336 */
typedef struct rt_bvh_node {
	float min[3];
	float max[3];
	int64_t children[2]; /* -1,-1 = None (leaf) */
	uint64_t primitive_offset;
	uint64_t number_primitives;
} rt_bvh_node;

typedef struct rt_scene rt_scene;

/* ---- library ---- */
const char *rt_last_error(void);
uint32_t rt_abi_version(void);
int rt_device_count(void);

/* ---- SimpleCamera::new  camera.rs:20-54 (aspect is 16/9 in the loader: loader/src/misc.rs:15) ---- */
int rt_camera_new(rt_camera *out, const float origin[3], const float lookat[3], const float vup[3],
                  float fov_degrees, float aspect_ratio, float aperture, float focus_dist);

/* ---- Bvh::new + upload: builds the BVH on the host exactly as acceleration/mod.rs:58-160 and
 * acceleration/split.rs:78-210 do, builds the sky tables (textures/mod.rs:32-50,
 * statistics/distributions.rs:12-99), and lays everything out in the HBM of `device`. ---- */
int rt_scene_create(const rt_scene_desc *desc, int device, rt_scene **out);
/* device = RT_DEVICE_NONE: Bvh::new on the host only.  rt_scene_counts / rt_scene_get_nodes /
 * rt_scene_get_primitive_order / rt_scene_get_lights work; every call that would render or trace
 * returns RT_ERR_NO_DEVICE (there is no CPU fallback). */
#define RT_DEVICE_NONE (-1)
void rt_scene_destroy(rt_scene *scene);

/* ---- one scene on SEVERAL GPUs of the node, behind the same calls.  The reference's caller is one process with one
 * `Scene::render` (src/scene.rs:35-42) whose sampler partitions the frame into independent chunks
 * (samplers/random_sampler.rs:45-52); this is that partition across devices: Bvh::new runs once on the host, the scene is
 * replicated into the HBM of every listed device, tile t of the frame (8 x 8 pixels unless opts say otherwise) belongs to
 * device t % n_devices.  rt_render / rt_render_device / rt_render_rgb8 / rt_sample_image on the returned handle render every
 * device's tiles concurrently (one stream per device, no host thread per device, nothing synchronises with the host inside
 * rt_render_device), gather the shards into the HBM of devices[0] -- grouped ncclSend / ncclRecv (RCCL is loaded on demand
 * when the devices are distinct; hipMemcpyPeerAsync if it is not usable; a plain copy between members on the same device) --
 * and write them into the frame there; `rays_shot` is the sum over the devices.  The frame, `d_out_rgb`, `d_rays_shot` and
 * `hip_stream` belong to devices[0].  opts->shard_count must be 1 and the layout RT_LAYOUT_FRAME: the scene shards by itself.
 * opts->sample_split: 1 = every pixel folded strictly in pass order, so the frame equals the single-device frame bit for bit
 * (a device then cannot use more lanes than it owns pixels); 0 = automatic (rt_scene_auto_sample_split below: 16, 32, 64, 64 for
 * 1, 2, 4, 8 GPUs at 1080p x 1024 passes); S > 1 as documented at rt_render_opts.  A list of ONE device is
 * rt_scene_create.  The same device may be listed more than once (two members then share that GPU).  rt_check_hit[_index]
 * and the introspection calls use devices[0]. ---- */
int rt_scene_create_multi(const rt_scene_desc *desc, const int *devices, uint32_t n_devices, rt_scene **out);
int rt_scene_device_count(const rt_scene *scene, uint32_t *n_devices); /* 0 for a host-only scene */
/* How a multi-device scene moves its members' shards into devices[0] -- decided by rt_scene_create_multi (never by a render: no
 * render initialises a communicator, opens a library or changes peer mappings, so rt_render_device is free of host
 * synchronisation and HIP-graph capturable from its first call on):
 *   RT_GATHER_NONE          one device, nothing to gather
 *   RT_GATHER_RCCL          distinct devices: ncclCommInitAll at creation, grouped ncclSend / ncclRecv per frame.  The RCCL used
 *                           is the file RT_HIP_RCCL_LIB names, else a copy already loaded into the process (e.g. PyTorch's),
 *                           else the system's librccl.so.1
 *   RT_GATHER_PEER          hipMemcpyPeerAsync with peer access devices[0] <- member enabled at creation: RT_HIP_NO_RCCL is set,
 *                           RCCL is missing / lacks a symbol / refused the device list, or the list repeats a device
 *   RT_GATHER_PEER_STAGED   the same, but at least one pair has no peer access: the runtime stages those copies through the
 *                           host (correct, slower; the note names the devices)
 *   RT_GATHER_SAME_DEVICE   every member shares devices[0]: plain device-to-device copies
 * `note` (may be NULL) receives a NUL-terminated sentence saying why.  hipDeviceEnablePeerAccess failing on a pair that reports
 * peer access as possible fails rt_scene_create_multi (RT_ERR_HIP, rt_last_error names the pair).
 * STATUS: on hardware only RT_GATHER_SAME_DEVICE has run, and RT_GATHER_RCCL's call sequence against a stand-in library on one GPU
 * (tests/cpp/fake_rccl.cpp); RCCL itself and peer copies between two GPUs have not (the test pool hands out one-GPU boxes). */
typedef enum rt_gather_mode { RT_GATHER_NONE = 0, RT_GATHER_RCCL = 1, RT_GATHER_PEER = 2, RT_GATHER_PEER_STAGED = 3, RT_GATHER_SAME_DEVICE = 4 } rt_gather_mode;
int rt_scene_gather_info(const rt_scene *scene, int *mode, char *note, uint64_t note_capacity);
/* Which RCCL rt_scene_create_multi would bind, without touching a GPU: *usable = 1 when a library with all six entry points was
 * found (and RT_HIP_NO_RCCL is not set); `note` names the file and how it was found, or what is missing. */
int rt_rccl_probe(int *usable, char *note, uint64_t note_capacity);
/* What opts->sample_split = 0 (automatic) resolves to for these options on this scene -- the ONE rule the library, bench.py and the
 * tests share: the power of two S <= 64 that gives a device >= 64 work items (pixels x S) per resident lane (CUs x 1024), with
 * chunks of at least 16 passes when the device renders the whole frame and at least 4 when the frame is sharded (over the
 * scene's own devices, or opts->shard_count > 1).  rt_last_launch_info reports the split a render really used. */
int rt_scene_auto_sample_split(const rt_scene *scene, const rt_render_opts *opts, uint32_t *split);

/* introspection of what Bvh::new produced (for parity tests against the oracle) */
int rt_scene_counts(const rt_scene *scene, uint64_t *n_nodes, uint64_t *n_primitives, uint64_t *n_lights);
int rt_scene_get_nodes(const rt_scene *scene, rt_bvh_node *out, uint64_t capacity);
/* primitive_order[i] = index in rt_scene_desc.primitives of the primitive at BVH slot i
 * (the permutation sort_by_indices applies, acceleration/mod.rs:79-82) */
int rt_scene_get_primitive_order(const rt_scene *scene, uint64_t *out, uint64_t capacity);
int rt_scene_get_lights(const rt_scene *scene, uint64_t *out, uint64_t capacity); /* Bvh.lights :84-88 */
/* The wide tree (four-child quantised regrouping of the reference tree that pruned walks of regular rays descend;
 * see rt_types.h DevNodeQ4 and rt_intersect.h) for inspection by tests: n_wide_nodes 64-byte records
 * { float origin[3]; uint32 exps; uint32 qlo[3]; uint32 qhi[3]; uint32 child[4]; uint32 pad[2]; },
 * child: bit 31 set = leaf (bits 26-30 primitive count or 0 = big leaf, bits 0-25 first slot), 0x7FFFFFFE = absent,
 * otherwise a wide-node index; root_ref in the same encoding; stack_depth = traversal stack entries the scene needs.
 * n_wide_nodes = 0: no wide tree (non-finite or huge bounds, or a single leaf). */
/* (This is the EXPLICIT form, for inspection.  The kernels fetch a compact re-encoding of the same nodes -- inner children are
 * consecutive nodes, leaf children consecutive leaf indices, so the four references fold into two words and a node step reads
 * 48 bytes instead of 64; every node is decoded back to this form when the scene is built: csrc/rt_types.h, csrc/rt_build.cpp.) */
int rt_scene_wide_info(const rt_scene *scene, uint64_t *n_wide_nodes, uint32_t *root_ref, uint32_t *stack_depth);
int rt_scene_get_wide_nodes(const rt_scene *scene, void *out, uint64_t capacity_nodes);
/* exact leaf boxes of the wide walk, one { float lo[3], pad, hi[3], pad } per primitive slot (meaningful at the first
 * slot of every leaf) */
int rt_scene_get_leaf_boxes(const rt_scene *scene, float *out, uint64_t capacity_slots);
/* ... and the COMPACT form itself, byte for byte what the kernels fetch (csrc/rt_types.h DevNodeQ4): n_wide_nodes 64-byte records
 * of the same layout with  child[0] = first inner child | leaf mask << 26,  child[1] = first leaf index | present mask << 26,
 * the 2-bit per-child offsets in the top byte of `exps`; and the exact leaf boxes BY LEAF INDEX, { lo[3], ref, hi[3], pad } with
 * ref = the leaf's own reference (bit pattern).  *n_leaves receives the leaf count (may be asked for with out = NULL).
 * tests/test_host_bvh.py walks this form in numpy the way rt_intersect.h descend4 does. */
int rt_scene_get_wide_nodes_compact(const rt_scene *scene, void *out, uint64_t capacity_nodes);
int rt_scene_get_leaf_boxes_compact(const rt_scene *scene, float *out, uint64_t capacity_leaves, uint64_t *n_leaves);

/* How the BVH is walked: -1 automatic (default), 0 exhaustive = every AABB-hit node and every
 * primitive of every hit leaf, the reference's own amount of work (acceleration/mod.rs:199-224,
 * 270-293), 1 = near-first with t-pruning.  All modes return the same hits. */
int rt_scene_set_traversal(rt_scene *scene, int mode);
/* Other knobs that change HOW the kernels run, never what they return (used by the parity tests to
 * cover every kernel variant):
 *   RT_TUNE_TRAVERSAL     as rt_scene_set_traversal
 *   RT_TUNE_FEATURE_SET   0 spheres-only, 1 + triangles and emissive primitives, 2 every material /
 *                         texture; the library picks the smallest that covers the scene, a caller may
 *                         only raise it.  A scene whose tree is ONE node over two leaves of one sphere each
 *                         (rtweekend1.ssml) runs, where set 0 would run under the exhaustive coarse schedule,
 *                         kernels compiled for exactly that tree (rt_launch_info.feature_set = 3); naming a
 *                         set here, 0 included, turns that off (same pixels: a test renders both)
 *   RT_TUNE_SCENE_IN_LDS  1 (default): tiny scenes are staged whole into LDS; 0: read from HBM/L2
 *   RT_TUNE_SCHEDULE      -1 automatic, 0 coarse (two voted super-phases), 1 fine (every step of the
 *                         per-lane state machine is voted; implies the pruned walk)
 *   RT_TUNE_WALK          0 automatic: pruned walks use the wide (four-child, 128-byte-node) regrouping of the
 *                         reference tree for regular rays and the two-child tree for the rest; 1: the two-child
 *                         tree for every ray
 *   RT_TUNE_STACK_CAP     0 automatic; n: keep at most n traversal-stack entries per lane in LDS, the rest of the
 *                         tree's worst case in the global overflow area (exercises that path on small trees)
 *   RT_TUNE_EXCHANGE      0 off (default); 1: the waves of a workgroup trade whole lane states (path, pixel, random
 *                         stream) through pools in LDS -- under the coarse schedule with MIS and exhaustive traversal so
 *                         that each super-phase runs on full waves, under the fine schedule so that one wave of a
 *                         512-thread workgroup shades what the other seven walk.  Same pixels either way (measured
 *                         slower on every workload so far); other kernels ignore it */
typedef enum rt_tuning_key { RT_TUNE_TRAVERSAL = 0, RT_TUNE_FEATURE_SET = 1, RT_TUNE_SCENE_IN_LDS = 2, RT_TUNE_SCHEDULE = 3, RT_TUNE_WALK = 4,
                             RT_TUNE_STACK_CAP = 5, RT_TUNE_EXCHANGE = 6 } rt_tuning_key;
int rt_scene_set_tuning(rt_scene *scene, int key, int value);

/* ---- Sampler::sample_image  samplers/random_sampler.rs:10-99 ----
 * Renders opts->samples_per_pixel passes and returns their running mean
 * `mean += (pass - mean) / i`, i = 1..samples_per_pixel -- the accumulation the reference's
 * callback performs on the host after every pass (src/main.rs:175-191) -- and the sum of
 * the integrators' ray counters (SamplerProgress.rays_shot).  Blocking.
 * out_rgb: HOST buffer, layout per opts->output_layout.  rays_shot may be NULL. */
int rt_render(rt_scene *scene, const rt_camera *camera, const rt_render_opts *opts, float *out_rgb,
              uint64_t *rays_shot);

/* Same, asynchronous on a caller-supplied HIP stream, into DEVICE memory of the scene's GPU
 * (d_out_rgb sized by rt_render_output_floats, d_rays_shot one uint64 or NULL).  No host sync. */
int rt_render_device(rt_scene *scene, const rt_camera *camera, const rt_render_opts *opts, float *d_out_rgb,
                     uint64_t *d_rays_shot, void *hip_stream);

/* number of floats rt_render writes for these options (FRAME: w*h*3; SHARD: 3*owned pixels) */
int rt_render_output_floats(const rt_render_opts *opts, uint64_t *n_floats);
/* pixel indices (y*width+x) of the shard's pixels in the order RT_LAYOUT_SHARD packs them */
int rt_shard_pixel_order(const rt_render_opts *opts, uint64_t *out, uint64_t capacity);

/* ---- Sampler::sample_image with its presentation callback (samplers/mod.rs:7-20,
 * samplers/random_sampler.rs:10-99), batched.  The reference renders pass i into one of two
 * SamplerProgress buffers, then hands the OTHER buffer (pass i-1) to the callback
 * `F: Fn(&mut T, &SamplerProgress, u64) -> bool`; `true` cancels and the pass already rendered is
 * dropped; the last image is delivered after the loop and its return value is ignored (:82-98).
 * rt_sample_image keeps that shape with `batch` passes per image (0 = all of them in one batch): batch j+1
 * renders on the GPU while batch j is copied to pinned host memory on a second HIP stream and the
 * callback runs, from two device and two pinned host buffers owned by the scene.  `progress->current_image`
 * is the MEAN of the batch's passes (samples_completed of them; layout per opts->output_layout), valid
 * during the callback only; `samples_done` counts passes delivered so far including this batch.  With
 * batch = 1 the contract is the reference's, image for image.  Blocking; one host thread per scene. ---- */
typedef struct rt_sampler_progress { /* SamplerProgress  samplers/mod.rs:49-53 */
	uint64_t samples_completed;
	uint64_t rays_shot;
	const float *current_image;
	uint64_t n_floats;
} rt_sampler_progress;
typedef int (*rt_presentation_update)(void *data, const rt_sampler_progress *progress, uint64_t samples_done);
int rt_sample_image(rt_scene *scene, const rt_camera *camera, const rt_render_opts *opts, uint64_t batch,
                    rt_presentation_update update, void *data);

/* Milliseconds the GPU spent in the render kernel of the most recent rt_render /
 * rt_render_device on this scene, measured with HIP events on the launch stream
 * (synchronises that stream).  The kernel's launch count is returned through n_launches. */
int rt_last_kernel_ms(rt_scene *scene, float *ms, uint32_t *n_launches);

/* What the most recent rt_render / rt_render_device on this scene actually launched: the kernel
 * instantiation rt_render_device selected (its automatic choices depend on the scene, the method and
 * the occupancy query), its launch geometry and its LDS.  For measurement tools (bench.py prints it);
 * `kernel` is the demangled instantiation name as rocprofv3 reports it.  Does not synchronise. */
typedef struct rt_launch_info {
	int32_t method;         /* rt_render_method */
	int32_t pruned;         /* 1: t-pruned walk, 0: exhaustive (the reference's amount of work) */
	int32_t fine;           /* 1: every phase voted (big trees), 0: two super-phases */
	int32_t sky_in_lds;     /* sky CDF + guide tables staged in LDS */
	int32_t scene_in_lds;   /* whole scene staged in LDS (tiny scenes) */
	int32_t feature_set;    /* 0 spheres-only, 1 simple, 2 full, 3 spheres-only specialised for a two-leaf tree */
	uint32_t block_threads; /* workgroup size */
	uint32_t n_blocks;      /* persistent grid */
	uint32_t blocks_per_cu; /* resident workgroups per CU (occupancy query) */
	uint32_t waves_per_simd;/* blocks_per_cu * block_threads / 256 */
	uint32_t lds_bytes;     /* dynamic LDS per workgroup */
	uint32_t n_cus;
	uint32_t sample_split;
	uint32_t reserved;
	uint64_t n_items;       /* work items of the launch (pixels x sample_split, incl. edge-tile padding) */
	char kernel[160];
} rt_launch_info;
int rt_last_launch_info(const rt_scene *scene, rt_launch_info *out);

/* ---- output stage, the step right after the path: crates/output/src/lib.rs:74-113 save_data_to_image.
 * Host-side (no GPU needed).  rt_output_rgb8 is the reference's pixel conversion
 * `(val.powf(1.0 / gamma) * 255.999) as u8` (`as u8` saturates, NaN -> 0); rt_output_save dispatches on
 * the extension the way save_data_to_image does: .png (stored deflate blocks), .ppm, .bmp (24-bit) and
 * .tiff (one uncompressed strip) receive those RGB8 pixels; .exr receives the float image itself with gamma
 * ignored (lib.rs:99-106), as an uncompressed scanline file with FLOAT channels B, G, R.  jpg/jpeg, which the
 * reference hands to the image crate, and unknown extensions return RT_ERR_UNSUPPORTED. ---- */
int rt_output_rgb8(const float *rgb, uint64_t n_values, float gamma, uint8_t *out);
int rt_output_save(const char *filename, const float *rgb, uint32_t width, uint32_t height, float gamma);

/* The same conversion on the GPU, for frames that are already there (`powf` is include/rt_detmath.h's rt_powf on both
 * sides, so device bytes == host bytes == the oracle's).  rt_output_rgb8_device: d_rgb / d_out are DEVICE pointers on the
 * scene's GPU, asynchronous on hip_stream; any alignment is accepted (16-byte aligned input and 4-byte aligned output take the
 * vectorised kernel).  rt_render_rgb8 = rt_render followed by that conversion and a copy of the
 * BYTES to the host: W*H*3 bytes cross PCIe instead of W*H*12 (6.2 MB instead of 24.9 MB at 1080p).  Blocking. */
int rt_output_rgb8_device(rt_scene *scene, const float *d_rgb, uint64_t n_values, float gamma, uint8_t *d_out, void *hip_stream);
int rt_render_rgb8(rt_scene *scene, const rt_camera *camera, const rt_render_opts *opts, float gamma, uint8_t *out_rgb8,
                   uint64_t *rays_shot);

/* ---- AccelerationStructure::check_hit / check_hit_index for a batch of rays
 * (acceleration/mod.rs:226-298), run on the GPU; host buffers ---- */
int rt_check_hit(rt_scene *scene, const rt_ray_desc *rays, uint64_t n_rays, rt_hit_record *out);
int rt_check_hit_index(rt_scene *scene, const rt_ray_desc *rays, const uint64_t *object_index, uint64_t n_rays,
                       rt_hit_record *out);

/* Division by a constant a launch knows beforehand (image size - 1, sky table resolution, pi, 2 pi): the kernels replace `x / c` by
 * two fma steps on rc = RN(1 / c) where -- and only where -- the host has verified, by enumerating all 2^23 significands of x, that
 * this returns the bits of the division (csrc/rt_build.cpp verified_reciprocal, csrc/rt_lean.h div_by_verified).  This call runs that
 * verification for one divisor: *exact = 1 and *reciprocal = rc, or *exact = 0 (the kernels then keep the plain division).  No GPU. */
int rt_selftest_division(float divisor, float *reciprocal, int *exact);

/* ---- self-test of the kernels' arithmetic.  The render kernels compute `a / b`, `sqrtf` and the elementary functions
 * of include/rt_detmath.h through shorter instruction sequences wherever the operands make the omitted steps the identity
 * (raytracing-rust_amd/csrc/rt_lean.h).  This call runs both forms side by side ON `device` over 262 144 x n_per_thread random
 * and edge-case operands per class and counts results whose bits differ (NaN == NaN): classes 0 division, 1 division with a
 * zero / infinite / NaN numerator, 2 reciprocal, 3 vector / scalar, 4 square root, 5 sin + cos, 6 acos, 7 atan2,
 * 8 Ray::new (rt_core/src/ray.rs:13-46).  Every count must be zero. ---- */
#define RT_SELFTEST_LEAN_CLASSES 9
int rt_selftest_lean(int device, uint64_t n_per_thread, uint64_t seed, uint64_t mismatches[RT_SELFTEST_LEAN_CLASSES]);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
