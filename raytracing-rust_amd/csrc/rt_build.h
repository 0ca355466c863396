// rt_build.h -- host-side scene build: Bvh::new of the reference plus the re-layout for HBM.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rt_hip.h"
#include "rt_types.h"

namespace rt {

// Node of acceleration/mod.rs:331-336, kept on the host for rt_scene_get_nodes()
struct HostNode {
	float min[3], max[3];
	int64_t child[2]; // -1 = None
	uint64_t primitive_offset, number_primitives;
};

struct HostScene {
	// what Bvh::new produces
	std::vector<HostNode> nodes;
	std::vector<uint64_t> primitive_order; // slot -> index in rt_scene_desc.primitives
	std::vector<uint64_t> lights;          // Bvh.lights (slots)
	// device images
	std::vector<DevNode> dev_nodes;
	std::vector<DevNodeQ4> dev_nodes4; // wide tree with explicit child references (empty: not built): what rt_scene_get_wide_nodes shows
	std::vector<DevLeafBox> leaf_box;  // exact leaf boxes by first slot (only with the wide tree): what rt_scene_get_leaf_boxes shows
	// ... and as the kernels read it (rt_types.h "compact wide node"): the same nodes with the four child references folded into
	// two words, so a node step fetches three 16-byte pieces instead of four, and the leaf boxes by LEAF INDEX, each carrying
	// its leaf's (first slot, count) reference
	std::vector<DevNodeQ4> dev_nodes4c;
	std::vector<DevLeafBox> leaf_box_c;
	uint32_t root4_ref = 0;
	std::vector<DevPrim> dev_prims;
	std::vector<DevShade> dev_shade;
	std::vector<uint32_t> prim_rank;
	std::vector<uint32_t> big_leaves; // pairs (first slot, count)
	std::vector<uint32_t> blob;       // scene blob for LDS staging (empty when the scene is not tiny)
	uint32_t blob_off[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // nodes, prims, shade, rank, materials, textures, lights, big_leaves
	uint32_t root_ref = 0;
	std::vector<uint32_t> dev_lights;
	std::vector<DevMaterial> materials;
	std::vector<DevTexture> textures;            // image / perlin pointers are patched after upload
	std::vector<std::vector<float>> tex_images;  // per texture: decoded pixels (may be empty)
	std::vector<std::vector<float>> tex_perlin_vecs;
	std::vector<std::vector<uint32_t>> tex_perlin_perm;
	std::vector<float> sky_cdf;                  // rows (res_y * (res_x+1)) then marginal (res_y+1)
	std::vector<uint8_t> sky_guide;              // (res_y+1) rows of sky_guide_k upper-bound indices (rt_types.h DevSky)
	uint32_t sky_guide_k = 0;
	rt_sky_desc sky;
	float root_min[3], root_max[3];
	uint32_t stack_depth = 2;        // traversal stack entries per lane: enough for the two-child AND the wide walk
	uint32_t stack_depth_narrow = 2; // ... for the two-child walk alone (all an exhaustive or narrow-only launch needs)
	bool has_triangles = false;
};

// returns RT_OK or an rt_status; `err` receives the message
int build_host_scene(const rt_scene_desc *desc, HostScene &out, std::string &err);

// Division by a constant the kernels know before they start (rt_lean.h div_by_verified): for f32 c with |c| in [2^-20, 2^32] returns
// true and rc = RN(1 / c) iff  fma(fma(-c, x * rc, x), rc, x * rc) == x / c  for EVERY significand x (2^23 of them, enumerated
// here; the sequence is scale-invariant while nothing under- or overflows).  Cached per process.
bool verified_reciprocal(float c, float *rc);

// SimpleCamera::new  camera.rs:20-54
void camera_new(rt_camera *out, const float origin[3], const float lookat[3], const float vup[3], float fov,
                float aspect_ratio, float aperture, float focus_dist);

} // namespace rt
