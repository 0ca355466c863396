// rt_types.h -- how a scene lies in HBM for the gfx950 kernels.
//
// Host reference types (Node 64 B with Option<[usize;2]>, AllPrimitives ~88 B enum behind a
// reference, Arc<MeshData> vertex indirection: SURVEY section 8) are re-laid-out for 64-lane
// wavefronts:
//   * DevNode (64 B, one 64-B-aligned record = four dwordx4 loads): an inner node carries the
//     bounds of BOTH children, so one fetch decides both AABB tests of acceleration/mod.rs:
//     208-210 and the child records are touched only when their box is hit.
//   * DevPrim (48 B, three dwordx4): vertex positions are gathered out of MeshData at upload
//     (no index chase during traversal); the type tag and material id ride in the spare .w lane.
//   * DevShade (48 B): vertex normals, read once per ACCEPTED closest hit, never during traversal.
//   * sky CDF rows (res_y x (res_x+1) floats) + marginal CDF: small enough for LDS (40.8 KB at the
//     loader's default 100x100); the pdf table is not stored -- pdf[i] == cdf[i+1]-cdf[i] bit for
//     bit, because that is how Distribution1D::new computes it (statistics/distributions.rs:32-38).
#pragma once

#include <stdint.h>
#include "rt_vec.h"

namespace rt {

constexpr uint32_t kNoPrim = 0xFFFFFFFFu; // usize::MAX on the device side

// Compile-time description of what a scene can contain.  The kernels are instantiated for a few
// feature sets and rt_api.cpp launches the smallest one that covers the scene: code for absent
// primitive / material / texture types is not even compiled in (smaller kernel, fewer registers).
// The arithmetic of what IS present is untouched, so every variant returns the same pixels.
template <bool TRI, bool LIGHTS, bool CMAT, bool CTEX> struct Feat {
	static constexpr bool tri = TRI;       // Triangle / MeshTriangle primitives
	static constexpr bool lights = LIGHTS; // emissive primitives (Bvh.lights non-empty)
	static constexpr bool cmat = CMAT;     // Reflect / Refract / TrowbridgeReitz materials
	static constexpr bool ctex = CTEX;     // Checkered / Image / Perlin textures
	static constexpr bool pair = false;    // (see FeatPair)
	static constexpr bool known_materials = false; // (see FeatPair)
};
using FeatFull = Feat<true, true, true, true>;
// The smallest set plus one fact about the TREE: one inner node over two leaves of one primitive each (scenes/rtweekend1.ssml:
// the ground and one sphere).  The walk is then two box tests and two sphere tests on records that arrive through one
// round of scalar loads (rt_intersect.h), and with the general walk not even compiled in the coarse kernels need no
// spilled register (profiles/resource_table.json).  Chosen by the host per launch (rt_api.cpp), like every other set.
// ... and one fact about its MATERIALS: both primitives carry a Lambertian over a SolidColour and the sky's material is an
// Emit (what the loader makes of `sky ( texture ... )`).  A material's type then follows from what a ray hit, and the
// dependent loads that only looked it up are not compiled in (rt_shade.h, kMatRead).  A two-leaf scene with other materials
// runs the spheres-only set, which walks such a tree the same way.
struct FeatPair : Feat<false, false, false, false> {
	static constexpr bool pair = true;
	static constexpr bool known_materials = true;
};

// child reference: bit 31 clear = inner node index; bit 31 set = leaf, bits 26-30 primitive count
// (1..31; 0 = entry of DevScene::big_leaves), bits 0-25 first primitive slot (or big-leaf index)
constexpr uint32_t kLeafFlag = 0x80000000u;
constexpr uint32_t kLeafInlineMax = 31u;
constexpr uint32_t kLeafSlotMask = 0x03FFFFFFu;
struct alignas(64) DevNode {
	float c0min[3], c0max[3];
	float c1min[3], c1max[3];
	uint32_t c0, c1;
	uint32_t pad0, pad1;
};
static_assert(sizeof(DevNode) == 64, "DevNode must be one 64-byte record");

// Wide node: up to four children of a collapsed subtree of the reference tree in ONE 64-byte record.  What bounds a
// walk over a big tree on this chip is the rate at which a CU's texture addresser / L1 serves DIVERGENT 16-byte lane
// loads (about 0.75 per cycle per CU; profiles/archive_r01_r02/r02c_*): the cost of a walk is the number of 16-byte pieces its lanes
// fetch, so the node carries as much tree per piece as it can: child boxes are 8-bit offsets on a per-node grid
// (origin + q * 2^exp per axis), rounded OUTWARD, i.e. conservative supersets of the reference boxes.  Conservative boxes
// only decide where the walk goes; whether a leaf's primitives are tested is decided by the leaf's EXACT reference
// box (DevScene::leaf_box) and the reference's own predicate.  Why that returns the reference's hits: rt_intersect.h.
// Absent children have child == kRefNone.
// COMPACT FORM (what the kernels fetch; rt_build.cpp encodes it from the explicit form above and decodes every node back as a
// check): the inner children of a node are consecutive nodes and its leaf children consecutive LEAF INDICES, so the four
// references fold into   child[0] = first inner child | leaf mask << 26   child[1] = first leaf index | present mask << 26
// and a 2-bit offset per child in the top byte of `exps` (child k is inner node child[0] + offset_k or leaf child[1] +
// offset_k by bit k of the mask).  A node step then needs only the first THREE 16-byte pieces of the record -- the walk
// is bound by the number of such pieces its lanes fetch -- and a leaf's (first slot, count) reference
// rides in the spare word of its exact box (DevLeafBox::ref), which the walk fetches anyway before it touches a primitive.
// An absent child has its bit of the present mask clear (and an inverted interval on every axis: but the walk's padded box
// test lets an inverted interval through once the node's extent is below ~2e-5 of its distance, so the bit is what counts).
constexpr uint32_t kRefNone = 0x7FFFFFFEu;
struct alignas(64) DevNodeQ4 {
	float origin[3];     // lower corner of the node's grid
	uint32_t exps;       // byte k (k = 0..2): biased exponent of axis k's grid step, i.e. step = bits(exp << 23)
	uint32_t qlo[3];     // qlo[axis]: byte c = child c's lower bound on the grid
	uint32_t qhi[3];     //            ... upper bound
	uint32_t child[4];
	uint32_t pad[2];
};
static_assert(sizeof(DevNodeQ4) == 64, "DevNodeQ4 must be one 64-byte record");

// exact reference box of a leaf (32-byte stride: two dwordx4), indexed by the leaf's index in the wide tree; pad0 carries the
// leaf's reference (kLeafFlag | count << 26 | first slot, or a big-leaf index) as a bit pattern
struct alignas(32) DevLeafBox {
	float lo[3], pad0;
	float hi[3], pad1;
};

enum : uint32_t { kPrimSphere = 0, kPrimTriangle = 1, kPrimMeshTriangle = 2 };

// Material HANDLE: what a primitive record and DevSky carry and what a path holds on to between bounces --
//   index << 6 | texture type << 3 | material type
// The two type tags decide which arm of the shading code runs; read from the record they were the first two links of a
// chain of dependent loads (type -> texture type -> colours).  Riding in the handle they are known the moment the primitive
// record (which the hit needs anyway) has arrived, and the one load of the record that remains goes out at once.
constexpr uint32_t kMatHandleShift = 6u;
constexpr uint32_t kMatMaxCount = 1u << 24; // meta = prim type (2 bits) | handle << 2
inline __host__ __device__ uint32_t mat_handle_make(uint32_t index, int32_t type, int32_t tex_type)
{
	return (index << kMatHandleShift) | (((uint32_t)tex_type & 7u) << 3) | ((uint32_t)type & 7u);
}
inline __host__ __device__ uint32_t mat_handle_index(uint32_t h) { return h >> kMatHandleShift; }
inline __host__ __device__ int32_t mat_handle_type(uint32_t h) { return (int32_t)(h & 7u); }
inline __host__ __device__ int32_t mat_handle_tex_type(uint32_t h) { return (int32_t)((h >> 3) & 7u); }

// sphere:   a = (centre.xyz, meta)  b = (radius, 0, 0, 0)   c unused
// triangle: a = (p0.xyz, meta)      b = (p1.xyz, 0)         c = (p2.xyz, 0)
// meta = type | material handle << 2 (bit pattern stored in the float lane)
struct alignas(16) DevPrim {
	float a[4], b[4], c[4];
};
static_assert(sizeof(DevPrim) == 48, "DevPrim must be three 16-byte lanes");

struct alignas(16) DevShade {
	float n0[4], n1[4], n2[4]; // vertex normals (triangles); .w unused
};

// 64 B.  The material's texture is repeated inside the record when it is a SolidColour or a Lerp (type + two
// colours are all there is to it): shading then reads ONE record instead of chasing material -> texture,
// one dependent load less on every evaluation.
struct alignas(16) DevMaterial {
	int32_t type;
	uint32_t texture;
	float param;
	int32_t tex_type;   // copy of textures[texture].type
	float tex_c1[3];    // ... .c1
	float tex_c2[3];    // ... .c2
	float ior[3];
	float metallic;
	uint32_t pad[2];
};
static_assert(sizeof(DevMaterial) == 64, "DevMaterial must be 64 bytes");

struct DevTexture {
	int32_t type;
	float c1[3];
	float c2[3];
	uint32_t dim_x, dim_y;      // ImageTexture.dim = (w-1, h-1)
	uint32_t pad;
	const float *image;         // (dim_x+1)*(dim_y+1)*3
	const float *perlin_vecs;   // 256*3
	const uint32_t *perlin_perm; // 3*256
};

struct DevSky {
	uint32_t texture, material; // (material: a handle, see kMatHandleShift)
	uint32_t res_x, res_y;     // (0,0): not samplable
	const float *row_cdf;      // res_y * (res_x+1)
	const float *marginal_cdf; // res_y + 1
	// Guide tables for the CDF inversions (0 = none): guide[r * guide_k + k] is the upper-bound index of
	// k / guide_k in row r (row res_y = the marginal).  A search starts there and steps right -- the
	// same index Distribution1D::sample's binary search returns (distributions.rs:51-72), in about
	// three dependent reads instead of seven.  guide_k is a power of two, so (uint)(num * guide_k) is exact.
	const uint8_t *guide;      // (res_y + 1) * guide_k
	uint32_t guide_k;
	// x / res_x and x / res_y as two fma steps on the host's correctly rounded reciprocals -- set only when the host has verified,
	// by enumeration, that this returns the bits of the division for every x (rt_build.h verified_reciprocal); 0: plain division
	uint32_t inv_res_ok;
	float inv_res_x, inv_res_y;
};

struct DevScene {
	const DevNode *nodes;
	const DevPrim *prims;
	const DevShade *shade;
	const uint32_t *prim_rank; // position of each primitive slot in reference BFS-leaf order
	const DevMaterial *materials;
	const DevTexture *textures;
	const uint32_t *lights;    // Bvh.lights
	const uint2 *big_leaves;   // (first slot, count) of leaves with more than 31 primitives
	uint32_t n_nodes, n_prims, n_lights, n_materials, n_textures;
	uint32_t root_ref;         // child-style reference to the root (a leaf ref when the tree is one leaf)
	float root_min[3], root_max[3];
	// "scene blob": every array above packed into one allocation (16-byte aligned sections) so a
	// workgroup can copy a tiny scene into LDS with one loop; offsets are in bytes, 0 bytes = no blob
	const uint32_t *blob;
	uint32_t blob_bytes;
	uint32_t off_nodes, off_prims, off_shade, off_rank, off_materials, off_textures, off_lights, off_big_leaves;
	uint32_t stack_depth;      // traversal stack entries per lane (enough for the two-child AND the wide walk)
	const DevNodeQ4 *nodes4;   // wide tree (null: none was built, e.g. non-finite bounds)
	const DevLeafBox *leaf_box; // exact leaf boxes (+ leaf references) for the wide walk, by leaf index
	uint32_t root4_ref;        // child-style reference to its root
	uint32_t n_nodes4;
	uint32_t narrow_only;      // RT_TUNE_WALK = 1: every ray takes the two-child walk (tests, A/B measurements)
	uint32_t has_triangles;
	// every Lambertian record's (colour x albedo) components are zero or >= 2^-30 (host check): with a cosine that is zero or
	// >= 2^-30 the numerators of Lambertian::eval are then zero or >= 2^-60, where the verified division by pi applies (rt_shade.h)
	uint32_t lambert_tame;
	// Bvh.lights has exactly one entry: its primitive slot (else kNoPrim).  Every lane that samples "a light" then samples THE
	// light, whose record is wave-uniform: it arrives through scalar loads (rt_render.hip do_light / do_scatter)
	uint32_t single_light;
	DevSky sky;
};

// FeatPair's scene as KERNEL ARGUMENTS (rt_render.hip RenderArgs::pair): one node, two spheres, two Lambertian records and
// the sky's Emit are 40 dwords that are the same for every lane of every wave.  A super-phase of the coarse schedule reads
// them from the kernarg segment in ONE round of scalar loads at its top (load_pair_scene) and its walks, make_hit and
// material evaluations take them from SGPRs -- instead of chasing them through the scene's arrays (node -> slots ->
// primitive records -> material records: two dependent rounds of scalar loads per walk, a per-lane load per hit record and
// per material evaluation).  Everything is in CHILD order (entry 0 belongs to the node's first child, the first candidate
// of Bvh::get_intersection_candidates): a per-lane slot selects between the two entries by `slot == slot0`.  Filled by the
// host from the arrays the other kernels read (rt_api.cpp), so both routes see the same bits.  Measured same-box at 256
// spp: 24.0 -> 23.55 ms; the same block read through the kernarg pointer at each use (seven rounds per iteration instead of
// two) measured 24.2 (profiles/r03_ab_logs/r05b_small_ab.log, r05c_small_ab.log).
struct alignas(64) DevPairScene {
	float c0min[3], c0max[3], c1min[3], c1max[3]; // DevNode's child boxes
	uint32_t slot0, slot1;                        // primitive slot of each leaf
	uint32_t rank0, rank1;                        // DevScene::prim_rank of those slots (exact-t ties)
	float sphere[2][4];                           // centre.xyz, radius
	float lambert[2][4];                          // SolidColour rgb, albedo (Lambertian::param)
	float sky_param;                              // the sky's Emit: strength,
	int32_t sky_tex_type;                         // its texture (SolidColour or Lerp)
	float sky_c1[3], sky_c2[3];
	float inv_radius[2];                          // RN(1 / radius): host-verified (rt_build.h verified_reciprocal) -- part of the contract
	uint32_t pad[6];
};
static_assert(sizeof(DevPairScene) == 192, "DevPairScene: three 64-byte lines");
// (what the kernels of every other feature set pass where a pair scene is expected: never read)
__device__ static const DevPairScene kNoPairScene = {};

struct DevCamera {
	float origin[3], lower_left[3], horizontal[3], vertical[3];
};

struct DevRenderParams {
	uint32_t width, height;
	uint32_t spp;
	uint32_t sample_begin_lo, sample_begin_hi;
	uint32_t seed_lo, seed_hi;
	uint32_t max_depth, rr_threshold;
	uint32_t shard_index, shard_count;
	uint32_t tile_w, tile_h, tiles_x, tiles_y;
	uint32_t n_work;          // pixels this launch owns (incl. out-of-image padding of edge tiles)
	uint32_t sample_split;    // S: chunks per pixel (1 = the reference's strictly sequential fold)
	uint32_t n_items;         // n_work * S work items, chunk-major: item w = chunk (w / n_work) of pixel (w % n_work)
	int32_t shard_layout;     // 1: packed shard output
	int32_t prune;            // t-pruned traversal (validated equal to the reference's exhaustive one)
	uint32_t sky_in_lds;      // sky CDF tables are staged in LDS
	uint32_t scene_in_lds;    // the scene blob is staged in LDS (tiny scenes)
	uint32_t stack_cap;       // traversal stack entries per lane kept in LDS (rt_intersect.h StackMem)
	uint32_t stack_ovf_depth; // ... and in the global overflow area (0: the LDS part covers the worst case)
	uint32_t xchg_slots;      // RT_TUNE_EXCHANGE: parked paths / pixels the workgroup's pool holds (rt_render.hip, XCHG)
	// Tiles of exactly 64 pixels with a power-of-two width (the default 8 x 8) on an image below 65 536 x 65 536 and a
	// sample_split that is a power of two <= 64: log2 of the tile width | log2 of the split << 8.  A wave's claim of 64 work
	// items then lies inside ONE tile -- 64 / S of its pixels times their S chunks -- whose origin is worked out once per claim
	// (rt_render.hip, acquire_tiles).  0xFFFFFFFF: any other tiling or split, every item is decoded on its own (chunk-major).
	uint32_t tile_log2_w;
	// (jitter + x) / (W - 1) and (jitter + y) / (H - 1) of the pixel loop (random_sampler.rs:55-59) by verified reciprocals
	// (rt_build.h verified_reciprocal); w1h1_ok = 0: plain division
	uint32_t w1h1_ok;
	float inv_w1, inv_h1;
};

} // namespace rt
