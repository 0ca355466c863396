// rt_build.cpp -- host-side scene build of the HIP back end.
//
// Part 1 builds the BVH exactly as the reference does, because the resulting primitive order and
// tree shape decide which of two equal-t hits wins and which primitives are lights:
//   Bvh::new / build_bvh / PrimitiveInfo   acceleration/mod.rs:22-160
//   SplitType::{Sah,Middle,EqualCounts}, partition!, calculate_b, split_equal   acceleration/split.rs:5-210
//   AABB::{merge, extend_contains, surface_area}   acceleration/aabb.rs:59-86
//   Sky::new -> generate_values + Distribution2D::new   sky.rs:22-39, textures/mod.rs:32-50,
//                                                       statistics/distributions.rs:12-44,83-99
// Part 2 re-lays the result out for the GPU (rt_types.h): two-child 64-byte nodes, gathered 48-byte
// primitive records, BFS-leaf rank per primitive, CDF-only sky tables.
#include "rt_build.h"

#include <algorithm>
#include <cmath>
#include <mutex>
#include <unordered_map>
#include <chrono>
#include <cstring>
#include <deque>
#include <functional>
#include <future>

namespace rt {
namespace {

struct PrimInfo { // PrimitiveInfo  mod.rs:22-27
	uint64_t index;
	V3 min, max, center;
};

struct Bounds { // Option<AABB>
	bool some = false;
	V3 min{0, 0, 0}, max{0, 0, 0};
	void merge(V3 mn, V3 mx) // AABB::merge  aabb.rs:59-67
	{
		if (some) {
			min = min_by_component(min, mn);
			max = max_by_component(max, mx);
		} else {
			min = mn;
			max = mx;
			some = true;
		}
	}
	float surface_area() const // aabb.rs:83-86
	{
		const V3 e = max - min;
		return 2.0f * (e.x * e.y + e.x * e.z + e.y * e.z);
	}
};

inline float axis_of(int axis, V3 p) { return axis == 0 ? p.x : (axis == 1 ? p.y : p.z); }

constexpr int kNumBuckets = 12;    // split.rs:5
constexpr uint64_t kMaxInNode = 255; // split.rs:6

// `(expr) as usize` saturates and maps NaN to 0
inline uint64_t as_usize(float f)
{
	if (!(f > 0.0f))
		return 0;
	if (f >= 1.8446744e19f)
		return UINT64_MAX;
	return (uint64_t)f;
}

// independent per-element host loops over millions of primitives: contiguous chunks on a few threads
template <class Fn> void parallel_for(uint64_t n, Fn fn)
{
	constexpr uint64_t kMinChunk = 65536;
	const uint64_t chunks = std::min<uint64_t>(16, (n + kMinChunk - 1) / kMinChunk);
	if (chunks <= 1) {
		fn(0, n);
		return;
	}
	std::vector<std::future<void>> pending;
	for (uint64_t c = 1; c < chunks; ++c)
		pending.push_back(std::async(std::launch::async, [=, &fn] { fn(n * c / chunks, n * (c + 1) / chunks); }));
	fn(0, n / chunks);
	for (auto &f : pending)
		f.get();
}

class Builder {
  public:
	Builder(int split_type, std::vector<HostNode> &nodes) : split_type_(split_type), nodes_(nodes) {}

	// One node of Bvh::build_bvh (mod.rs:97-160): push the node, decide leaf or split.  Returns the node's
	// index; `mid` = 0 means leaf, otherwise the children are info[0, mid) and info[mid, n).
	uint64_t node_and_split(PrimInfo *info, uint64_t n, uint64_t offset, uint64_t &mid)
	{
		mid = 0;
		Bounds bounds;
		for (uint64_t i = 0; i < n; ++i)
			bounds.merge(info[i].min, info[i].max);
		const uint64_t node_index = nodes_.size();
		HostNode hn;
		hn.min[0] = bounds.min.x; hn.min[1] = bounds.min.y; hn.min[2] = bounds.min.z;
		hn.max[0] = bounds.max.x; hn.max[1] = bounds.max.y; hn.max[2] = bounds.max.z;
		hn.child[0] = hn.child[1] = -1;
		hn.primitive_offset = offset;
		hn.number_primitives = n;
		nodes_.push_back(hn);
		if (n == 1)
			return node_index;

		Bounds cb;
		for (uint64_t i = 0; i < n; ++i)
			cb.merge(info[i].center, info[i].center); // extend_contains
		const V3 extent = cb.max - cb.min;
		const int axis = (extent.x > extent.y && extent.x > extent.z) ? 0 : (extent.y > extent.z ? 1 : 2); // Axis::get_max_axis
		if (std::fabs(axis_of(axis, cb.min) - axis_of(axis, cb.max)) < 100.0f * kF32Epsilon)
			return node_index; // all centroids coincide on the widest axis: one leaf

		mid = split(bounds, cb, axis, info, n);
		return node_index;
	}

	// Bvh::build_bvh, sequential (node pushed before its children: preorder numbering)
	uint64_t build(PrimInfo *info, uint64_t n, uint64_t offset)
	{
		uint64_t mid;
		const uint64_t node_index = node_and_split(info, n, offset, mid);
		if (mid == 0)
			return node_index;
		const uint64_t c0 = build(info, mid, offset);
		const uint64_t c1 = build(info + mid, n - mid, offset + mid);
		nodes_[node_index].child[0] = (int64_t)c0;
		nodes_[node_index].child[1] = (int64_t)c1;
		return node_index;
	}

	// The same tree with the two subtrees of every large node built concurrently.  A subtree touches only its own
	// slice of `info` and is built into its own vector with local indices; splicing [node][left][right] and
	// shifting the child indices restores the preorder numbering, so the result is the sequential one bit for bit.
	static void build_concurrent(int split_type, PrimInfo *info, uint64_t n, uint64_t offset, std::vector<HostNode> &out, int depth)
	{
		Builder b(split_type, out);
		static const int max_depth = std::getenv("RT_HIP_BUILD_DEPTH") ? std::atoi(std::getenv("RT_HIP_BUILD_DEPTH")) : kConcurrentDepth;
		if (n < kConcurrentMin || depth >= max_depth) {
			b.build(info, n, offset);
			return;
		}
		uint64_t mid;
		b.node_and_split(info, n, offset, mid);
		if (mid == 0)
			return;
		std::vector<HostNode> left, right;
		left.reserve(2 * mid);
		right.reserve(2 * (n - mid));
		std::future<void> other = std::async(std::launch::async, [&] { build_concurrent(split_type, info, mid, offset, left, depth + 1); });
		build_concurrent(split_type, info + mid, n - mid, offset + mid, right, depth + 1);
		other.get();
		const int64_t base_left = (int64_t)out.size(), base_right = base_left + (int64_t)left.size();
		out[(size_t)base_left - 1].child[0] = base_left;
		out[(size_t)base_left - 1].child[1] = base_right;
		out.reserve(out.size() + left.size() + right.size());
		for (int side = 0; side < 2; ++side) {
			const std::vector<HostNode> &sub = side == 0 ? left : right;
			const int64_t base = side == 0 ? base_left : base_right;
			for (HostNode nd : sub) {
				if (nd.child[0] >= 0) {
					nd.child[0] += base;
					nd.child[1] += base;
				}
				out.push_back(nd);
			}
		}
	}
	static constexpr uint64_t kConcurrentMin = 32768; // smaller subtrees are not worth a thread
	static constexpr int kConcurrentDepth = 5;        // up to 32 subtrees in flight

  private:
	// the partition! macro  split.rs:8-32
	template <class Pred> static uint64_t partition(PrimInfo *a, uint64_t len, Pred pred)
	{
		uint64_t left = 0, right = len - 1;
		for (;;) {
			while (left < len && pred(a[left]))
				++left;
			while (right > 0 && !pred(a[right]))
				--right;
			if (left >= right)
				return left;
			std::swap(a[left], a[right]);
		}
	}
	static void sort_by_axis(PrimInfo *a, uint64_t len, int axis) // slice::sort_by is stable
	{
		std::stable_sort(a, a + len, [axis](const PrimInfo &l, const PrimInfo &r) { return axis_of(axis, l.center) < axis_of(axis, r.center); });
	}
	static uint64_t bucket_of(int axis, const PrimInfo &p, float min, float extent) // calculate_b  split.rs:189-199
	{
		uint64_t b = as_usize((float)kNumBuckets * (axis_of(axis, p.center) - min) / extent);
		if (b == (uint64_t)kNumBuckets)
			b -= 1;
		return b;
	}

	// SplitType::split  split.rs:78-187
	uint64_t split(const Bounds &bounds, const Bounds &cb, int axis, PrimInfo *info, uint64_t len)
	{
		if (split_type_ == RT_SPLIT_MIDDLE) {
			const float point_mid = 0.5f * (axis_of(axis, cb.min) + axis_of(axis, cb.max));
			const uint64_t mid = partition(info, len, [&](const PrimInfo &p) { return axis_of(axis, p.center) < point_mid; });
			if (mid == 0 || mid == len - 1)
				sort_by_axis(info, len, axis);
			return mid;
		}
		if (split_type_ == RT_SPLIT_EQUAL_COUNTS || len <= 4) { // split_equal  split.rs:201-210
			sort_by_axis(info, len, axis);
			return len / 2;
		}
		// SAH, 12 buckets
		uint32_t count[kNumBuckets] = {};
		Bounds bb[kNumBuckets];
		const float max_val = axis_of(axis, cb.max), min_val = axis_of(axis, cb.min);
		const float centroid_extent = max_val - min_val;
		for (uint64_t i = 0; i < len; ++i) {
			const uint64_t b = bucket_of(axis, info[i], min_val, centroid_extent);
			count[b] += 1;
			bb[b].merge(info[i].min, info[i].max);
		}
		float costs[kNumBuckets - 1];
		for (int i = 0; i < kNumBuckets - 1; ++i) {
			Bounds left, right;
			uint32_t count_left = 0, count_right = 0;
			for (int j = 0; j <= i; ++j)
				if (bb[j].some) {
					left.merge(bb[j].min, bb[j].max);
					count_left += count[j];
				}
			for (int j = i + 1; j < kNumBuckets; ++j)
				if (bb[j].some) {
					right.merge(bb[j].min, bb[j].max);
					count_right += count[j];
				}
			const float left_sa = left.some ? left.surface_area() : 0.0f;
			const float right_sa = right.some ? right.surface_area() : 0.0f;
			costs[i] = 0.125f + ((float)count_left * left_sa + (float)count_right * right_sa) / bounds.surface_area();
		}
		float min_cost = costs[0];
		uint64_t min_cost_index = 0;
		for (int i = 1; i < kNumBuckets - 1; ++i)
			if (costs[i] < min_cost) {
				min_cost = costs[i];
				min_cost_index = (uint64_t)i;
			}
		if (len > kMaxInNode || min_cost < (float)len)
			return partition(info, len, [&](const PrimInfo &p) { return bucket_of(axis, p, min_val, centroid_extent) <= min_cost_index; });
		return 0;
	}

	int split_type_;
	std::vector<HostNode> &nodes_;
};

// ---- host copy of Texture::colour_value, used only by generate_values (textures/mod.rs:32-50) ----
V3 host_texture_colour(const HostScene &hs, uint32_t tex, V3 direction, V3 point)
{
	const DevTexture &t = hs.textures[tex];
	const V3 c1 = v3(t.c1[0], t.c1[1], t.c1[2]), c2 = v3(t.c2[0], t.c2[1], t.c2[2]);
	switch (t.type) {
	case RT_TEX_SOLID:
		return c1;
	case RT_TEX_LERP: {
		const float tt = direction.z * 0.5f + 0.5f;
		return c1 * tt + c2 * (1.0f - tt);
	}
	case RT_TEX_CHECKERED: {
		const float sign = rt_sinf(10.0f * point.x) * rt_sinf(10.0f * point.y) * rt_sinf(10.0f * point.z);
		return sign > 0.0f ? c1 : c2;
	}
	case RT_TEX_IMAGE: {
		const float phi = rt_atan2f(direction.y, direction.x) + kPi;
		const float theta = rt_acosf(direction.z);
		const uint64_t x_pixel = as_usize((float)t.dim_x * (phi / (2.0f * kPi)));
		const uint64_t y_pixel = as_usize((float)t.dim_y * (theta / kPi));
		uint64_t index = y_pixel * (t.dim_x + 1u) + x_pixel;
		const uint64_t n = (uint64_t)(t.dim_x + 1u) * (t.dim_y + 1u);
		if (index >= n)
			index = n - 1;
		const float *px = hs.tex_images[tex].data() + 3 * index;
		return v3(px[0], px[1], px[2]);
	}
	case RT_TEX_PERLIN: { // Perlin::noise at `point` (textures/mod.rs:110-169)
		const std::vector<float> &rv = hs.tex_perlin_vecs[tex];
		const std::vector<uint32_t> &pm = hs.tex_perlin_perm[tex];
		const float u = point.x - std::floor(point.x), v = point.y - std::floor(point.y), w = point.z - std::floor(point.z);
		auto as_i32 = [](float f) -> int32_t {
			if (f != f) return 0;
			if (f >= 2147483648.0f) return INT32_MAX;
			if (f <= -2147483648.0f) return INT32_MIN;
			return (int32_t)f;
		};
		const int32_t i = as_i32(std::floor(point.x)), j = as_i32(std::floor(point.y)), k = as_i32(std::floor(point.z));
		const float uu = u * u * (3.0f - 2.0f * u), vv = v * v * (3.0f - 2.0f * v), ww = w * w * (3.0f - 2.0f * w);
		float value = 0.0f;
		for (int index = 0; index < 8; ++index) {
			const int ii = index / 4, jj = (index / 2) % 2, kk = index % 2;
			const uint32_t a = pm[(uint32_t)(i + ii) & 255u] ^ pm[256 + ((uint32_t)(j + jj) & 255u)] ^ pm[512 + ((uint32_t)(k + kk) & 255u)];
			const V3 c = v3(rv[3 * (a & 255u)], rv[3 * (a & 255u) + 1], rv[3 * (a & 255u) + 2]);
			const float fi = (float)ii, fj = (float)jj, fk = (float)kk;
			value += (fi * uu + (1.0f - fi) * (1.0f - uu)) * (fj * vv + (1.0f - fj) * (1.0f - vv)) *
			         (fk * ww + (1.0f - fk) * (1.0f - ww)) * dot(c, v3(u - fi, v - fj, w - fk));
		}
		return (0.5f * v3s(1.0f)) * (1.0f + value);
	}
	default:
		return v3s(1.0f);
	}
}

// Distribution1D::new  distributions.rs:12-44 -- only the cdf is kept (pdf[i] = cdf[i+1]-cdf[i])
void make_cdf(const float *values, uint64_t n, float *cdf)
{
	cdf[0] = 0.0f;
	for (uint64_t i = 1; i <= n; ++i)
		cdf[i] = cdf[i - 1] + values[i - 1];
	const float c = cdf[n];
	if (c != 0.0f)
		for (uint64_t i = 0; i <= n; ++i)
			cdf[i] /= c;
}

} // namespace

void camera_new(rt_camera *out, const float origin_[3], const float lookat_[3], const float vup_[3], float fov,
                float aspect_ratio, float aperture, float focus_dist)
{
	(void)aperture; // stored as lens_radius and never read (camera.rs:51,57-63)
	const V3 origin = v3(origin_[0], origin_[1], origin_[2]);
	const V3 lookat = v3(lookat_[0], lookat_[1], lookat_[2]);
	const V3 vup = v3(vup_[0], vup_[1], vup_[2]);
	const float viewport_width = 2.0f * rt_tanf(rt_to_radians(fov) / 2.0f);
	const float viewport_height = viewport_width / aspect_ratio;
	const V3 w = normalised(origin - lookat);
	const V3 u = normalised(cross(w, vup));
	const V3 v = cross(u, w);
	const V3 horizontal = focus_dist * u * viewport_width;
	const V3 vertical = focus_dist * v * viewport_height;
	const V3 lower_left = origin - horizontal / 2.0f - vertical / 2.0f - focus_dist * w;
	const V3 all[4] = {origin, lower_left, horizontal, vertical};
	float *dst[4] = {out->origin, out->lower_left, out->horizontal, out->vertical};
	for (int i = 0; i < 4; ++i) {
		dst[i][0] = all[i].x;
		dst[i][1] = all[i].y;
		dst[i][2] = all[i].z;
	}
}

// ---- division by a launch constant, verified (rt_build.h, rt_lean.h) ----
// x / c costs eleven instructions on gfx950 (two v_div_scale, v_rcp, a Newton step, the quotient with two corrections,
// v_div_fmas, v_div_fixup).  For a divisor known before the launch the reciprocal can be formed here, correctly rounded, and
// one correction step is enough IF the rounding works out for that divisor -- which is decided by trying every significand:
// the three operations are IEEE multiply / fma on both sides, so what holds here holds on the device, bit for bit.
#if defined(__x86_64__)
__attribute__((target("fma")))
#endif
static bool three_op_quotient_is_exact(float c, float rc)
{
	for (uint32_t m = 0; m < (1u << 23); ++m) {
		float x;
		const uint32_t u = (127u << 23) | m; // [1, 2): every other binade is this one times a power of two
		std::memcpy(&x, &u, 4);
		const float q0 = x * rc;
		const float e = __builtin_fmaf(-c, q0, x);
		const float q = __builtin_fmaf(e, rc, q0);
		const float ref = x / c;
		if (std::memcmp(&q, &ref, 4) != 0)
			return false;
	}
	return true;
}
bool verified_reciprocal(float c, float *rc)
{
	static std::mutex mu;
	static std::unordered_map<uint32_t, std::pair<bool, float>> cache;
	uint32_t key;
	std::memcpy(&key, &c, 4);
	{
		std::lock_guard<std::mutex> lock(mu);
		auto it = cache.find(key);
		if (it != cache.end()) {
			*rc = it->second.second;
			return it->second.first;
		}
	}
	const float a = std::fabs(c);
	bool ok = a >= 0x1p-20f && a <= 0x1p32f; // false for NaN
	const float r = ok ? 1.0f / c : 0.0f;     // correctly rounded (IEEE division on the host)
#if defined(__x86_64__)
	ok = ok && __builtin_cpu_supports("fma"); // (without hardware fma the enumeration would take seconds: plain division then)
#endif
	ok = ok && three_op_quotient_is_exact(c, r);
	std::lock_guard<std::mutex> lock(mu);
	cache[key] = {ok, r};
	*rc = r;
	return ok;
}

int build_host_scene(const rt_scene_desc *d, HostScene &hs, std::string &err)
{
	if (!d) {
		err = "null scene descriptor";
		return RT_ERR_INVALID_ARGUMENT;
	}
	if (d->abi_version != RT_ABI_VERSION) {
		err = "abi version mismatch";
		return RT_ERR_INVALID_ARGUMENT;
	}
	if (d->n_primitives == 0) {
		err = "scene has no primitives (Bvh::new would panic on bounds.unwrap())";
		return RT_ERR_INVALID_ARGUMENT;
	}
	if (d->n_primitives >= 0x7FFFFFFFull) {
		err = "too many primitives for 31-bit leaf references";
		return RT_ERR_UNSUPPORTED;
	}

	// RT_HIP_BUILD_TIMING=1: phase times on stderr
	const bool timing = std::getenv("RT_HIP_BUILD_TIMING") != nullptr;
	auto t_prev = std::chrono::steady_clock::now();
	auto lap = [&](const char *what) {
		if (!timing)
			return;
		const auto now = std::chrono::steady_clock::now();
		std::fprintf(stderr, "[rt_build] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
		t_prev = now;
	};
	// ---- textures ----
	hs.textures.resize(d->n_textures);
	hs.tex_images.resize(d->n_textures);
	hs.tex_perlin_vecs.resize(d->n_textures);
	hs.tex_perlin_perm.resize(d->n_textures);
	for (uint32_t i = 0; i < d->n_textures; ++i) {
		const rt_texture_desc &t = d->textures[i];
		DevTexture &o = hs.textures[i];
		std::memset(&o, 0, sizeof o);
		o.type = t.type;
		for (int k = 0; k < 3; ++k) {
			o.c1[k] = t.colour_one[k];
			o.c2[k] = t.colour_two[k];
		}
		if (t.type == RT_TEX_IMAGE) {
			if (!t.image_rgb || t.image_width == 0 || t.image_height == 0) {
				err = "image texture without pixels";
				return RT_ERR_INVALID_ARGUMENT;
			}
			const size_t n = (size_t)t.image_width * t.image_height * 3;
			hs.tex_images[i].assign(t.image_rgb, t.image_rgb + n);
			o.dim_x = t.image_width - 1; // textures/mod.rs:232
			o.dim_y = t.image_height - 1;
		} else if (t.type == RT_TEX_PERLIN) {
			if (!t.perlin_ran_vecs || !t.perlin_perm) {
				err = "perlin texture without tables";
				return RT_ERR_INVALID_ARGUMENT;
			}
			hs.tex_perlin_vecs[i].assign(t.perlin_ran_vecs, t.perlin_ran_vecs + 256 * 3);
			hs.tex_perlin_perm[i].assign(t.perlin_perm, t.perlin_perm + 3 * 256);
		} else if (t.type < 0 || t.type > RT_TEX_PERLIN) {
			err = "unknown texture type";
			return RT_ERR_INVALID_ARGUMENT;
		}
	}
	// ---- materials ----
	hs.materials.resize(d->n_materials);
	for (uint32_t i = 0; i < d->n_materials; ++i) {
		const rt_material_desc &m = d->materials[i];
		if (m.texture >= d->n_textures || m.type < 0 || m.type > RT_MAT_REFRACT) {
			err = "material descriptor out of range";
			return RT_ERR_INVALID_ARGUMENT;
		}
		DevMaterial &o = hs.materials[i];
		o.type = m.type;
		o.texture = m.texture;
		o.param = m.param;
		o.ior[0] = m.ior[0]; o.ior[1] = m.ior[1]; o.ior[2] = m.ior[2];
		o.metallic = m.metallic;
		o.pad[0] = o.pad[1] = 0;
		const DevTexture &t = hs.textures[m.texture];
		o.tex_type = t.type;
		for (int k = 0; k < 3; ++k) {
			o.tex_c1[k] = t.c1[k];
			o.tex_c2[k] = t.c2[k];
		}
	}
	if (d->n_materials >= kMatMaxCount) { // (a primitive record carries the material as a 24-bit index + type tags: rt_types.h)
		err = "more than 2^24 - 1 materials";
		return RT_ERR_UNSUPPORTED;
	}
	if (d->sky.texture >= d->n_textures || d->sky.material >= d->n_materials) {
		err = "sky texture/material index out of range";
		return RT_ERR_INVALID_ARGUMENT;
	}
	hs.sky = d->sky;

	// ---- primitives: validate and fetch geometry ----
	const uint64_t n = d->n_primitives;
	auto vertex = [&](const rt_primitive_desc &p, int k, bool normal, V3 &out) -> bool {
		if (p.type == RT_PRIM_TRIANGLE) {
			if (p.u.triangle.data >= d->n_triangles)
				return false;
			const rt_triangle_data &t = d->triangles[p.u.triangle.data];
			const float *src = normal ? t.normals : t.points;
			out = v3(src[3 * k], src[3 * k + 1], src[3 * k + 2]);
			return true;
		}
		const uint32_t mesh = p.u.mesh_triangle.mesh;
		if (mesh >= d->n_meshes)
			return false;
		const rt_mesh_desc &m = d->meshes[mesh];
		const uint32_t idx = normal ? p.u.mesh_triangle.normal_indices[k] : p.u.mesh_triangle.point_indices[k];
		if (idx >= (normal ? m.n_normals : m.n_vertices))
			return false;
		const float *src = (normal ? m.normals : m.vertices) + 3 * (size_t)idx;
		out = v3(src[0], src[1], src[2]);
		return true;
	};

	std::vector<PrimInfo> info(n);
	for (uint64_t i = 0; i < n; ++i) { // PrimitiveInfo::new  mod.rs:29-41 with get_aabb
		const rt_primitive_desc &p = d->primitives[i];
		if (p.material >= d->n_materials) {
			err = "primitive material index out of range";
			return RT_ERR_INVALID_ARGUMENT;
		}
		V3 mn, mx;
		if (p.type == RT_PRIM_SPHERE) { // sphere.rs:175-182
			const V3 c = v3(p.u.sphere.centre[0], p.u.sphere.centre[1], p.u.sphere.centre[2]);
			mn = c - p.u.sphere.radius * v3s(1.0f);
			mx = c + p.u.sphere.radius * v3s(1.0f);
			if (mn.x > mx.x || mn.y > mx.y || mn.z > mx.z) {
				err = "Maximum value in AABB must be greater than the minimum!"; // AABB::new panics
				return RT_ERR_INVALID_ARGUMENT;
			}
		} else if (p.type == RT_PRIM_TRIANGLE || p.type == RT_PRIM_MESH_TRIANGLE) { // triangle.rs:285-307
			V3 a, b, c;
			if (!vertex(p, 0, false, a) || !vertex(p, 1, false, b) || !vertex(p, 2, false, c) ||
			    !vertex(p, 0, true, mn) || !vertex(p, 1, true, mn) || !vertex(p, 2, true, mn)) {
				err = "triangle descriptor out of range";
				return RT_ERR_INVALID_ARGUMENT;
			}
			mn = min_by_component(a, min_by_component(b, c));
			mx = max_by_component(a, max_by_component(b, c));
			hs.has_triangles = true;
		} else {
			err = "unknown primitive type";
			return RT_ERR_INVALID_ARGUMENT;
		}
		info[i].index = i;
		info[i].min = mn;
		info[i].max = mx;
		info[i].center = 0.5f * (mn + mx);
	}

	lap("validate + fetch geometry");
	// ---- Bvh::new ----
	hs.nodes.clear();
	hs.nodes.reserve(2 * n);
	Builder::build_concurrent(d->split_type, info.data(), n, 0, hs.nodes, 0);
	hs.primitive_order.resize(n);
	for (uint64_t i = 0; i < n; ++i)
		hs.primitive_order[i] = info[i].index; // sort_by_indices: slot i <- primitives[info[i].index]
	hs.lights.clear();
	for (uint64_t i = 0; i < n; ++i) // mod.rs:84-88 (material_is_light -> Emit)
		if (hs.materials[d->primitives[hs.primitive_order[i]].material].type == RT_MAT_EMIT)
			hs.lights.push_back(i);
	hs.dev_lights.assign(hs.lights.begin(), hs.lights.end());

	lap("Bvh::new");
	// ---- device primitive records in slot order ----
	hs.dev_prims.resize(n);
	hs.dev_shade.resize(hs.has_triangles ? n : 1);
	parallel_for(n, [&](uint64_t slot_begin, uint64_t slot_end) {
	for (uint64_t slot = slot_begin; slot < slot_end; ++slot) {
		const rt_primitive_desc &p = d->primitives[hs.primitive_order[slot]];
		DevPrim &o = hs.dev_prims[slot];
		std::memset(&o, 0, sizeof o);
		uint32_t type = kPrimSphere;
		if (p.type == RT_PRIM_SPHERE) {
			o.a[0] = p.u.sphere.centre[0]; o.a[1] = p.u.sphere.centre[1]; o.a[2] = p.u.sphere.centre[2];
			o.b[0] = p.u.sphere.radius; // (b[1]: the radius' verified reciprocal, filled below)
		} else {
			type = p.type == RT_PRIM_TRIANGLE ? kPrimTriangle : kPrimMeshTriangle;
			V3 q[3], nn[3];
			for (int k = 0; k < 3; ++k) {
				vertex(p, k, false, q[k]);
				vertex(p, k, true, nn[k]);
			}
			o.a[0] = q[0].x; o.a[1] = q[0].y; o.a[2] = q[0].z;
			o.b[0] = q[1].x; o.b[1] = q[1].y; o.b[2] = q[1].z;
			o.c[0] = q[2].x; o.c[1] = q[2].y; o.c[2] = q[2].z;
			DevShade &s = hs.dev_shade[slot];
			std::memset(&s, 0, sizeof s);
			s.n0[0] = nn[0].x; s.n0[1] = nn[0].y; s.n0[2] = nn[0].z;
			s.n1[0] = nn[1].x; s.n1[1] = nn[1].y; s.n1[2] = nn[1].z;
			s.n2[0] = nn[2].x; s.n2[1] = nn[2].y; s.n2[2] = nn[2].z;
		}
		const DevMaterial &pm = hs.materials[p.material];
		const uint32_t meta = type | (mat_handle_make(p.material, pm.type, pm.tex_type) << 2);
		std::memcpy(&o.a[3], &meta, 4);
	}
	});

	{
		// Spheres: RN(1 / radius) next to the radius, for the hit record's (p - c) / r (rt_intersect.h make_sphere_hit_by_reciprocal)
		// -- only where the host has verified the reciprocal over every significand (15 ms per distinct radius, cached), and only for
		// the first few distinct radii of a scene: 0 = not verified, the kernels divide
		constexpr size_t kMaxVerifiedRadii = 16;
		std::vector<std::pair<uint32_t, float>> known; // (radius bits, reciprocal or 0)
		for (uint64_t slot = 0; slot < n; ++slot) {
			DevPrim &o = hs.dev_prims[slot];
			uint32_t meta;
			std::memcpy(&meta, &o.a[3], 4);
			if ((meta & 3u) != kPrimSphere)
				continue;
			uint32_t bits;
			std::memcpy(&bits, &o.b[0], 4);
			float rc = 0.0f;
			bool found = false;
			for (const auto &k : known)
				if (k.first == bits) {
					rc = k.second;
					found = true;
					break;
				}
			if (!found && known.size() < kMaxVerifiedRadii) {
				if (!verified_reciprocal(o.b[0], &rc))
					rc = 0.0f;
				known.push_back({bits, rc});
			}
			o.b[1] = rc;
		}
	}
	lap("device primitive records");
	// ---- BFS-leaf rank: the order get_intersection_candidates lists leaves in (mod.rs:199-224) ----
	hs.prim_rank.assign(n, 0);
	{
		std::deque<uint64_t> queue;
		queue.push_back(0);
		uint32_t rank = 0;
		while (!queue.empty()) {
			const HostNode &hn = hs.nodes[queue.front()];
			queue.pop_front();
			if (hn.child[0] >= 0) {
				queue.push_back((uint64_t)hn.child[0]);
				queue.push_back((uint64_t)hn.child[1]);
			} else {
				for (uint64_t s = hn.primitive_offset; s < hn.primitive_offset + hn.number_primitives; ++s)
					hs.prim_rank[s] = rank++;
			}
		}
	}

	lap("BFS-leaf rank");
	// ---- two-child device nodes; inner nodes renumbered in preorder ----
	const HostNode &root = hs.nodes[0];
	std::memcpy(hs.root_min, root.min, sizeof hs.root_min);
	std::memcpy(hs.root_max, root.max, sizeof hs.root_max);
	hs.dev_nodes.clear();
	hs.big_leaves.clear();
	if (n > kLeafSlotMask) {
		err = "more than 2^26 primitives: leaf references hold 26-bit slots";
		return RT_ERR_UNSUPPORTED;
	}
	// one reference per leaf, made once and serially (big leaves append to a table), shared by both device trees
	std::vector<uint32_t> leaf_ref_of(hs.nodes.size(), 0u);
	for (size_t i = 0; i < hs.nodes.size(); ++i) {
		const HostNode &leaf = hs.nodes[i];
		if (leaf.child[0] >= 0)
			continue;
		if (leaf.number_primitives <= kLeafInlineMax) {
			leaf_ref_of[i] = kLeafFlag | ((uint32_t)leaf.number_primitives << 26) | (uint32_t)leaf.primitive_offset;
		} else {
			hs.big_leaves.push_back((uint32_t)leaf.primitive_offset);
			hs.big_leaves.push_back((uint32_t)leaf.number_primitives);
			leaf_ref_of[i] = kLeafFlag | (uint32_t)(hs.big_leaves.size() / 2 - 1);
		}
	}
	auto leaf_ref = [&](const HostNode &leaf) -> uint32_t { return leaf_ref_of[(size_t)(&leaf - hs.nodes.data())]; };
	uint32_t max_depth = 1;
	if (root.child[0] >= 0) {
		std::vector<int64_t> dev_index(hs.nodes.size(), -1);
		uint32_t n_inner = 0;
		for (size_t i = 0; i < hs.nodes.size(); ++i) // host order is already preorder
			if (hs.nodes[i].child[0] >= 0)
				dev_index[i] = n_inner++;
		hs.dev_nodes.resize(n_inner);
		hs.root_ref = 0;
		// depth of the inner-node tree (stack entries needed by the depth-first walk): children follow their
		// parent in preorder, so one forward pass settles every level
		std::vector<uint32_t> level(hs.nodes.size(), 0u);
		level[0] = 1;
		for (size_t i = 0; i < hs.nodes.size(); ++i) {
			const HostNode &hn = hs.nodes[i];
			if (hn.child[0] < 0)
				continue;
			max_depth = std::max(max_depth, level[i]);
			level[(size_t)hn.child[0]] = level[(size_t)hn.child[1]] = level[i] + 1;
		}
		parallel_for(hs.nodes.size(), [&](uint64_t begin, uint64_t end) {
			for (uint64_t id = begin; id < end; ++id) {
				const HostNode &hn = hs.nodes[id];
				if (hn.child[0] < 0)
					continue;
				DevNode &dn = hs.dev_nodes[(size_t)dev_index[id]];
				dn.pad0 = dn.pad1 = 0;
				for (int c = 0; c < 2; ++c) {
					const HostNode &ch = hs.nodes[(size_t)hn.child[c]];
					float *mn = c == 0 ? dn.c0min : dn.c1min, *mx = c == 0 ? dn.c0max : dn.c1max;
					std::memcpy(mn, ch.min, 12);
					std::memcpy(mx, ch.max, 12);
					(c == 0 ? dn.c0 : dn.c1) = ch.child[0] >= 0 ? (uint32_t)dev_index[(size_t)hn.child[c]] : leaf_ref(ch);
				}
			}
		});
	} else {
		hs.dev_nodes.resize(1);
		std::memset(hs.dev_nodes.data(), 0, sizeof(DevNode));
		hs.root_ref = leaf_ref(root);
	}
	hs.stack_depth = max_depth + 1;
	hs.stack_depth_narrow = hs.stack_depth;

	// ---- wide tree: the reference tree collapsed to up to four children per node (rt_types.h DevNodeQ4).
	// A node's children start as its two reference children; the inner child with the largest surface area is
	// replaced by ITS two children until there are four (or only leaves are left).  Leaves are the reference's leaves.
	// Child boxes go onto a per-node grid, origin + q * 2^e with q in 0..255, rounded outward in exact (double)
	// arithmetic, so the stored box CONTAINS the reference box as a set of real numbers: the walk over it visits a
	// superset of the reference's nodes, and the leaf's exact box (leaf_box) decides candidacy (rt_intersect.h).
	// Built only when every bound is finite and of magnitude <= 2^60 (the regular-ray argument needs products of
	// bounds and inverse directions to stay finite). ----
	hs.dev_nodes4.clear();
	hs.leaf_box.clear();
	hs.dev_nodes4c.clear();
	hs.leaf_box_c.clear();
	hs.root4_ref = hs.root_ref;
	bool tame_bounds = true;
	for (const HostNode &hn : hs.nodes)
		for (int k = 0; k < 3; ++k)
			tame_bounds = tame_bounds && std::fabs(hn.min[k]) <= 0x1p60f && std::fabs(hn.max[k]) <= 0x1p60f; // false for NaN
	if (tame_bounds && root.child[0] >= 0) {
		auto area = [&](const HostNode &b) {
			const double dx = (double)b.max[0] - b.min[0], dy = (double)b.max[1] - b.min[1], dz = (double)b.max[2] - b.min[2];
			return dx * dy + dy * dz + dz * dx;
		};
		struct Pending { uint64_t host; uint32_t wide; uint32_t need; };
		std::vector<Pending> todo;
		std::vector<uint64_t> wide_host; // the reference node each wide node stands for
		hs.dev_nodes4.reserve(hs.nodes.size() / 3 + 16);
		hs.dev_nodes4.emplace_back();
		wide_host.push_back(0);
		hs.root4_ref = 0;
		todo.push_back({0, 0, 0});
		uint32_t wide_stack = 1;
		bool ok = true;
		while (!todo.empty() && ok) {
			const Pending it = todo.back();
			todo.pop_back();
			uint64_t kids[4];
			int n_kids = 2;
			kids[0] = (uint64_t)hs.nodes[it.host].child[0];
			kids[1] = (uint64_t)hs.nodes[it.host].child[1];
			while (n_kids < 4) {
				int open = -1;
				double best = -1.0;
				for (int k = 0; k < n_kids; ++k)
					if (hs.nodes[kids[k]].child[0] >= 0 && area(hs.nodes[kids[k]]) > best) {
						best = area(hs.nodes[kids[k]]);
						open = k;
					}
				if (open < 0)
					break;
				const HostNode &o = hs.nodes[kids[open]];
				for (int k = n_kids; k > open + 1; --k) // keep the reference's left-to-right order
					kids[k] = kids[k - 1];
				kids[open] = (uint64_t)o.child[0];
				kids[open + 1] = (uint64_t)o.child[1];
				++n_kids;
			}
			// a depth-first walk that descends into one child leaves at most n_kids - 1 siblings on its stack
			const uint32_t need_below = it.need + (uint32_t)(n_kids - 1);
			wide_stack = std::max(wide_stack, need_below + 1);
			DevNodeQ4 dn;
			std::memset(&dn, 0, sizeof dn);
			const HostNode &self = hs.nodes[it.host]; // contains every child exactly (mod.rs:105-108)
			for (int a = 0; a < 3; ++a) {
				dn.origin[a] = self.min[a];
				// smallest power-of-two step with origin + 255 * step >= max (all in double: f32 bounds <= 2^60 are exact there)
				const double extent = (double)self.max[a] - (double)self.min[a];
				int e = -126;
				if (extent > 0.0) {
					int ex;
					(void)std::frexp(extent / 255.0, &ex); // extent / 255 = m * 2^ex, m in [0.5, 1)  =>  2^ex >= extent / 255
					e = std::max(ex, -126);
				}
				while ((double)self.min[a] + 255.0 * std::ldexp(1.0, e) < (double)self.max[a])
					++e;
				if (e > 127) {
					ok = false;
					break;
				}
				dn.exps |= (uint32_t)(e + 127) << (8 * a);
				const double step = std::ldexp(1.0, e);
				for (int k = 0; k < n_kids; ++k) {
					const HostNode &ch = hs.nodes[kids[k]];
					long lo = (long)std::floor(((double)ch.min[a] - (double)self.min[a]) / step);
					long hi = (long)std::ceil(((double)ch.max[a] - (double)self.min[a]) / step);
					lo = std::min(std::max(lo, 0l), 255l);
					hi = std::min(std::max(hi, 0l), 255l);
					while (lo > 0 && (double)self.min[a] + (double)lo * step > (double)ch.min[a])
						--lo;
					while (hi < 255 && (double)self.min[a] + (double)hi * step < (double)ch.max[a])
						++hi;
					if ((double)self.min[a] + (double)lo * step > (double)ch.min[a] || (double)self.min[a] + (double)hi * step < (double)ch.max[a])
						ok = false; // cannot happen (the child lies inside `self`); never ship a box that does not contain
					dn.qlo[a] |= (uint32_t)lo << (8 * k);
					dn.qhi[a] |= (uint32_t)hi << (8 * k);
				}
				for (int k = n_kids; k < 4; ++k) // absent children: an inverted interval (and child == kRefNone)
					dn.qlo[a] |= 255u << (8 * k);
			}
			for (int k = 0; k < 4 && ok; ++k) {
				if (k >= n_kids) {
					dn.child[k] = kRefNone;
					continue;
				}
				const HostNode &ch = hs.nodes[kids[k]];
				if (ch.child[0] >= 0) {
					const uint32_t w = (uint32_t)hs.dev_nodes4.size();
					hs.dev_nodes4.emplace_back();
					wide_host.push_back(kids[k]);
					dn.child[k] = w;
					todo.push_back({kids[k], w, need_below});
				} else {
					dn.child[k] = leaf_ref(ch);
				}
			}
			hs.dev_nodes4[it.wide] = dn;
		}
		if (!ok || wide_stack > 96) { // (the stack would not fit the per-lane LDS column): keep the two-child tree only
			hs.dev_nodes4.clear();
			hs.root4_ref = hs.root_ref;
		} else {
			hs.stack_depth = std::max(hs.stack_depth, wide_stack);
			// exact boxes of the leaves, by first slot
			hs.leaf_box.assign(n, DevLeafBox{});
			for (const HostNode &hn : hs.nodes)
				if (hn.child[0] < 0) {
					DevLeafBox &b = hs.leaf_box[hn.primitive_offset];
					std::memcpy(b.lo, hn.min, 12);
					std::memcpy(b.hi, hn.max, 12);
				}
			// ---- the compact form the kernels walk (rt_types.h): inner children of a node are consecutive nodes (they were
			// created so), its leaf children get consecutive LEAF INDICES, and the node stores the two bases, a leaf mask
			// and a 2-bit offset per child instead of four 32-bit references.  Checked by decoding every node back. ----
			hs.dev_nodes4c.assign(hs.dev_nodes4.size(), DevNodeQ4{});
			hs.leaf_box_c.clear();
			hs.leaf_box_c.reserve(hs.nodes.size() / 2 + 1);
			for (size_t w = 0; w < hs.dev_nodes4.size() && ok; ++w) {
				const DevNodeQ4 &src = hs.dev_nodes4[w];
				DevNodeQ4 dn = src;
				uint32_t inner_base = 0, n_inner = 0, n_leaf = 0, leaf_mask = 0, deltas = 0, present = 0;
				const uint32_t leaf_base = (uint32_t)hs.leaf_box_c.size();
				for (int k = 0; k < 4; ++k) {
					const uint32_t c = src.child[k];
					if (c == kRefNone)
						continue; // (its stored interval is inverted on every axis AND its bit of the present mask stays clear: the
						          // padded box test alone lets an inverted interval through once the node is small and far away)
					present |= 1u << k;
					if (c & kLeafFlag) {
						uint32_t first = c & kLeafSlotMask;
						if (((c >> 26) & 31u) == 0u)
							first = hs.big_leaves[2 * (size_t)first];
						DevLeafBox lb = hs.leaf_box[first];
						std::memcpy(&lb.pad0, &c, 4); // the leaf's own reference: (count, first slot) or a big-leaf index
						hs.leaf_box_c.push_back(lb);
						leaf_mask |= 1u << k;
						deltas |= n_leaf << (2 * k);
						++n_leaf;
					} else {
						if (n_inner == 0)
							inner_base = c;
						ok = ok && c == inner_base + n_inner;
						deltas |= n_inner << (2 * k);
						++n_inner;
					}
				}
				ok = ok && inner_base < (1u << 26) && leaf_base + n_leaf < (1u << 26);
				dn.exps = (src.exps & 0x00FFFFFFu) | (deltas << 24);
				dn.child[0] = inner_base | (leaf_mask << 26);
				dn.child[1] = leaf_base | (present << 26);
				dn.child[2] = dn.child[3] = 0u; // (never fetched)
				hs.dev_nodes4c[w] = dn;
				// the walk looks up the present bits of slots 2 and 3 only: a collapsed reference node has its two children at least,
				// and absent children are the trailing ones
				ok = ok && (present == 0x3u || present == 0x7u || present == 0xFu);
				for (int k = 0; k < 4 && ok; ++k) { // decode as the kernel does (rt_intersect.h descend4: ref_of, present bits) and compare
					const uint32_t c = src.child[k];
					const bool is_present = ((dn.child[1] >> (26 + k)) & 1u) != 0u;
					ok = ok && is_present == (c != kRefNone);
					if (c == kRefNone) { // absent: never visited, and its interval stays inverted on every axis all the same
						for (int a = 0; a < 3; ++a)
							ok = ok && ((dn.qlo[a] >> (8 * k)) & 255u) == 255u && ((dn.qhi[a] >> (8 * k)) & 255u) == 0u;
						continue;
					}
					const uint32_t delta = (dn.exps >> (24 + 2 * k)) & 3u;
					const bool is_leaf = ((dn.child[0] >> (26 + k)) & 1u) != 0u;
					if (is_leaf) {
						uint32_t back;
						std::memcpy(&back, &hs.leaf_box_c[(dn.child[1] & kLeafSlotMask) + delta].pad0, 4);
						ok = ok && back == c;
					} else {
						ok = ok && (dn.child[0] & kLeafSlotMask) + delta == c;
					}
				}
			}
			if (!ok) { // cannot happen; never ship a tree that does not decode to itself
				hs.dev_nodes4.clear();
				hs.dev_nodes4c.clear();
				hs.leaf_box.clear();
				hs.leaf_box_c.clear();
				hs.root4_ref = hs.root_ref;
				hs.stack_depth = hs.stack_depth_narrow;
			}
		}
	}
	if (hs.stack_depth > 96) {
		err = "BVH deeper than 95 inner levels: the per-lane LDS traversal stack does not fit";
		return RT_ERR_UNSUPPORTED;
	}

	lap("device nodes");
	// ---- Sky::new  sky.rs:22-39 ----
	const uint64_t rx = d->sky.sampler_res_x, ry = d->sky.sampler_res_y;
	hs.sky_cdf.clear();
	if ((rx | ry) != 0) {
		if (rx == 0 || ry == 0) {
			err = "sky sampler_res must be both zero or both non-zero (Distribution2D::new would panic)";
			return RT_ERR_INVALID_ARGUMENT;
		}
		std::vector<float> values(rx * ry);
		const float step0 = 1.0f / (float)rx, step1 = 1.0f / (float)ry;
		size_t k = 0;
		for (uint64_t y = 0; y < ry; ++y)
			for (uint64_t x = 0; x < rx; ++x) { // generate_values  textures/mod.rs:32-50
				const float u = ((float)x + 0.5f) * step0;
				const float v = ((float)y + 0.5f) * step1;
				const float phi = u * 2.0f * kPi;
				const float theta = v * kPi;
				const float sin_theta = rt_sinf(theta);
				const V3 direction = v3(rt_cosf(phi) * sin_theta, rt_sinf(phi) * sin_theta, rt_cosf(theta));
				const V3 col = host_texture_colour(hs, d->sky.texture, direction, v3s(0.0f));
				values[k++] = (0.2126f * col.x + 0.7152f * col.y + 0.0722f * col.z) * sin_theta;
			}
		hs.sky_cdf.resize(ry * (rx + 1) + ry + 1);
		std::vector<float> y_values(ry);
		for (uint64_t r = 0; r < ry; ++r) { // Distribution2D::new  distributions.rs:83-99
			make_cdf(values.data() + r * rx, rx, hs.sky_cdf.data() + r * (rx + 1));
			float row_sum = 0.0f;
			for (uint64_t i = 0; i < rx; ++i)
				row_sum += values[r * rx + i];
			y_values[r] = row_sum;
		}
		make_cdf(y_values.data(), ry, hs.sky_cdf.data() + ry * (rx + 1));

		// guide tables (rt_types.h DevSky): only for tables the byte entries can index and whose CDFs are
		// finite and non-decreasing (then a rightward scan from the guide finds the binary search's index)
		hs.sky_guide.clear();
		hs.sky_guide_k = 0;
		bool usable = rx <= 254 && ry <= 254;
		for (float v : hs.sky_cdf)
			usable = usable && std::isfinite(v);
		for (uint64_t r = 0; r <= ry && usable; ++r) {
			const float *cdf = hs.sky_cdf.data() + r * (rx + 1);
			const uint64_t len = (r < ry ? rx : ry) + 1;
			for (uint64_t i = 1; i < len; ++i)
				usable = usable && cdf[i - 1] <= cdf[i];
		}
		if (usable) {
			uint32_t K = 16;
			while (K < std::max(rx, ry) && K < 256)
				K *= 2;
			hs.sky_guide_k = K;
			hs.sky_guide.assign((size_t)(ry + 1) * K, 0);
			for (uint64_t r = 0; r <= ry; ++r) {
				const float *cdf = hs.sky_cdf.data() + r * (rx + 1);
				const uint64_t len = (r < ry ? rx : ry) + 1; // entries 0..n
				uint64_t first = 0;
				for (uint32_t k = 0; k < K; ++k) {
					const float threshold = (float)k / (float)K; // exact
					while (first < len && cdf[first] <= threshold)
						++first; // upper bound: first index with cdf > threshold
					hs.sky_guide[r * K + k] = (uint8_t)first;
				}
			}
		}
	}
	return RT_OK;
}

} // namespace rt
