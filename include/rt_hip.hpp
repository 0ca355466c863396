// rt_hip.hpp -- header-only C++17 host side above the C ABI (rt_hip.h).
//
// The reference's host is compiled Rust; no Rust toolchain exists in this pipeline, so this header is
// the compiled-language twin of the binding in INTEGRATION.md.  It mirrors the reference's names and
// argument meaning for the one path the library replaces:
//   SceneBuilder            what crates/loader builds inside the Region arena (textures, materials, primitives, sky)
//   Bvh                     Bvh::new(primitives, sky, split_type)            acceleration/mod.rs:58-93
//   SimpleCamera            SimpleCamera::new                                camera.rs:20-54
//   RenderOptions, RenderMethod, SamplerProgress                             samplers/mod.rs:22-63
//   HipSampler::sample_image(options, camera, bvh, (data, callback))         samplers/mod.rs:7-20, random_sampler.rs:10-99
// Errors are exceptions carrying the rt_status and rt_last_error() text.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "rt_hip.h"

namespace rt_hip {

struct Error : std::runtime_error {
	int code;
	Error(int c, const std::string &what) : std::runtime_error("rt_hip error " + std::to_string(c) + ": " + what), code(c) {}
};
inline void check(int rc)
{
	if (rc != RT_OK)
		throw Error(rc, rt_last_error());
}

struct Vec3 {
	float x, y, z;
};

enum class RenderMethod { Naive = RT_METHOD_NAIVE, MIS = RT_METHOD_MIS }; // samplers/mod.rs:43-47
enum class SplitType { Sah = RT_SPLIT_SAH, Middle = RT_SPLIT_MIDDLE, EqualCounts = RT_SPLIT_EQUAL_COUNTS };

struct RenderOptions { // samplers/mod.rs:22-41 (Default impl)
	uint64_t samples_per_pixel = 128;
	RenderMethod render_method = RenderMethod::MIS;
	uint64_t width = 1920;
	uint64_t height = 1080;
	float gamma = 2.2f;
};

struct SamplerProgress { // samplers/mod.rs:49-63
	uint64_t samples_completed = 0;
	uint64_t rays_shot = 0;
	std::vector<float> current_image;
	SamplerProgress(uint64_t pixel_num, uint64_t channels) : current_image(pixel_num * channels, 0.0f) {}
};
// `&SamplerProgress` as the presentation callback sees it: the image lives in the sampler's pinned buffer
struct SamplerProgressRef {
	uint64_t samples_completed = 0;
	uint64_t rays_shot = 0;
	struct Image {
		const float *ptr = nullptr;
		size_t n = 0;
		const float *data() const { return ptr; }
		size_t size() const { return n; }
		float operator[](size_t k) const { return ptr[k]; }
	} current_image;
};

class SimpleCamera { // camera.rs:6-54
  public:
	SimpleCamera(Vec3 origin, Vec3 lookat, Vec3 vup, float fov, float aspect_ratio, float aperture, float focus_dist)
	{
		const float o[3] = {origin.x, origin.y, origin.z}, l[3] = {lookat.x, lookat.y, lookat.z}, u[3] = {vup.x, vup.y, vup.z};
		check(rt_camera_new(&cam_, o, l, u, fov, aspect_ratio, aperture, focus_dist));
	}
	const rt_camera &raw() const { return cam_; }

  private:
	rt_camera cam_{};
};

// Everything Bvh::new consumes.  Handles returned by the add-functions are indices, the way the
// loader resolves names to RegionRes handles (loader/src/lib.rs:28-84).
class SceneBuilder {
  public:
	uint32_t solid(Vec3 colour) { return texture(RT_TEX_SOLID, colour, {0, 0, 0}); }                       // SolidColour::new
	uint32_t lerp(Vec3 colour_one, Vec3 colour_two) { return texture(RT_TEX_LERP, colour_one, colour_two); } // Lerp::new
	uint32_t checkered(Vec3 one, Vec3 two) { return texture(RT_TEX_CHECKERED, one, two); }                  // CheckeredTexture::new

	uint32_t emissive(uint32_t tex, float strength) { return material(RT_MAT_EMIT, tex, strength); }        // Emit::new
	uint32_t lambertian(uint32_t tex, float albedo) { return material(RT_MAT_LAMBERTIAN, tex, albedo); }    // Lambertian::new
	uint32_t reflect(uint32_t tex, float fuzz) { return material(RT_MAT_REFLECT, tex, fuzz); }              // Reflect::new
	uint32_t refract(uint32_t tex, float eta) { return material(RT_MAT_REFRACT, tex, eta); }                // Refract::new
	uint32_t trowbridge_reitz(uint32_t tex, float roughness, Vec3 ior, float metallic)                      // TrowbridgeReitz::new
	{
		const uint32_t m = material(RT_MAT_TROWBRIDGE_REITZ, tex, roughness * roughness); // alpha = roughness^2 (:17-24)
		materials_[m].ior[0] = ior.x; materials_[m].ior[1] = ior.y; materials_[m].ior[2] = ior.z;
		materials_[m].metallic = metallic;
		return m;
	}

	void sphere(Vec3 centre, float radius, uint32_t mat) // Sphere::new
	{
		rt_primitive_desc p{};
		p.type = RT_PRIM_SPHERE;
		p.material = mat;
		p.u.sphere.centre[0] = centre.x; p.u.sphere.centre[1] = centre.y; p.u.sphere.centre[2] = centre.z;
		p.u.sphere.radius = radius;
		primitives_.push_back(p);
	}
	void triangle(const Vec3 points[3], const Vec3 normals[3], uint32_t mat) // Triangle::new
	{
		rt_triangle_data t{};
		for (int k = 0; k < 3; ++k) {
			t.points[3 * k] = points[k].x; t.points[3 * k + 1] = points[k].y; t.points[3 * k + 2] = points[k].z;
			t.normals[3 * k] = normals[k].x; t.normals[3 * k + 1] = normals[k].y; t.normals[3 * k + 2] = normals[k].z;
		}
		triangles_.push_back(t);
		rt_primitive_desc p{};
		p.type = RT_PRIM_TRIANGLE;
		p.material = mat;
		p.u.triangle.data = triangles_.size() - 1;
		primitives_.push_back(p);
	}
	// Sky::new(texture, Emit(texture, 1.0), sampler_res)  sky.rs:22-39, loader/src/misc.rs:20-38
	void sky(uint32_t tex, uint32_t res_x = 100, uint32_t res_y = 100)
	{
		sky_.texture = tex;
		sky_.material = emissive(tex, 1.0f);
		sky_.sampler_res_x = res_x;
		sky_.sampler_res_y = res_y;
		has_sky_ = true;
	}

	rt_scene_desc desc(SplitType split) const
	{
		if (!has_sky_)
			throw Error(RT_ERR_INVALID_ARGUMENT, "scene has no sky");
		rt_scene_desc d{};
		d.abi_version = RT_ABI_VERSION;
		d.n_textures = (uint32_t)textures_.size();
		d.textures = textures_.data();
		d.n_materials = (uint32_t)materials_.size();
		d.materials = materials_.data();
		d.n_primitives = primitives_.size();
		d.primitives = primitives_.data();
		d.n_triangles = triangles_.size();
		d.triangles = triangles_.data();
		d.sky = sky_;
		d.split_type = (int32_t)split;
		return d;
	}

  private:
	uint32_t texture(int type, Vec3 a, Vec3 b)
	{
		rt_texture_desc t{};
		t.type = type;
		t.colour_one[0] = a.x; t.colour_one[1] = a.y; t.colour_one[2] = a.z;
		t.colour_two[0] = b.x; t.colour_two[1] = b.y; t.colour_two[2] = b.z;
		textures_.push_back(t);
		return (uint32_t)textures_.size() - 1;
	}
	uint32_t material(int type, uint32_t tex, float param)
	{
		rt_material_desc m{};
		m.type = type;
		m.texture = tex;
		m.param = param;
		m.ior[0] = m.ior[1] = m.ior[2] = 1.0f;
		materials_.push_back(m);
		return (uint32_t)materials_.size() - 1;
	}
	std::vector<rt_texture_desc> textures_;
	std::vector<rt_material_desc> materials_;
	std::vector<rt_primitive_desc> primitives_;
	std::vector<rt_triangle_data> triangles_;
	rt_sky_desc sky_{};
	bool has_sky_ = false;
};

class Bvh { // acceleration/mod.rs:44-93, resident in the HBM of `device`
  public:
	Bvh(const SceneBuilder &scene, SplitType split = SplitType::Sah, int device = 0)
	{
		const rt_scene_desc d = scene.desc(split);
		check(rt_scene_create(&d, device, &h_));
	}
	// one Bvh replicated over several GPUs of the node: every render through it shards the frame's tiles over `devices`
	// and gathers into devices[0] (rt_scene_create_multi; the partition of samplers/random_sampler.rs:45-52 across devices)
	Bvh(const SceneBuilder &scene, const std::vector<int> &devices, SplitType split = SplitType::Sah)
	{
		const rt_scene_desc d = scene.desc(split);
		check(rt_scene_create_multi(&d, devices.data(), (uint32_t)devices.size(), &h_));
	}
	uint32_t device_count() const
	{
		uint32_t n = 0;
		check(rt_scene_device_count(h_, &n));
		return n;
	}
	// how a multi-device Bvh moves its members' shards into devices[0], and why (rt_gather_mode; settled at construction)
	std::pair<int, std::string> gather_info() const
	{
		int mode = 0;
		char note[512] = {0};
		check(rt_scene_gather_info(h_, &mode, note, sizeof note));
		return {mode, std::string(note)};
	}
	// what rt_render_opts.sample_split = 0 resolves to for these options on this Bvh (the library's one rule)
	uint32_t auto_sample_split(const rt_render_opts &opts) const
	{
		uint32_t split = 1;
		check(rt_scene_auto_sample_split(h_, &opts, &split));
		return split;
	}
	~Bvh() { rt_scene_destroy(h_); }
	Bvh(const Bvh &) = delete;
	Bvh &operator=(const Bvh &) = delete;
	uint64_t number_nodes() const // Bvh::number_nodes  mod.rs:94-96
	{
		uint64_t n = 0;
		check(rt_scene_counts(h_, &n, nullptr, nullptr));
		return n;
	}
	std::vector<uint64_t> lights() const // pub lights  mod.rs:48
	{
		uint64_t n = 0;
		check(rt_scene_counts(h_, nullptr, nullptr, &n));
		std::vector<uint64_t> out(n ? n : 1);
		check(rt_scene_get_lights(h_, out.data(), out.size()));
		out.resize(n);
		return out;
	}
	rt_scene *raw() const { return h_; }

  private:
	rt_scene *h_ = nullptr;
};

// `impl Sampler` on the GPU.  The reference calls the update function after EVERY pass with that
// pass's image (random_sampler.rs:82-98); this sampler renders `batch` passes per call of the ABI and
// hands the callback the batch's mean, progress.samples_completed = passes in the batch, and the total
// number of passes so far -- with batch = 1 that is the reference's contract verbatim.  Returning true
// cancels.  `running_mean` is the TUI callback's accumulation (src/main.rs:175-191) for any batch size.
struct HipSampler {
	uint64_t batch = 0; // 0 = all passes in one launch
	uint64_t seed = 1;
	uint32_t sample_split = 1; // rt_render_opts.sample_split; 0 = automatic (>= 64 work items per resident lane on every device: 7 - 29 % faster
	                           // than whole-pixel items on the BASELINE workloads, image moves by ~1e-7); 1 = the reference's strictly sequential fold
	uint32_t max_depth = 50, rr_threshold = 3; // integrators/mod.rs:7-8

	// update(data, previous, i) -> bool, the reference's presentation_update: called once per batch with the
	// batch's mean image while the next batch renders; `true` cancels (random_sampler.rs:82-98)
	template <class T, class F>
	void sample_image(const RenderOptions &o, const SimpleCamera &camera, const Bvh &bvh, T *data, F update) const
	{
		rt_render_opts opts;
		rt_render_opts_default(&opts);
		opts.width = o.width;
		opts.height = o.height;
		opts.samples_per_pixel = o.samples_per_pixel;
		opts.seed = seed;
		opts.render_method = (int32_t)o.render_method;
		opts.max_depth = max_depth;
		opts.rr_threshold = rr_threshold;
		opts.sample_split = sample_split;
		struct Closure {
			T *data;
			F *update;
		} closure{data, &update};
		check(rt_sample_image(bvh.raw(), &camera.raw(), &opts, batch,
		                      [](void *c, const rt_sampler_progress *p, uint64_t done) -> int {
			                      auto *cl = static_cast<Closure *>(c);
			                      SamplerProgressRef ref;
			                      ref.samples_completed = p->samples_completed;
			                      ref.rays_shot = p->rays_shot;
			                      ref.current_image.ptr = p->current_image;
			                      ref.current_image.n = (size_t)p->n_floats;
			                      return (*cl->update)(cl->data, ref, done) ? 1 : 0;
		                      },
		                      &closure));
	}
};

// The output stage on the GPU (crates/output/src/lib.rs:89-97): renders and returns the 8-bit RGB image
// save_data_to_image would write, `(val.powf(1.0 / gamma) * 255.999) as u8` computed where the frame lives.
inline std::vector<uint8_t> render_rgb8(const RenderOptions &o, const SimpleCamera &camera, const Bvh &bvh, uint64_t seed = 1,
                                        uint64_t *rays_shot = nullptr)
{
	rt_render_opts opts;
	rt_render_opts_default(&opts);
	opts.width = o.width;
	opts.height = o.height;
	opts.samples_per_pixel = o.samples_per_pixel;
	opts.seed = seed;
	opts.render_method = (int32_t)o.render_method;
	std::vector<uint8_t> out((size_t)o.width * o.height * 3);
	check(rt_render_rgb8(bvh.raw(), &camera.raw(), &opts, o.gamma, out.data(), rays_shot));
	return out;
}

struct Presentation { // what render_tui keeps: the mean image and the ray total (src/main.rs:160-173)
	SamplerProgress sampler_progress;
	Presentation(uint64_t pixel_num) : sampler_progress(pixel_num, 3) {}
};
inline bool running_mean(Presentation *sp, const SamplerProgressRef &previous, uint64_t i)
{
	sp->sampler_progress.samples_completed += previous.samples_completed;
	sp->sampler_progress.rays_shot += previous.rays_shot;
	const float w = (float)previous.samples_completed / (float)i;
	for (size_t k = 0; k < previous.current_image.size(); ++k) {
		float &pres = sp->sampler_progress.current_image[k];
		pres += (previous.current_image[k] - pres) * w; // *pres += (acc - *pres) / i for batch = 1
	}
	return false;
}

} // namespace rt_hip
