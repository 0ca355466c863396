/*
 * ora_render.c -- CPU oracle: the two integrators, the image sampler, and the library's entry
 * points.  TEST INFRASTRUCTURE (see ora_internal.h).
 * Restates crates/implementations/src/{integrators/mod.rs, integrators/mis.rs,
 * samplers/random_sampler.rs, camera.rs} and the running-mean callback of src/main.rs:175-191.
 */
#include "ora_internal.h"
#include <pthread.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static _Thread_local char g_err[256];
const char *ora_last_error(void) { return g_err; }
static int fail(int code, const char *msg)
{
	snprintf(g_err, sizeof g_err, "%s", msg);
	return code;
}

/* rt_core/src/lib.rs:36-40 */
static inline float power_heuristic(float pdf_a, float pdf_b)
{
	const float a_sq = pdf_a * pdf_a;
	return a_sq / (a_sq + pdf_b * pdf_b);
}

/* ---------------- NaiveIntegrator::get_colour  integrators/mod.rs:22-78 ---------------- */
vec3 ora_naive_get_colour(const ora_scene *s, ora_ray *ray, uint32_t max_depth, uint32_t rr_threshold,
                          uint64_t *ray_count_out, ora_ctx *ctx)
{
	vec3 throughput = v3_one(), output = v3_zero();
	uint32_t depth = 0;
	uint64_t ray_count = 0;

	while (depth < max_depth) {
		ora_si si;
		(void)ora_bvh_check_hit(s, ray, &si, ctx);
		ray_count += 1;

		const ora_hit *hit = &si.hit;
		const uint32_t mat = si.material;
		const vec3 wo = ray->direction;

		const vec3 emission = ora_mat_get_emission(s, mat, hit, wo);
		const bool exit = ora_mat_scatter_ray(s, mat, ray, hit, ctx);

		if (depth == 0) {
			output = v3_add(output, emission);
			if (exit)
				break;
		}
		if (exit) {
			output = v3_add(output, v3_mul(throughput, emission));
			break;
		}

		if (!ora_mat_is_delta(s, mat))
			throughput = v3_mul(throughput, ora_mat_eval_over_scattering_pdf(s, mat, hit, wo, ray->direction));
		else
			throughput = v3_mul(throughput, ora_mat_eval(s, mat, hit, wo, ray->direction));

		if (depth > rr_threshold) {
			const float p = v3_component_max(throughput);
			if (ora_random_float(ctx) > p)
				break;
			throughput = v3_divs(throughput, p);
		}
		depth += 1;
	}
	*ray_count_out = ray_count;
	if (v3_contains_nan(output) || !v3_is_finite(output))
		return v3_zero();
	return output;
}

/* ---------------- sample_lights  integrators/mis.rs:95-157 ---------------- */
typedef struct light_sample {
	vec3 l_wi, le;
	float l_pdf;
} light_sample;

static bool sample_sky(const ora_scene *s, const ora_hit *hit, float pdf_multiplier, light_sample *out, ora_ctx *ctx)
{
	const vec3 l_wi = ora_sky_sample(s, ctx);
	const ora_ray ray = ora_ray_new(v3_add(hit->point, v3_smul(0.0001f, hit->normal)), l_wi, 0.0f);
	ora_si sa;
	const uint64_t index = ora_bvh_check_hit(s, &ray, &sa, ctx);
	if (index == ORA_NO_INDEX) {
		out->l_wi = l_wi;
		out->le = ora_mat_get_emission(s, sa.material, hit, l_wi);
		out->l_pdf = ora_sky_pdf(s, l_wi, ctx) * pdf_multiplier;
		return true;
	}
	return false;
}

static bool sample_light(const ora_scene *s, const ora_hit *hit, float pdf_multiplier, uint64_t light_slot,
                         light_sample *out, ora_ctx *ctx)
{
	const uint64_t index = s->lights[light_slot];
	const ora_primitive *light = &s->primitives[index];
	const vec3 l_wi = ora_prim_sample_visible_from_point(s, light, hit->point, ctx);
	const ora_ray ray = ora_ray_new(v3_add(hit->point, v3_smul(0.0001f, hit->normal)), l_wi, 0.0f);
	ora_si si;
	if (ora_bvh_check_hit_index(s, &ray, index, &si, ctx)) {
		const float l_pdf = ora_prim_scattering_pdf(s, light, hit->point, l_wi, &si.hit);
		if (l_pdf > 0.0f) {
			out->l_wi = l_wi;
			out->le = ora_mat_get_emission(s, si.material, &si.hit, l_wi);
			out->l_pdf = l_pdf * pdf_multiplier;
			return true;
		}
	}
	return false;
}

static bool sample_lights(const ora_scene *s, const ora_hit *hit, light_sample *out, ora_ctx *ctx)
{
	const uint64_t samplable_len = s->n_lights;
	const bool sky_can_sample = ora_sky_can_sample(s);
	if (samplable_len == 0 && !sky_can_sample)
		return false;
	if (samplable_len == 0)
		return sample_sky(s, hit, 1.0f, out, ctx);
	if (!sky_can_sample) {
		const float multiplier = 1.0f / (float)samplable_len;
		ctx->c.rng_draws++;
		const uint64_t light_index = rt_rng_below(&ctx->rng, (uint32_t)samplable_len); /* gen_range(0..len) */
		return sample_light(s, hit, multiplier, light_index, out, ctx);
	}
	const float multiplier = 1.0f / (float)(samplable_len + 1);
	ctx->c.rng_draws++;
	const uint64_t light_index = rt_rng_below(&ctx->rng, (uint32_t)samplable_len + 1u); /* gen_range(0..=len) */
	if (light_index == samplable_len)
		return sample_sky(s, hit, multiplier, out, ctx);
	return sample_light(s, hit, multiplier, light_index, out, ctx);
}

static bool lights_contain(const ora_scene *s, uint64_t index) /* bvh.get_samplable().contains(&index) */
{
	for (uint64_t i = 0; i < s->n_lights; ++i)
		if (s->lights[i] == index)
			return true;
	return false;
}

/* ---------------- MisIntegrator::get_colour  integrators/mis.rs:7-92 ---------------- */
vec3 ora_mis_get_colour(const ora_scene *s, ora_ray *ray, uint32_t max_depth, uint32_t rr_threshold,
                        uint64_t *ray_count_out, ora_ctx *ctx)
{
	vec3 throughput = v3_one(), output = v3_zero();
	uint64_t ray_count = 0;

	ora_si surface_intersection;
	(void)ora_bvh_check_hit(s, ray, &surface_intersection, ctx);
	ora_hit hit = surface_intersection.hit;
	uint32_t mat = surface_intersection.material;
	vec3 wo = ray->direction;

	const vec3 emission = ora_mat_get_emission(s, mat, &hit, wo);
	ora_ray clone = *ray; /* scatter on a clone: draws are consumed, the ray is discarded (:25, P-hazard 10) */
	const bool exit0 = ora_mat_scatter_ray(s, mat, &clone, &hit, ctx);
	output = v3_add(output, emission);
	if (exit0) {
		*ray_count_out = ray_count;
		return output; /* :29-31 returns before the NaN filter */
	}

	uint32_t depth = 1;
	while (depth < max_depth) {
		light_sample ls;
		const bool have_ls = sample_lights(s, &hit, &ls, ctx);
		ray_count += 1;
		if (have_ls) {
			const float m_pdf = ora_mat_scattering_pdf(s, mat, &hit, wo, ls.l_wi);
			const float mis_weight = power_heuristic(ls.l_pdf, m_pdf);
			/* throughput * eval * mis_weight * le / l_pdf */
			const vec3 contrib = v3_divs(
			    v3_mul(v3_muls(v3_mul(throughput, ora_mat_eval(s, mat, &hit, wo, ls.l_wi)), mis_weight), ls.le),
			    ls.l_pdf);
			output = v3_add(output, contrib);
		}

		const bool exit = ora_mat_scatter_ray(s, mat, ray, &hit, ctx);
		if (exit)
			break;
		const vec3 m_wi = ray->direction;

		ora_si intersection;
		const uint64_t index = ora_bvh_check_hit(s, ray, &intersection, ctx);

		const float m_pdf = ora_mat_scattering_pdf(s, mat, &hit, wo, m_wi);
		const vec3 le = ora_mat_get_emission(s, intersection.material, &hit /* the OLD hit, :55 */, m_wi);
		throughput = v3_mul(throughput, ora_mat_eval_over_scattering_pdf(s, mat, &hit, wo, m_wi));
		if (!v3_eq(le, v3_zero())) {
			if ((lights_contain(s, index) && !ora_mat_is_delta(s, mat)) ||
			    (index == ORA_NO_INDEX && ora_sky_can_sample(s))) {
				const float l_pdf = ora_bvh_get_pdf_from_index(s, &hit, &intersection.hit, m_wi, index, ctx);
				const float mis_weight = power_heuristic(m_pdf, l_pdf);
				output = v3_add(output, v3_muls(v3_mul(throughput, le), mis_weight));
			} else {
				output = v3_add(output, v3_mul(throughput, le));
			}
		}

		if (ora_mat_is_light(s, intersection.material))
			break;

		if (depth > rr_threshold) {
			const float p = v3_component_max(throughput);
			if (ora_random_float(ctx) > p)
				break;
			throughput = v3_divs(throughput, p);
		}

		wo = m_wi;
		hit = intersection.hit;
		mat = intersection.material;
		depth += 1;
	}
	*ray_count_out = ray_count;
	if (v3_contains_nan(output) || !v3_is_finite(output))
		return v3_zero();
	return output;
}

/* ---------------- SimpleCamera  camera.rs:20-63 ---------------- */
int ora_camera_new(rt_camera *out, const float origin_[3], const float lookat_[3], const float vup_[3],
                   float fov, float aspect_ratio, float aperture, float focus_dist)
{
	(void)aperture; /* lens_radius is stored and never used: camera.rs:51,57-63 */
	if (!out || !origin_ || !lookat_ || !vup_)
		return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
	const vec3 origin = v3_from(origin_), lookat = v3_from(lookat_), vup = v3_from(vup_);
	const float viewport_width = 2.0f * rt_tanf(rt_to_radians(fov) / 2.0f);
	const float viewport_height = viewport_width / aspect_ratio;
	const vec3 w = v3_normalised(v3_sub(origin, lookat));
	const vec3 u = v3_normalised(v3_cross(w, vup));
	const vec3 v = v3_cross(u, w);
	const vec3 horizontal = v3_muls(v3_smul(focus_dist, u), viewport_width);
	const vec3 vertical = v3_muls(v3_smul(focus_dist, v), viewport_height);
	const vec3 lower_left =
	    v3_sub(v3_sub(v3_sub(origin, v3_divs(horizontal, 2.0f)), v3_divs(vertical, 2.0f)), v3_smul(focus_dist, w));
	out->origin[0] = origin.x; out->origin[1] = origin.y; out->origin[2] = origin.z;
	out->lower_left[0] = lower_left.x; out->lower_left[1] = lower_left.y; out->lower_left[2] = lower_left.z;
	out->horizontal[0] = horizontal.x; out->horizontal[1] = horizontal.y; out->horizontal[2] = horizontal.z;
	out->vertical[0] = vertical.x; out->vertical[1] = vertical.y; out->vertical[2] = vertical.z;
	return RT_OK;
}
/* camera.rs:57-63 */
static ora_ray camera_get_ray(const rt_camera *cam, float u, float v, ora_ctx *ctx)
{
	const vec3 origin = v3_from(cam->origin);
	const vec3 dir = v3_sub(
	    v3_add(v3_add(v3_from(cam->lower_left), v3_muls(v3_from(cam->horizontal), u)), v3_muls(v3_from(cam->vertical), v)),
	    origin);
	return ora_ray_new(origin, dir, ora_random_float(ctx));
}

/* ---------------- scene ---------------- */
int ora_scene_create(const rt_scene_desc *d, ora_scene **out)
{
	if (!d || !out)
		return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
	if (d->abi_version != RT_ABI_VERSION)
		return fail(RT_ERR_INVALID_ARGUMENT, "abi version mismatch");
	if (d->n_primitives == 0)
		return fail(RT_ERR_INVALID_ARGUMENT, "scene has no primitives (Bvh::new would panic)");
	ora_scene *s = (ora_scene *)calloc(1, sizeof *s);
	s->split_type = d->split_type;

	s->n_textures = d->n_textures;
	s->textures = (ora_texture *)calloc(d->n_textures ? d->n_textures : 1, sizeof(ora_texture));
	for (uint32_t i = 0; i < d->n_textures; ++i) {
		const rt_texture_desc *t = &d->textures[i];
		ora_texture *o = &s->textures[i];
		o->type = t->type;
		o->colour_one = v3_from(t->colour_one);
		o->colour_two = v3_from(t->colour_two);
		if (t->type == RT_TEX_IMAGE) {
			if (!t->image_rgb || t->image_width == 0 || t->image_height == 0) {
				ora_scene_destroy(s);
				return fail(RT_ERR_INVALID_ARGUMENT, "image texture without pixels");
			}
			const uint64_t n = (uint64_t)t->image_width * t->image_height;
			o->image = (vec3 *)malloc(n * sizeof(vec3));
			memcpy(o->image, t->image_rgb, n * sizeof(vec3));
			o->dim_x = t->image_width - 1; /* textures/mod.rs:232 */
			o->dim_y = t->image_height - 1;
		}
		if (t->type == RT_TEX_PERLIN) {
			if (!t->perlin_ran_vecs || !t->perlin_perm) {
				ora_scene_destroy(s);
				return fail(RT_ERR_INVALID_ARGUMENT, "perlin texture without tables");
			}
			memcpy(o->ran_vecs, t->perlin_ran_vecs, sizeof o->ran_vecs);
			memcpy(o->perm_x, t->perlin_perm, 256 * 4);
			memcpy(o->perm_y, t->perlin_perm + 256, 256 * 4);
			memcpy(o->perm_z, t->perlin_perm + 512, 256 * 4);
		}
	}

	s->n_materials = d->n_materials;
	s->materials = (ora_material *)calloc(d->n_materials ? d->n_materials : 1, sizeof(ora_material));
	for (uint32_t i = 0; i < d->n_materials; ++i) {
		const rt_material_desc *m = &d->materials[i];
		if (m->texture >= d->n_textures) {
			ora_scene_destroy(s);
			return fail(RT_ERR_INVALID_ARGUMENT, "material texture index out of range");
		}
		s->materials[i].type = m->type;
		s->materials[i].texture = m->texture;
		s->materials[i].param = m->param;
		s->materials[i].ior = v3_from(m->ior);
		s->materials[i].metallic = m->metallic;
	}

	s->n_meshes = d->n_meshes;
	s->meshes = (ora_mesh *)calloc(d->n_meshes ? d->n_meshes : 1, sizeof(ora_mesh));
	for (uint32_t i = 0; i < d->n_meshes; ++i) {
		const rt_mesh_desc *m = &d->meshes[i];
		s->meshes[i].n_vertices = m->n_vertices;
		s->meshes[i].n_normals = m->n_normals;
		s->meshes[i].vertices = (vec3 *)malloc((m->n_vertices ? m->n_vertices : 1) * sizeof(vec3));
		s->meshes[i].normals = (vec3 *)malloc((m->n_normals ? m->n_normals : 1) * sizeof(vec3));
		memcpy(s->meshes[i].vertices, m->vertices, m->n_vertices * sizeof(vec3));
		memcpy(s->meshes[i].normals, m->normals, m->n_normals * sizeof(vec3));
	}

	s->n_primitives = d->n_primitives;
	s->primitives = (ora_primitive *)calloc(d->n_primitives, sizeof(ora_primitive));
	for (uint64_t i = 0; i < d->n_primitives; ++i) {
		const rt_primitive_desc *p = &d->primitives[i];
		ora_primitive *o = &s->primitives[i];
		o->type = p->type;
		o->material = p->material;
		bool ok = p->material < d->n_materials;
		if (p->type == RT_PRIM_SPHERE) {
			o->center = v3_from(p->u.sphere.centre);
			o->radius = p->u.sphere.radius;
		} else if (p->type == RT_PRIM_TRIANGLE) {
			ok = ok && p->u.triangle.data < d->n_triangles;
			if (ok) {
				const rt_triangle_data *t = &d->triangles[p->u.triangle.data];
				for (int k = 0; k < 3; ++k) {
					o->points[k] = v3_from(&t->points[3 * k]);
					o->normals[k] = v3_from(&t->normals[3 * k]);
				}
			}
		} else if (p->type == RT_PRIM_MESH_TRIANGLE) {
			o->mesh = p->u.mesh_triangle.mesh;
			ok = ok && o->mesh < d->n_meshes;
			for (int k = 0; k < 3 && ok; ++k) {
				o->point_indices[k] = p->u.mesh_triangle.point_indices[k];
				o->normal_indices[k] = p->u.mesh_triangle.normal_indices[k];
				ok = o->point_indices[k] < d->meshes[o->mesh].n_vertices &&
				     o->normal_indices[k] < d->meshes[o->mesh].n_normals;
			}
		} else {
			ok = false;
		}
		if (!ok) {
			ora_scene_destroy(s);
			return fail(RT_ERR_INVALID_ARGUMENT, "primitive descriptor out of range");
		}
	}

	if (d->sky.texture >= d->n_textures || d->sky.material >= d->n_materials) {
		ora_scene_destroy(s);
		return fail(RT_ERR_INVALID_ARGUMENT, "sky texture/material index out of range");
	}
	s->sky.texture = d->sky.texture;
	s->sky.material = d->sky.material;
	s->sky.res_x = d->sky.sampler_res_x;
	s->sky.res_y = d->sky.sampler_res_y;
	if (ora_sky_build(s) != RT_OK) {
		ora_scene_destroy(s);
		return fail(RT_ERR_INVALID_ARGUMENT, "sky sampler_res must be both zero or both non-zero");
	}
	ora_bvh_build(s);
	*out = s;
	return RT_OK;
}

void ora_scene_destroy(ora_scene *s)
{
	if (!s)
		return;
	for (uint32_t i = 0; i < s->n_textures && s->textures; ++i)
		free(s->textures[i].image);
	free(s->textures);
	free(s->materials);
	for (uint32_t i = 0; i < s->n_meshes && s->meshes; ++i) {
		free(s->meshes[i].vertices);
		free(s->meshes[i].normals);
	}
	free(s->meshes);
	free(s->primitives);
	free(s->primitive_order);
	free(s->nodes);
	free(s->lights);
	ora_sky_free(s);
	free(s);
}

int ora_scene_counts(const ora_scene *s, uint64_t *n_nodes, uint64_t *n_primitives, uint64_t *n_lights)
{
	if (!s)
		return fail(RT_ERR_INVALID_ARGUMENT, "null scene");
	if (n_nodes) *n_nodes = s->n_nodes;
	if (n_primitives) *n_primitives = s->n_primitives;
	if (n_lights) *n_lights = s->n_lights;
	return RT_OK;
}
int ora_scene_get_nodes(const ora_scene *s, rt_bvh_node *out, uint64_t capacity)
{
	if (!s || !out || capacity < s->n_nodes)
		return fail(RT_ERR_INVALID_ARGUMENT, "bad arguments");
	for (uint64_t i = 0; i < s->n_nodes; ++i) {
		const ora_node *n = &s->nodes[i];
		out[i].min[0] = n->bounds.min.x; out[i].min[1] = n->bounds.min.y; out[i].min[2] = n->bounds.min.z;
		out[i].max[0] = n->bounds.max.x; out[i].max[1] = n->bounds.max.y; out[i].max[2] = n->bounds.max.z;
		out[i].children[0] = n->has_children ? (int64_t)n->children[0] : -1;
		out[i].children[1] = n->has_children ? (int64_t)n->children[1] : -1;
		out[i].primitive_offset = n->primitive_offset;
		out[i].number_primitives = n->number_primitives;
	}
	return RT_OK;
}
int ora_scene_get_primitive_order(const ora_scene *s, uint64_t *out, uint64_t capacity)
{
	if (!s || !out || capacity < s->n_primitives)
		return fail(RT_ERR_INVALID_ARGUMENT, "bad arguments");
	memcpy(out, s->primitive_order, s->n_primitives * sizeof(uint64_t));
	return RT_OK;
}
int ora_scene_get_lights(const ora_scene *s, uint64_t *out, uint64_t capacity)
{
	if (!s || !out || capacity < s->n_lights)
		return fail(RT_ERR_INVALID_ARGUMENT, "bad arguments");
	memcpy(out, s->lights, s->n_lights * sizeof(uint64_t));
	return RT_OK;
}

/* ---------------- hit queries for batches ---------------- */
static void fill_record(rt_hit_record *r, const ora_si *si, uint64_t index, bool found)
{
	r->t = si->hit.t;
	r->point[0] = si->hit.point.x; r->point[1] = si->hit.point.y; r->point[2] = si->hit.point.z;
	r->error[0] = si->hit.error.x; r->error[1] = si->hit.error.y; r->error[2] = si->hit.error.z;
	r->normal[0] = si->hit.normal.x; r->normal[1] = si->hit.normal.y; r->normal[2] = si->hit.normal.z;
	r->uv[0] = si->hit.uv.x; r->uv[1] = si->hit.uv.y;
	r->has_uv = si->hit.has_uv;
	r->out = si->hit.out;
	r->material = si->material;
	r->found = found;
	r->index = index;
}
int ora_check_hit(const ora_scene *s, const rt_ray_desc *rays, uint64_t n, rt_hit_record *out)
{
	if (!s || !rays || !out)
		return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
	ora_ctx ctx;
	ora_ctx_init(&ctx);
	for (uint64_t i = 0; i < n; ++i) {
		const ora_ray ray = ora_ray_new(v3_from(rays[i].origin), v3_from(rays[i].direction), 0.0f);
		ora_si si;
		const uint64_t index = ora_bvh_check_hit(s, &ray, &si, &ctx);
		fill_record(&out[i], &si, index, true);
	}
	ora_ctx_free(&ctx);
	return RT_OK;
}
int ora_check_hit_index(const ora_scene *s, const rt_ray_desc *rays, const uint64_t *object_index, uint64_t n,
                        rt_hit_record *out)
{
	if (!s || !rays || !out || !object_index)
		return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
	ora_ctx ctx;
	ora_ctx_init(&ctx);
	for (uint64_t i = 0; i < n; ++i) {
		if (object_index[i] >= s->n_primitives) {
			ora_ctx_free(&ctx);
			return fail(RT_ERR_INVALID_ARGUMENT, "object index out of range");
		}
		const ora_ray ray = ora_ray_new(v3_from(rays[i].origin), v3_from(rays[i].direction), 0.0f);
		ora_si si;
		memset(&si, 0, sizeof si);
		const bool found = ora_bvh_check_hit_index(s, &ray, object_index[i], &si, &ctx);
		if (!found)
			memset(&si, 0, sizeof si);
		fill_record(&out[i], &si, object_index[i], found);
	}
	ora_ctx_free(&ctx);
	return RT_OK;
}

/* ---------------- RandomSampler::sample_image  samplers/random_sampler.rs:10-99 ---------------- */
#define PIXEL_CHUNK_SIZE 10000u /* random_sampler.rs:31 */

typedef struct render_job {
	const ora_scene *scene;
	const rt_camera *camera;
	rt_render_opts opts;
	float *mean; /* W*H*3 running mean (the TUI's presentation buffer, src/main.rs:160-185) */
	float *acc;  /* sample_split > 1: sum over finished chunks of their sums */
	uint32_t split;
	uint64_t n_chunks;
	atomic_uint_fast64_t next_chunk;
	atomic_uint_fast64_t rays_shot;
	pthread_barrier_t barrier;
	uint32_t n_threads;
	uint32_t tile_w, tile_h;
	ora_counters_i *thread_counters;
} render_job;

static inline bool pixel_owned(const render_job *j, uint64_t x, uint64_t y)
{
	if (j->opts.shard_count <= 1)
		return true;
	const uint64_t tiles_x = (j->opts.width + j->tile_w - 1) / j->tile_w;
	const uint64_t tile = (y / j->tile_h) * tiles_x + (x / j->tile_w);
	return tile % j->opts.shard_count == j->opts.shard_index;
}

static void render_chunk(render_job *j, uint64_t chunk_i, uint64_t pass, ora_ctx *ctx)
{
	const uint64_t width = j->opts.width, height = j->opts.height;
	const uint64_t pixel_num = width * height;
	const uint64_t first = chunk_i * PIXEL_CHUNK_SIZE;
	const uint64_t last = first + PIXEL_CHUNK_SIZE < pixel_num ? first + PIXEL_CHUNK_SIZE : pixel_num;
	const uint64_t sample_index = j->opts.sample_begin + pass;
	/* sample_split (rt_hip.h): chunk c holds passes [floor(c*spp/S), floor((c+1)*spp/S)); with S = 1 the fold is the
	 * callback's running mean with `i as Float` = pass + 1 */
	const uint64_t spp = j->opts.samples_per_pixel, S = j->split;
	uint64_t chunk = pass * S / spp;
	while ((chunk + 1) * spp / S <= pass)
		chunk++;
	while (chunk * spp / S > pass)
		chunk--;
	const uint64_t chunk_begin = chunk * spp / S, chunk_end = (chunk + 1) * spp / S;
	const float i_f = (float)(pass - chunk_begin + 1);
	const bool chunk_starts = pass == chunk_begin, chunk_ends = pass + 1 == chunk_end;
	uint64_t rays_shot = 0;
	for (uint64_t pixel_i = first; pixel_i < last; ++pixel_i) {
		const uint64_t x = pixel_i % width;
		const uint64_t y = (pixel_i - x) / width;
		if (!pixel_owned(j, x, y))
			continue;
		rt_rng_seed(&ctx->rng, j->opts.seed, pixel_i, sample_index);
		ctx->c.rng_draws += 2;
		const float u = (rt_rng_range_f32(&ctx->rng, 0.0f, 1.0f) + (float)x) / (float)(width - 1);
		const float v = 1.0f - (rt_rng_range_f32(&ctx->rng, 0.0f, 1.0f) + (float)y) / (float)(height - 1);
		ora_ray ray = camera_get_ray(j->camera, u, v, ctx);
		uint64_t rc = 0;
		const vec3 rgb = j->opts.render_method == RT_METHOD_NAIVE
		                     ? ora_naive_get_colour(j->scene, &ray, j->opts.max_depth, j->opts.rr_threshold, &rc, ctx)
		                     : ora_mis_get_colour(j->scene, &ray, j->opts.max_depth, j->opts.rr_threshold, &rc, ctx);
		rays_shot += rc;
		float *pres = &j->mean[pixel_i * 3];
		if (S == 1) {
			/* src/main.rs:179-185: *pres += (acc - *pres) / i as Float */
			pres[0] += (rgb.x - pres[0]) / i_f;
			pres[1] += (rgb.y - pres[1]) / i_f;
			pres[2] += (rgb.z - pres[2]) / i_f;
		} else {
			/* sample_split S > 1 (rt_hip.h): the passes of a chunk are SUMMED in pass order, the chunk sums are summed in chunk
			 * order and the total is divided by spp once */
			if (chunk_starts)
				pres[0] = pres[1] = pres[2] = 0.0f;
			pres[0] += rgb.x;
			pres[1] += rgb.y;
			pres[2] += rgb.z;
			if (chunk_ends) {
				float *a = &j->acc[pixel_i * 3];
				for (int k = 0; k < 3; ++k)
					a[k] = a[k] + pres[k];
				if (pass + 1 == spp)
					for (int k = 0; k < 3; ++k)
						pres[k] = a[k] / (float)spp;
			}
		}
	}
	atomic_fetch_add(&j->rays_shot, rays_shot);
}

typedef struct worker_arg {
	render_job *job;
	uint32_t tid;
} worker_arg;

static void *render_worker(void *p)
{
	worker_arg *a = (worker_arg *)p;
	render_job *j = a->job;
	ora_ctx ctx;
	ora_ctx_init(&ctx);
	for (uint64_t pass = 0; pass < j->opts.samples_per_pixel; ++pass) {
		for (;;) { /* rayon's par_chunks_mut: chunks handed to whichever worker is free */
			const uint64_t c = atomic_fetch_add(&j->next_chunk, 1);
			if (c >= j->n_chunks)
				break;
			render_chunk(j, c, pass, &ctx);
		}
		/* one barrier per pass (random_sampler.rs:34-81) */
		if (pthread_barrier_wait(&j->barrier) == PTHREAD_BARRIER_SERIAL_THREAD)
			atomic_store(&j->next_chunk, 0);
		pthread_barrier_wait(&j->barrier);
	}
	j->thread_counters[a->tid] = ctx.c;
	ora_ctx_free(&ctx);
	return NULL;
}

int ora_render(const ora_scene *scene, const rt_camera *camera, const rt_render_opts *opts, float *out_rgb,
               uint64_t *rays_shot, uint32_t n_threads, ora_counters *counters)
{
	if (!scene || !camera || !opts || !out_rgb)
		return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
	if (opts->width < 2 || opts->height < 2)
		return fail(RT_ERR_INVALID_ARGUMENT, "width and height must be >= 2 (u,v divide by W-1, H-1)");
	if (opts->output_layout != RT_LAYOUT_FRAME)
		return fail(RT_ERR_UNSUPPORTED, "the oracle renders FRAME layout only");
	if (opts->shard_count == 0 || opts->shard_index >= opts->shard_count)
		return fail(RT_ERR_INVALID_ARGUMENT, "bad shard");
	if (n_threads == 0)
		n_threads = 1;

	render_job job;
	memset(&job, 0, sizeof job);
	job.scene = scene;
	job.camera = camera;
	job.opts = *opts;
	job.mean = out_rgb;
	if (opts->sample_split == 0) /* 0 = "the device library picks" (rt_hip.h): a rule about GPUs; the checker takes the split it is told */
		return fail(RT_ERR_INVALID_ARGUMENT, "sample_split 0 (automatic) is resolved by the device library (rt_scene_auto_sample_split): pass the split it chose");
	job.split = opts->sample_split;
	if (job.split > opts->samples_per_pixel)
		return fail(RT_ERR_INVALID_ARGUMENT, "sample_split larger than samples_per_pixel");
	job.acc = job.split > 1 ? (float *)calloc(opts->width * opts->height * 3, sizeof(float)) : NULL;
	job.tile_w = opts->tile_width ? opts->tile_width : 8;
	job.tile_h = opts->tile_height ? opts->tile_height : 8;
	const uint64_t pixel_num = opts->width * opts->height;
	memset(out_rgb, 0, pixel_num * 3 * sizeof(float));
	job.n_chunks = (pixel_num + PIXEL_CHUNK_SIZE - 1) / PIXEL_CHUNK_SIZE;
	atomic_init(&job.next_chunk, 0);
	atomic_init(&job.rays_shot, 0);
	job.n_threads = n_threads;
	job.thread_counters = (ora_counters_i *)calloc(n_threads, sizeof(ora_counters_i));
	pthread_barrier_init(&job.barrier, NULL, n_threads);

	pthread_t *threads = (pthread_t *)malloc(n_threads * sizeof(pthread_t));
	worker_arg *args = (worker_arg *)malloc(n_threads * sizeof(worker_arg));
	for (uint32_t t = 0; t < n_threads; ++t) {
		args[t].job = &job;
		args[t].tid = t;
		if (t + 1 < n_threads)
			pthread_create(&threads[t], NULL, render_worker, &args[t]);
	}
	render_worker(&args[n_threads - 1]); /* the calling thread works too */
	for (uint32_t t = 0; t + 1 < n_threads; ++t)
		pthread_join(threads[t], NULL);

	if (rays_shot)
		*rays_shot = atomic_load(&job.rays_shot);
	if (counters) {
		memset(counters, 0, sizeof *counters);
		for (uint32_t t = 0; t < n_threads; ++t) {
			counters->rays += job.thread_counters[t].rays;
			counters->node_tests += job.thread_counters[t].node_tests;
			counters->sphere_tests += job.thread_counters[t].sphere_tests;
			counters->triangle_tests += job.thread_counters[t].triangle_tests;
			counters->closest_hits += job.thread_counters[t].closest_hits;
			counters->sky_ops += job.thread_counters[t].sky_ops;
			counters->rng_draws += job.thread_counters[t].rng_draws;
		}
	}
	pthread_barrier_destroy(&job.barrier);
	free(job.acc);
	free(job.thread_counters);
	free(threads);
	free(args);
	return RT_OK;
}

/* ---------------- Integrator::get_colour on one fixed ray (furnace-style tests) ---------------- */
typedef struct integ_job {
	const ora_scene *scene;
	rt_ray_desc ray;
	int32_t method;
	uint32_t max_depth, rr_threshold;
	uint64_t seed, begin, end;
	double sum[3];
} integ_job;

static void *integ_worker(void *p)
{
	integ_job *j = (integ_job *)p;
	ora_ctx ctx;
	ora_ctx_init(&ctx);
	j->sum[0] = j->sum[1] = j->sum[2] = 0.0;
	for (uint64_t k = j->begin; k < j->end; ++k) {
		rt_rng_seed(&ctx.rng, j->seed, 0, k);
		ora_ray ray = ora_ray_new(v3_from(j->ray.origin), v3_from(j->ray.direction), 0.0f);
		uint64_t rc;
		const vec3 c = j->method == RT_METHOD_NAIVE
		                   ? ora_naive_get_colour(j->scene, &ray, j->max_depth, j->rr_threshold, &rc, &ctx)
		                   : ora_mis_get_colour(j->scene, &ray, j->max_depth, j->rr_threshold, &rc, &ctx);
		j->sum[0] += c.x;
		j->sum[1] += c.y;
		j->sum[2] += c.z;
	}
	ora_ctx_free(&ctx);
	return NULL;
}

int ora_integrate_ray(const ora_scene *scene, const rt_ray_desc *ray, int32_t method, uint32_t max_depth,
                      uint32_t rr_threshold, uint64_t seed, uint64_t n_samples, uint32_t n_threads, double out_mean[3])
{
	if (!scene || !ray || !out_mean || n_samples == 0)
		return fail(RT_ERR_INVALID_ARGUMENT, "bad arguments");
	if (n_threads == 0)
		n_threads = 1;
	integ_job *jobs = (integ_job *)calloc(n_threads, sizeof *jobs);
	pthread_t *threads = (pthread_t *)malloc(n_threads * sizeof(pthread_t));
	for (uint32_t t = 0; t < n_threads; ++t) {
		jobs[t].scene = scene;
		jobs[t].ray = *ray;
		jobs[t].method = method;
		jobs[t].max_depth = max_depth;
		jobs[t].rr_threshold = rr_threshold;
		jobs[t].seed = seed;
		jobs[t].begin = n_samples * t / n_threads;
		jobs[t].end = n_samples * (t + 1) / n_threads;
		pthread_create(&threads[t], NULL, integ_worker, &jobs[t]);
	}
	out_mean[0] = out_mean[1] = out_mean[2] = 0.0;
	for (uint32_t t = 0; t < n_threads; ++t) {
		pthread_join(threads[t], NULL);
		for (int k = 0; k < 3; ++k)
			out_mean[k] += jobs[t].sum[k];
	}
	for (int k = 0; k < 3; ++k)
		out_mean[k] /= (double)n_samples;
	free(jobs);
	free(threads);
	return RT_OK;
}

/* ---------------- hooks for the restated statistical / unit tests ---------------- */
int ora_detmath_eval(int32_t which, const float *a, const float *b, uint64_t n, float *out)
{
	for (uint64_t i = 0; i < n; ++i) {
		switch (which) {
		case 0: out[i] = rt_sinf(a[i]); break;
		case 1: out[i] = rt_cosf(a[i]); break;
		case 2: out[i] = rt_acosf(a[i]); break;
		case 3: out[i] = rt_atan2f(a[i], b[i]); break;
		case 4: out[i] = rt_tanf(a[i]); break;
		case 5: out[i] = rt_pow5f(a[i]); break;
		case 6: out[i] = rt_powf(a[i], b[i]); break;
		default: return fail(RT_ERR_INVALID_ARGUMENT, "unknown function");
		}
	}
	return RT_OK;
}
int ora_rng_fill_f32(uint64_t seed, uint64_t pixel, uint64_t sample, uint64_t n, float *out)
{
	rt_rng r;
	rt_rng_seed(&r, seed, pixel, sample);
	for (uint64_t i = 0; i < n; ++i)
		out[i] = rt_rng_f32(&r);
	return RT_OK;
}
int ora_rng_fill_u32(uint64_t seed, uint64_t pixel, uint64_t sample, uint64_t n, uint32_t *out)
{
	rt_rng r;
	rt_rng_seed(&r, seed, pixel, sample);
	for (uint64_t i = 0; i < n; ++i)
		out[i] = rt_rng_u32(&r);
	return RT_OK;
}
int ora_rng_fill_below(uint64_t seed, uint32_t bound, uint64_t n, uint32_t *out)
{
	rt_rng r;
	rt_rng_seed(&r, seed, 0, 0);
	for (uint64_t i = 0; i < n; ++i)
		out[i] = rt_rng_below(&r, bound);
	return RT_OK;
}

int ora_sample_directions(const ora_scene *scene, int32_t which, const float incoming_[3], const float normal_[3],
                          float alpha, uint64_t prim_index, uint64_t seed, uint64_t n, float *out)
{
	ora_ctx ctx;
	ora_ctx_init(&ctx);
	const vec3 incoming = incoming_ ? v3_from(incoming_) : v3_zero();
	const vec3 normal = normal_ ? v3_from(normal_) : v3(0.0f, 0.0f, 1.0f);
	for (uint64_t i = 0; i < n; ++i) {
		rt_rng_seed(&ctx.rng, seed, 0, i);
		vec3 d;
		switch (which) {
		case 0: d = ora_lambertian_sample(incoming, normal, &ctx); break;
		case 1: d = ora_tr_vndf_sample(alpha, incoming, normal, &ctx); break;
		case 2:
			if (!scene || !scene->sky.has_distribution) {
				ora_ctx_free(&ctx);
				return fail(RT_ERR_INVALID_ARGUMENT, "sky is not samplable");
			}
			d = ora_sky_sample(scene, &ctx);
			break;
		case 3: d = ora_random_unit_vector(&ctx.rng); break;
		case 4:
			if (!scene || prim_index >= scene->n_primitives) {
				ora_ctx_free(&ctx);
				return fail(RT_ERR_INVALID_ARGUMENT, "bad primitive index");
			}
			d = ora_prim_sample_visible_from_point(scene, &scene->primitives[prim_index], incoming, &ctx);
			break;
		case 5: d = ora_tr_vndf_sample_h(alpha, incoming, &ctx); break; /* half vectors, local frame */
		default:
			ora_ctx_free(&ctx);
			return fail(RT_ERR_INVALID_ARGUMENT, "unknown sampler");
		}
		out[3 * i] = d.x;
		out[3 * i + 1] = d.y;
		out[3 * i + 2] = d.z;
	}
	ora_ctx_free(&ctx);
	return RT_OK;
}

int ora_eval_pdfs(const ora_scene *scene, int32_t which, const float incoming_[3], const float normal_[3], float alpha,
                  const float *dirs, uint64_t n, float *out)
{
	ora_ctx ctx;
	ora_ctx_init(&ctx);
	const vec3 incoming = incoming_ ? v3_from(incoming_) : v3_zero();
	const vec3 normal = normal_ ? v3_from(normal_) : v3(0.0f, 0.0f, 1.0f);
	for (uint64_t i = 0; i < n; ++i) {
		const vec3 d = v3_from(&dirs[3 * i]);
		switch (which) {
		case 0: out[i] = ora_lambertian_pdf(incoming, d, normal); break;
		case 1: out[i] = ora_tr_vndf_pdf(alpha, incoming, d, normal); break;
		case 2:
			if (!scene || !scene->sky.has_distribution) {
				ora_ctx_free(&ctx);
				return fail(RT_ERR_INVALID_ARGUMENT, "sky is not samplable");
			}
			out[i] = ora_sky_pdf(scene, d, &ctx);
			break;
		case 5: out[i] = ora_tr_vndf_h(alpha, d, incoming); break;
		default:
			ora_ctx_free(&ctx);
			return fail(RT_ERR_INVALID_ARGUMENT, "unknown pdf");
		}
	}
	ora_ctx_free(&ctx);
	return RT_OK;
}

int ora_dist1d_sample_many(const float *values, uint64_t n_values, uint64_t seed, uint64_t n, uint64_t *out_index,
                           float *out_pdf, float *out_cdf)
{
	if (!values || n_values == 0)
		return fail(RT_ERR_INVALID_ARGUMENT, "Empty pdf passed to Distribution1D"); /* distributions.rs:13-15 */
	/* Distribution1D::new  distributions.rs:12-44 */
	ora_dist1d d;
	d.n = n_values;
	d.cdf = (float *)malloc((n_values + 1) * sizeof(float));
	d.pdf = (float *)malloc(n_values * sizeof(float));
	d.cdf[0] = 0.0f;
	for (uint64_t i = 1; i <= n_values; ++i)
		d.cdf[i] = d.cdf[i - 1] + values[i - 1];
	const float c = d.cdf[n_values];
	if (c != 0.0f)
		for (uint64_t i = 0; i <= n_values; ++i)
			d.cdf[i] /= c;
	float last = 0.0f;
	for (uint64_t i = 0; i < n_values; ++i) {
		d.pdf[i] = d.cdf[i + 1] - last;
		last = d.cdf[i + 1];
	}
	ora_ctx ctx;
	ora_ctx_init(&ctx);
	rt_rng_seed(&ctx.rng, seed, 0, 0);
	for (uint64_t i = 0; i < n; ++i)
		out_index[i] = ora_dist1d_sample(&d, &ctx);
	if (out_pdf)
		memcpy(out_pdf, d.pdf, n_values * sizeof(float));
	if (out_cdf)
		memcpy(out_cdf, d.cdf, (n_values + 1) * sizeof(float));
	ora_ctx_free(&ctx);
	free(d.cdf);
	free(d.pdf);
	return RT_OK;
}

/* save_data_to_image's pixel conversion  crates/output/src/lib.rs:89-97:
 *   data.iter().map(|val| (val.powf(1.0 / gamma) * 255.999) as u8)
 * (powf = the contract's rt_powf; `as u8` saturates, NaN -> 0) */
int ora_output_rgb8(const float *rgb, uint64_t n_values, float gamma, uint8_t *out)
{
	if (!rgb || !out)
		return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
	const float inv_gamma = 1.0f / gamma;
	for (uint64_t i = 0; i < n_values; ++i)
		out[i] = rt_quantise_u8(rgb[i], inv_gamma);
	return RT_OK;
}

int ora_sort_by_indices(uint64_t *values, uint64_t n, const uint64_t *indices)
{
	uint64_t *idx = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
	memcpy(idx, indices, n * sizeof(uint64_t));
	ora_sort_by_indices_u64(values, n, idx);
	free(idx);
	return RT_OK;
}

int ora_sky_tables(const ora_scene *s, float *row_cdf, float *marginal_cdf)
{
	if (!s || !s->sky.has_distribution)
		return fail(RT_ERR_INVALID_ARGUMENT, "sky has no distribution");
	const ora_dist2d *d = &s->sky.distribution;
	for (uint64_t r = 0; r < d->dim_y; ++r)
		memcpy(row_cdf + r * (d->dim_x + 1), d->x_distributions[r].cdf, (d->dim_x + 1) * sizeof(float));
	memcpy(marginal_cdf, d->y_distribution.cdf, (d->dim_y + 1) * sizeof(float));
	return RT_OK;
}

int ora_utility_eval(int32_t which, const float *a, uint64_t n, float *out)
{
	for (uint64_t i = 0; i < n; ++i) {
		switch (which) {
		case 0: out[i] = ora_next_float(a[i]); break;
		case 1: out[i] = ora_previous_float(a[i]); break;
		case 2: out[i] = ora_gamma((uint32_t)a[i]); break;
		default: return fail(RT_ERR_INVALID_ARGUMENT, "unknown function");
		}
	}
	return RT_OK;
}
int ora_offset_ray(const float origin[3], const float normal[3], const float error[3], int32_t is_brdf, float out[3])
{
	const vec3 r = ora_offset_ray_v(v3_from(origin), v3_from(normal), v3_from(error), is_brdf != 0);
	out[0] = r.x;
	out[1] = r.y;
	out[2] = r.z;
	return RT_OK;
}

/* raw Philox4x32-10 block (for the Random123 known-answer vectors) */
int ora_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
	uint32_t c[4] = { ctr[0], ctr[1], ctr[2], ctr[3] };
	rt_philox4x32_10(c, key[0], key[1]);
	memcpy(out, c, sizeof c);
	return RT_OK;
}

/* Coordinate::new_from_z(z) then to_coord / inverse.to_coord (utility/coord.rs:9-31) */
int ora_coord_apply(const float z[3], const float v[3], int32_t inverse, float out[3])
{
	const ora_coord c = ora_coord_new_from_z(v3_from(z));
	const ora_coord inv = ora_coord_create_inverse(&c);
	const vec3 r = ora_coord_to_coord(inverse ? &inv : &c, v3_from(v));
	out[0] = r.x;
	out[1] = r.y;
	out[2] = r.z;
	return RT_OK;
}
