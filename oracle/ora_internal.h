/*
 * ora_internal.h -- shared declarations of the CPU oracle.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  oracle/ is a plain-C restatement of the reference's
 * (nonl4331/raytracing-rust) path-tracing hot path, used only by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg to CHECK the HIP back end.
 * Nothing under raytracing-rust_amd/ includes, links or calls it.
 *
 * PARITY STATUS: pixel-level parity with the Rust binary is "parity unpinned" -- the reference
 * cannot be built here (no Rust toolchain, nightly-only, no Cargo.lock), is unseeded, and ships
 * no golden images, hit records or seeded vectors.  What does pin the oracle is (a) the analytic targets of the reference's
 * (commented-out) integration tests -- furnace = 0.25, MIS == naive -- (b) restated
 * statistical sampler tests, (c) hand-derivable pixels (primary sky misses), and (d) the
 * reference's one deterministic unit test (sort_by_indices).  See DESIGN.md section 3.
 *
 * Arithmetic: f32 everywhere, operation order as written in the Rust source, compiled with
 * -ffp-contract=off; elementary functions and the random stream come from
 * include/rt_detmath.h (the boundary's arithmetic contract), not from libm / libc.
 */
#ifndef ORA_INTERNAL_H
#define ORA_INTERNAL_H

#include <stdint.h>
#include <stddef.h>
#include <stdbool.h>
#include <math.h>

#include "../include/rt_detmath.h"
#include "../include/rt_hip.h"
#include "rt_oracle.h"

/* ---- rt_core/src/lib.rs:23-40 ---- */
#define ORA_EPSILON 3.0e-4f                /* rt_core::EPSILON (f32 build) */
#define ORA_F32_EPSILON RT_F32_EPSILON     /* std::f32::EPSILON */
#define ORA_PI RT_PI
#define ORA_TAU RT_TAU
#define ORA_NO_INDEX UINT64_MAX            /* usize::MAX */

/* ---- Vec3 / Vec2: rt_core/src/vec.rs:108-121 ---- */
typedef struct vec3 { float x, y, z; } vec3;
typedef struct vec2 { float x, y; } vec2;

static inline vec3 v3(float x, float y, float z) { vec3 v = { x, y, z }; return v; }
static inline vec3 v3_zero(void) { return v3(0.0f, 0.0f, 0.0f); }
static inline vec3 v3_one(void) { return v3(1.0f, 1.0f, 1.0f); }
static inline vec3 v3_from(const float *p) { return v3(p[0], p[1], p[2]); }
/* impl_operator!(Add/Sub/Mul/Div)  vec.rs:10-29 : componentwise */
static inline vec3 v3_add(vec3 a, vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec3 v3_sub(vec3 a, vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline vec3 v3_mul(vec3 a, vec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline vec3 v3_div(vec3 a, vec3 b) { return v3(a.x / b.x, a.y / b.y, a.z / b.z); }
/* impl_operator_float!  vec.rs:53-88 : Vec3 op Float and Float op Vec3 */
static inline vec3 v3_muls(vec3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline vec3 v3_smul(float s, vec3 a) { return v3(s * a.x, s * a.y, s * a.z); }
static inline vec3 v3_divs(vec3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
static inline vec3 v3_ssub(float s, vec3 a) { return v3(s - a.x, s - a.y, s - a.z); }
static inline vec3 v3_sadd(float s, vec3 a) { return v3(s + a.x, s + a.y, s + a.z); }
static inline vec3 v3_sdiv(float s, vec3 a) { return v3(s / a.x, s / a.y, s / a.z); }
static inline vec3 v3_neg(vec3 a) { return v3(-a.x, -a.y, -a.z); }
/* vec.rs:165-168 */
static inline float v3_dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* vec.rs:170-177 */
static inline vec3 v3_cross(vec3 a, vec3 b)
{
	return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float v3_mag_sq(vec3 a) { return v3_dot(a, a); }
static inline float v3_mag(vec3 a) { return sqrtf(v3_dot(a, a)); }
static inline vec3 v3_normalised(vec3 a) { return v3_divs(a, v3_mag(a)); }
static inline vec3 v3_abs(vec3 a) { return v3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
/* f32::min / f32::max: a NaN operand is ignored (P-hazard 13).  Written out instead of fminf/fmaxf because
 * the sign of a +0/-0 tie is unspecified there (glibc returns the second argument, LLVM's x86 lowering of
 * llvm.minnum/maxnum -- what rustc emits for f32::min/max -- the first): `a.max(b)` keeps `a` on a tie. */
static inline float f_min(float a, float b) { return a != a ? b : (b < a ? b : a); }
static inline float f_max(float a, float b) { return a != a ? b : (b > a ? b : a); }
/* vec.rs:216-223 : x.max(y.max(z)) */
static inline float v3_component_max(vec3 a) { return f_max(a.x, f_max(a.y, a.z)); }
static inline vec3 v3_min_by_component(vec3 a, vec3 b) { return v3(f_min(a.x, b.x), f_min(a.y, b.y), f_min(a.z, b.z)); }
static inline vec3 v3_max_by_component(vec3 a, vec3 b) { return v3(f_max(a.x, b.x), f_max(a.y, b.y), f_max(a.z, b.z)); }
/* vec.rs:205-208 : 2.0 * self.dot(normal) * normal - *self */
static inline vec3 v3_reflected(vec3 self, vec3 normal)
{
	return v3_sub(v3_smul(2.0f * v3_dot(self, normal), normal), self);
}
/* vec.rs:241-247 : note is_finite uses || (P-hazard 1) */
static inline bool v3_contains_nan(vec3 a) { return isnan(a.x) || isnan(a.y) || isnan(a.z); }
static inline bool v3_is_finite(vec3 a) { return isfinite(a.x) || isfinite(a.y) || isfinite(a.z); }
static inline bool v3_eq(vec3 a, vec3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
/* vec.rs:155-163 */
static inline vec3 v3_from_spherical(float sin_theta, float cos_theta, float sin_phi, float cos_phi)
{
	return v3(sin_theta * cos_phi, sin_theta * sin_phi, cos_theta);
}

/* ---- Ray: rt_core/src/ray.rs:3-51 ---- */
typedef struct ora_ray {
	vec3 origin, direction, d_inverse, shear;
	float time;
} ora_ray;
ora_ray ora_ray_new(vec3 origin, vec3 direction, float time);
static inline vec3 ora_ray_at(const ora_ray *r, float t) { return v3_add(r->origin, v3_muls(r->direction, t)); }

/* ---- Hit / SurfaceIntersection: rt_core/src/primitive.rs:3-42 ---- */
typedef struct ora_hit {
	float t;
	vec3 point, error, normal;
	vec2 uv;
	bool has_uv;
	bool out;
} ora_hit;
typedef struct ora_si {
	ora_hit hit;
	uint32_t material;
} ora_si;

/* ---- utility/mod.rs ---- */
bool ora_check_side(vec3 *normal, vec3 ray_direction);       /* :6-13  */
vec3 ora_random_unit_vector(rt_rng *rng);                    /* :15-25 */
float ora_next_float(float f);                               /* :51-65 */
float ora_previous_float(float f);                           /* :67-81 */
float ora_gamma(uint32_t n);                                 /* :83-86 */
vec3 ora_offset_ray_v(vec3 origin, vec3 normal, vec3 error, bool is_brdf); /* :88-117 */
/* utility/coord.rs:3-31 */
typedef struct ora_coord { vec3 x, y, z; } ora_coord;
ora_coord ora_coord_new_from_z(vec3 z);
ora_coord ora_coord_create_inverse(const ora_coord *c);
vec3 ora_coord_to_coord(const ora_coord *c, vec3 v);

/* ---- acceleration/aabb.rs ---- */
typedef struct ora_aabb { vec3 min, max; } ora_aabb;

/* ---- scene objects (POD copies of the descriptors) ---- */
typedef struct ora_texture {
	int32_t type;
	vec3 colour_one, colour_two;
	vec3 *image;            /* ImageTexture.data */
	uint64_t dim_x, dim_y;  /* ImageTexture.dim = (width-1, height-1) */
	vec3 ran_vecs[256];
	uint32_t perm_x[256], perm_y[256], perm_z[256];
} ora_texture;

typedef struct ora_material {
	int32_t type;
	uint32_t texture;
	float param;
	vec3 ior;
	float metallic;
} ora_material;

typedef struct ora_mesh {
	vec3 *vertices;
	uint64_t n_vertices;
	vec3 *normals;
	uint64_t n_normals;
} ora_mesh;

typedef struct ora_primitive {
	int32_t type;
	uint32_t material;
	/* sphere */
	vec3 center;
	float radius;
	/* Triangle: inline; MeshTriangle: indices into mesh */
	vec3 points[3], normals[3];
	uint32_t mesh;
	uint32_t point_indices[3], normal_indices[3];
} ora_primitive;

/* statistics/distributions.rs:5-9,74-79 */
typedef struct ora_dist1d { float *pdf; float *cdf; uint64_t n; } ora_dist1d;
typedef struct ora_dist2d {
	ora_dist1d *x_distributions;
	ora_dist1d y_distribution;
	uint64_t dim_x, dim_y;
} ora_dist2d;

/* sky.rs:13-19 */
typedef struct ora_sky {
	uint32_t texture, material;
	bool has_distribution;
	ora_dist2d distribution;
	uint64_t res_x, res_y;
} ora_sky;

/* acceleration/mod.rs:331-336 */
typedef struct ora_node {
	ora_aabb bounds;
	bool has_children;
	uint64_t children[2];
	uint64_t primitive_offset, number_primitives;
} ora_node;

/* acceleration/mod.rs:22-27 */
typedef struct ora_prim_info {
	uint64_t index;
	vec3 min, max, center;
} ora_prim_info;

/* work counters under REFERENCE traversal semantics (every AABB-hit node visited, every
 * primitive of every hit leaf tested) -- the "algorithmic bytes" of SURVEY section 8(d) */
typedef struct ora_counters_i {
	uint64_t rays, node_tests, sphere_tests, triangle_tests, closest_hits, sky_ops, rng_draws;
} ora_counters_i;

/* acceleration/mod.rs:44-51 */
struct ora_scene {
	ora_texture *textures; uint32_t n_textures;
	ora_material *materials; uint32_t n_materials;
	ora_mesh *meshes; uint32_t n_meshes;
	ora_primitive *primitives; uint64_t n_primitives;
	uint64_t *primitive_order;  /* slot i holds desc primitive primitive_order[i] */
	ora_node *nodes; uint64_t n_nodes, cap_nodes;
	uint64_t *lights; uint64_t n_lights;
	ora_sky sky;
	int32_t split_type;
};

/* per-thread evaluation context: the random stream of the current path + counters */
typedef struct ora_ctx {
	rt_rng rng;
	ora_counters_i c;
	/* scratch for get_intersection_candidates (the Vec / VecDeque of acceleration/mod.rs:200-202) */
	uint64_t *queue; uint64_t queue_cap;
	uint64_t *cand_off, *cand_len; uint64_t cand_cap;
} ora_ctx;
void ora_ctx_init(ora_ctx *ctx);
void ora_ctx_free(ora_ctx *ctx);
static inline float ora_random_float(ora_ctx *ctx) { ctx->c.rng_draws++; return rt_rng_f32(&ctx->rng); } /* utility/mod.rs:41-44 */

/* ---- geometry (ora_geometry.c) ---- */
bool ora_aabb_does_int(const ora_aabb *b, const ora_ray *ray);                       /* aabb.rs:22-57 */
ora_aabb ora_prim_get_aabb(const ora_scene *s, const ora_primitive *p);              /* sphere.rs:175-182, triangle.rs:285-307 */
bool ora_prim_get_int(const ora_scene *s, const ora_primitive *p, const ora_ray *ray, ora_si *out, ora_ctx *ctx);
float ora_prim_area(const ora_scene *s, const ora_primitive *p);
vec3 ora_prim_sample_visible_from_point(const ora_scene *s, const ora_primitive *p, vec3 in_point, ora_ctx *ctx);
float ora_prim_scattering_pdf(const ora_scene *s, const ora_primitive *p, vec3 hit_point, vec3 wi, const ora_hit *sampled_hit);

/* ---- BVH (ora_bvh.c) ---- */
int ora_bvh_build(ora_scene *s);                                                      /* mod.rs:58-160 */
void ora_sort_by_indices_u64(uint64_t *vec, uint64_t n, uint64_t *indices);           /* utility/mod.rs:119-134 (exposed for its unit test) */
uint64_t ora_bvh_candidates(const ora_scene *s, const ora_ray *ray, ora_ctx *ctx);    /* mod.rs:199-224 */
uint64_t ora_bvh_check_hit(const ora_scene *s, const ora_ray *ray, ora_si *out, ora_ctx *ctx); /* mod.rs:265-298 */
bool ora_bvh_check_hit_index(const ora_scene *s, const ora_ray *ray, uint64_t index, ora_si *out, ora_ctx *ctx); /* mod.rs:226-263 */
float ora_bvh_get_pdf_from_index(const ora_scene *s, const ora_hit *last_hit, const ora_hit *light_hit, vec3 sampled_dir, uint64_t index, ora_ctx *ctx); /* mod.rs:299-318 */

/* ---- shading (ora_shading.c) ---- */
vec3 ora_texture_colour_value(const ora_scene *s, uint32_t tex, vec3 direction, vec3 point);   /* textures/mod.rs */
int ora_sky_build(ora_scene *s);                                                      /* sky.rs:22-39 */
void ora_sky_free(ora_scene *s);
bool ora_sky_can_sample(const ora_scene *s);                                          /* sky.rs:61-63 */
float ora_sky_pdf(const ora_scene *s, vec3 wi, ora_ctx *ctx);                         /* sky.rs:43-60 */
vec3 ora_sky_sample(const ora_scene *s, ora_ctx *ctx);                                /* sky.rs:64-78 */
ora_si ora_sky_get_si(const ora_scene *s);                                            /* sky.rs:79-92 */
uint64_t ora_dist1d_sample(const ora_dist1d *d, ora_ctx *ctx);                        /* distributions.rs:51-72 */
/* Scatter trait, enum-dispatched (rt_core/src/material.rs:4-30, proc/src/lib.rs:5-66) */
bool ora_mat_scatter_ray(const ora_scene *s, uint32_t mat, ora_ray *ray, const ora_hit *hit, ora_ctx *ctx);
bool ora_mat_is_light(const ora_scene *s, uint32_t mat);
bool ora_mat_is_delta(const ora_scene *s, uint32_t mat);
float ora_mat_scattering_pdf(const ora_scene *s, uint32_t mat, const ora_hit *hit, vec3 wo, vec3 wi);
vec3 ora_mat_eval(const ora_scene *s, uint32_t mat, const ora_hit *hit, vec3 wo, vec3 wi);
vec3 ora_mat_eval_over_scattering_pdf(const ora_scene *s, uint32_t mat, const ora_hit *hit, vec3 wo, vec3 wi);
vec3 ora_mat_get_emission(const ora_scene *s, uint32_t mat, const ora_hit *hit, vec3 wo);
/* bxdf sampling entry points used by the statistical tests */
vec3 ora_lambertian_sample(vec3 incoming, vec3 normal, ora_ctx *ctx);                 /* statistics/bxdfs/lambertian.rs:16-18 */
float ora_lambertian_pdf(vec3 incoming, vec3 outgoing, vec3 normal);                  /* :20-22 */
vec3 ora_tr_vndf_sample(float alpha, vec3 incoming, vec3 normal, ora_ctx *ctx);       /* trowbridge_reitz_vndf.rs:37-42 */
float ora_tr_vndf_pdf(float alpha, vec3 incoming, vec3 outgoing, vec3 normal);        /* :44-54 */
vec3 ora_tr_vndf_sample_h(float alpha, vec3 incoming, ora_ctx *ctx);                   /* :17-19 isotropic::sample_vndf */
float ora_tr_vndf_h(float alpha, vec3 h, vec3 incoming);                               /* :9-15  isotropic::vndf */

/* ---- integrators (ora_render.c) ---- */
vec3 ora_naive_get_colour(const ora_scene *s, ora_ray *ray, uint32_t max_depth, uint32_t rr_threshold, uint64_t *ray_count, ora_ctx *ctx); /* integrators/mod.rs:22-78 */
vec3 ora_mis_get_colour(const ora_scene *s, ora_ray *ray, uint32_t max_depth, uint32_t rr_threshold, uint64_t *ray_count, ora_ctx *ctx);   /* integrators/mis.rs:7-92 */

#endif
