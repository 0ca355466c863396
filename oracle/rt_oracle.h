/*
 * rt_oracle.h -- public entry points of the CPU oracle (liboracle.so).
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library.  It restates the reference's algorithm on the CPU
 * (see ora_internal.h for the parity status) and consumes the same POD descriptors as the
 * product ABI (include/rt_hip.h) so that one scene description drives both sides.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>
#include "../include/rt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ora_scene ora_scene;

/* work counted under reference traversal semantics, summed over all threads */
typedef struct ora_counters {
	uint64_t rays;            /* Bvh::check_hit + check_hit_index calls */
	uint64_t node_tests;      /* AABB::does_int calls */
	uint64_t sphere_tests;    /* Sphere::get_int calls */
	uint64_t triangle_tests;  /* triangle_intersection calls */
	uint64_t closest_hits;    /* check_hit calls that returned a primitive */
	uint64_t sky_ops;         /* Sky::sample + Sky::pdf calls */
	uint64_t rng_draws;       /* random numbers consumed */
} ora_counters;

const char *ora_last_error(void);

int ora_camera_new(rt_camera *out, const float origin[3], const float lookat[3], const float vup[3],
                   float fov_degrees, float aspect_ratio, float aperture, float focus_dist);

int ora_scene_create(const rt_scene_desc *desc, ora_scene **out);
void ora_scene_destroy(ora_scene *scene);
int ora_scene_counts(const ora_scene *scene, uint64_t *n_nodes, uint64_t *n_primitives, uint64_t *n_lights);
int ora_scene_get_nodes(const ora_scene *scene, rt_bvh_node *out, uint64_t capacity);
int ora_scene_get_primitive_order(const ora_scene *scene, uint64_t *out, uint64_t capacity);
int ora_scene_get_lights(const ora_scene *scene, uint64_t *out, uint64_t capacity);

/* RandomSampler::sample_image + the TUI callback's running mean, n_threads workers over
 * 10 000-pixel chunks with a barrier per pass (samplers/random_sampler.rs:31-81,
 * src/main.rs:175-191).  FRAME layout only; honours shard_index/shard_count (unowned
 * pixels are left at 0).  counters may be NULL. */
int ora_render(const ora_scene *scene, const rt_camera *camera, const rt_render_opts *opts, float *out_rgb,
               uint64_t *rays_shot, uint32_t n_threads, ora_counters *counters);

int ora_check_hit(const ora_scene *scene, const rt_ray_desc *rays, uint64_t n_rays, rt_hit_record *out);
int ora_check_hit_index(const ora_scene *scene, const rt_ray_desc *rays, const uint64_t *object_index,
                        uint64_t n_rays, rt_hit_record *out);

/* Integrator::get_colour on ONE fixed ray, n_samples times (sample k uses the stream
 * (seed, pixel=0, sample=k)); returns the arithmetic mean in f64.  This is the shape of the
 * reference's furnace / MIS-vs-naive tests (crates/implementations/tests/sampling.rs:239-297). */
int ora_integrate_ray(const ora_scene *scene, const rt_ray_desc *ray, int32_t render_method, uint32_t max_depth,
                      uint32_t rr_threshold, uint64_t seed, uint64_t n_samples, uint32_t n_threads, double out_mean[3]);

/* ---- hooks for the restated statistical tests ---- */
/* elementwise rt_detmath functions: which = 0 sin, 1 cos, 2 acos, 3 atan2(a,b), 4 tan, 5 pow5 */
int ora_detmath_eval(int32_t which, const float *a, const float *b, uint64_t n, float *out);
int ora_rng_fill_f32(uint64_t seed, uint64_t pixel, uint64_t sample, uint64_t n, float *out);
int ora_rng_fill_u32(uint64_t seed, uint64_t pixel, uint64_t sample, uint64_t n, uint32_t *out);
int ora_rng_fill_below(uint64_t seed, uint32_t bound, uint64_t n, uint32_t *out);
/* directions sampled by: 0 lambertian(incoming,normal), 1 trowbridge-reitz VNDF(alpha),
 * 2 Sky::sample, 3 random_unit_vector, 4 primitive.sample_visible_from_point(prim index, point=incoming) */
int ora_sample_directions(const ora_scene *scene, int32_t which, const float incoming[3], const float normal[3],
                          float alpha, uint64_t prim_index, uint64_t seed, uint64_t n, float *out_xyz);
/* pdfs of those samplers evaluated at given directions (which as above; 2 = Sky::pdf) */
int ora_eval_pdfs(const ora_scene *scene, int32_t which, const float incoming[3], const float normal[3],
                  float alpha, const float *dirs_xyz, uint64_t n, float *out);
/* Distribution1D::new + sample (statistics/distributions.rs:12-72): samples n indices */
int ora_dist1d_sample_many(const float *values, uint64_t n_values, uint64_t seed, uint64_t n, uint64_t *out_index,
                           float *out_pdf /* n_values */, float *out_cdf /* n_values+1 */);
/* utility::sort_by_indices (utility/mod.rs:119-134) on an array of u64 */
/* GGX terms over arrays and Distribution2D sampling: hooks for the restated reference tests
 * (statistics/bxdfs/trowbridge_reitz.rs:128-230, statistics/distributions.rs:206-300) */
int ora_tr_d_many(float alpha, const float *cos_theta, uint64_t n, float *out);
int ora_tr_g1_many(float alpha, const float normal[3], const float *h, const float v[3], uint64_t n, float *out);
int ora_tr_g2_many(float alpha, const float normal[3], const float *h, const float incoming[3], const float *outgoing, uint64_t n,
                   float *out);
int ora_dist2d_sample_many(const float *values, uint64_t n_values, uint64_t width, uint64_t seed, uint64_t n, uint32_t *out_x,
                           uint32_t *out_y, float *out_pdf);
/* the output stage's pixel conversion (crates/output/src/lib.rs:89-97) */
int ora_output_rgb8(const float *rgb, uint64_t n_values, float gamma, uint8_t *out);
int ora_sort_by_indices(uint64_t *values, uint64_t n, const uint64_t *indices);
/* sky tables as built by Sky::new: cdf rows (res_y x (res_x+1)) then marginal cdf (res_y+1) */
int ora_sky_tables(const ora_scene *scene, float *row_cdf, float *marginal_cdf);
/* utility helpers, elementwise: which = 0 next_float, 1 previous_float, 2 gamma(n = (uint)a) */
int ora_utility_eval(int32_t which, const float *a, uint64_t n, float *out);
int ora_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
int ora_coord_apply(const float z[3], const float v[3], int32_t inverse, float out[3]);
int ora_offset_ray(const float origin[3], const float normal[3], const float error[3], int32_t is_brdf, float out[3]);

#ifdef __cplusplus
}
#endif
#endif
