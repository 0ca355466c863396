/*
 * ora_geometry.c -- CPU oracle: rays, utility helpers, AABB slab test, sphere and triangle
 * intersectors and their light-sampling geometry.  TEST INFRASTRUCTURE (see ora_internal.h).
 * Every function names the reference lines it restates (paths relative to
 * /root/reference/crates/).
 */
#include "ora_internal.h"

/* ---- rt_core/src/ray.rs:13-46  Ray::new ---- */
ora_ray ora_ray_new(vec3 origin, vec3 direction, float time)
{
	ora_ray r;
	direction = v3_divs(direction, v3_mag(direction)); /* direction.normalise() */

	int max_axis;
	if (fabsf(direction.x) > fabsf(direction.y) && fabsf(direction.x) > fabsf(direction.z))
		max_axis = 0;
	else if (fabsf(direction.y) > fabsf(direction.z))
		max_axis = 1;
	else
		max_axis = 2;

	vec3 swaped_dir = direction;
	if (max_axis == 0 || max_axis == 1) { /* both arms swap x<->z (ray.rs:26-33, P-hazard 4) */
		const float tmp = swaped_dir.x;
		swaped_dir.x = swaped_dir.z;
		swaped_dir.z = tmp;
	}
	const float shear_x = -swaped_dir.x / swaped_dir.z;
	const float shear_y = -swaped_dir.y / swaped_dir.z;
	const float shear_z = 1.0f / swaped_dir.z;

	r.origin = origin;
	r.direction = direction;
	r.d_inverse = v3(1.0f / direction.x, 1.0f / direction.y, 1.0f / direction.z);
	r.shear = v3(shear_x, shear_y, shear_z);
	r.time = time;
	return r;
}

/* ---- implementations/src/utility/mod.rs ---- */
bool ora_check_side(vec3 *normal, vec3 ray_direction) /* :6-13 */
{
	if (v3_dot(*normal, ray_direction) > 0.0f) {
		*normal = v3_neg(*normal);
		return false;
	}
	return true;
}

vec3 ora_random_unit_vector(rt_rng *rng) /* :15-25 */
{
	float x = 1.0f, y = 1.0f, z = 1.0f;
	while (x * x + y * y + z * z > 1.0f) {
		x = rt_rng_range_f32(rng, -1.0f, 1.0f);
		y = rt_rng_range_f32(rng, -1.0f, 1.0f);
		z = rt_rng_range_f32(rng, -1.0f, 1.0f);
	}
	return v3_normalised(v3(x, y, z));
}

float ora_next_float(float f) /* :51-65 */
{
	if (isinf(f) && f > 0.0f)
		return f;
	if (f == -0.0f)
		f = 0.0f;
	const uint32_t bits = rt_f32_bits(f);
	return rt_bits_f32(f >= 0.0f ? bits + 1u : bits - 1u);
}

float ora_previous_float(float f) /* :67-81 */
{
	if (isinf(f) && f < 0.0f)
		return f;
	if (f == 0.0f)
		f = -0.0f;
	const uint32_t bits = rt_f32_bits(f);
	return rt_bits_f32(f <= 0.0f ? bits + 1u : bits - 1u);
}

float ora_gamma(uint32_t n) /* :83-86 */
{
	const float nm = (float)n * 0.5f * ORA_F32_EPSILON;
	return nm / (1.0f - nm);
}

vec3 ora_offset_ray_v(vec3 origin, vec3 normal, vec3 error, bool is_brdf) /* :88-117 */
{
	const float offset_val = v3_dot(v3_abs(normal), error);
	vec3 offset = v3_smul(offset_val, normal);
	if (!is_brdf)
		offset = v3_neg(offset);
	vec3 new_origin = v3_add(origin, offset);
	new_origin.x = offset.x > 0.0f ? ora_next_float(new_origin.x) : ora_previous_float(new_origin.x);
	new_origin.y = offset.y > 0.0f ? ora_next_float(new_origin.y) : ora_previous_float(new_origin.y);
	new_origin.z = offset.z > 0.0f ? ora_next_float(new_origin.z) : ora_previous_float(new_origin.z);
	return new_origin;
}

/* ---- implementations/src/utility/coord.rs:9-31 ---- */
ora_coord ora_coord_new_from_z(vec3 z)
{
	ora_coord c;
	if (fabsf(z.x) > fabsf(z.y))
		c.x = v3_divs(v3(-z.z, 0.0f, z.x), sqrtf(z.x * z.x + z.z * z.z));
	else
		c.x = v3_divs(v3(0.0f, z.z, -z.y), sqrtf(z.y * z.y + z.z * z.z));
	c.y = v3_cross(c.x, z);
	c.z = z;
	return c;
}
ora_coord ora_coord_create_inverse(const ora_coord *c)
{
	ora_coord r;
	r.x = v3(c->x.x, c->y.x, c->z.x);
	r.y = v3(c->x.y, c->y.y, c->z.y);
	r.z = v3(c->x.z, c->y.z, c->z.z);
	return r;
}
vec3 ora_coord_to_coord(const ora_coord *c, vec3 v)
{
	return v3_add(v3_add(v3_smul(v.x, c->x), v3_smul(v.y, c->y)), v3_smul(v.z, c->z));
}

/* ---- implementations/src/acceleration/aabb.rs:22-57  AABB::does_int ---- */
bool ora_aabb_does_int(const ora_aabb *b, const ora_ray *ray)
{
	const float widen = 1.0f + 2.0f * ora_gamma(3);

	float t1 = (b->min.x - ray->origin.x) * ray->d_inverse.x;
	float t2 = (b->max.x - ray->origin.x) * ray->d_inverse.x;
	if (t1 > t2) { const float s = t1; t1 = t2; t2 = s; }
	t2 *= widen;
	float tmin = f_min(t1, t2);
	float tmax = f_max(t1, t2);

	t1 = (b->min.y - ray->origin.y) * ray->d_inverse.y;
	t2 = (b->max.y - ray->origin.y) * ray->d_inverse.y;
	if (t1 > t2) { const float s = t1; t1 = t2; t2 = s; }
	t2 *= widen;
	tmin = f_max(tmin, f_min(t1, t2));
	tmax = f_min(tmax, f_max(t1, t2));

	t1 = (b->min.z - ray->origin.z) * ray->d_inverse.z;
	t2 = (b->max.z - ray->origin.z) * ray->d_inverse.z;
	if (t1 > t2) { const float s = t1; t1 = t2; t2 = s; }
	t2 *= widen;
	tmin = f_max(tmin, f_min(t1, t2));
	tmax = f_min(tmax, f_max(t1, t2));

	return tmax > f_max(tmin, 0.0f);
}

/* ---- triangle vertex access: TriangleTrait  primitives/triangle.rs:69-103 ---- */
static inline vec3 tri_point(const ora_scene *s, const ora_primitive *p, int i)
{
	if (p->type == RT_PRIM_MESH_TRIANGLE)
		return s->meshes[p->mesh].vertices[p->point_indices[i]];
	return p->points[i];
}
static inline vec3 tri_normal(const ora_scene *s, const ora_primitive *p, int i)
{
	if (p->type == RT_PRIM_MESH_TRIANGLE)
		return s->meshes[p->mesh].normals[p->normal_indices[i]];
	return p->normals[i];
}

/* ---- AABound: sphere.rs:175-182, triangle.rs:285-307 ---- */
ora_aabb ora_prim_get_aabb(const ora_scene *s, const ora_primitive *p)
{
	ora_aabb b;
	if (p->type == RT_PRIM_SPHERE) {
		b.min = v3_sub(p->center, v3_smul(p->radius, v3_one()));
		b.max = v3_add(p->center, v3_smul(p->radius, v3_one()));
	} else {
		const vec3 p0 = tri_point(s, p, 0), p1 = tri_point(s, p, 1), p2 = tri_point(s, p, 2);
		b.min = v3_min_by_component(p0, v3_min_by_component(p1, p2));
		b.max = v3_max_by_component(p0, v3_max_by_component(p1, p2));
	}
	return b;
}

/* ---- primitives/sphere.rs:34-105  Sphere::get_int ---- */
static bool sphere_get_int(const ora_primitive *p, const ora_ray *ray, ora_si *out)
{
	const vec3 dir = ray->direction;
	const vec3 center = p->center;
	const float radius = p->radius;
	const vec3 orig = ray->origin;

	const vec3 deltap = v3_sub(center, orig);
	const float ddp = v3_dot(dir, deltap);
	const float deltapdot = v3_dot(deltap, deltap);

	const vec3 remedy_term = v3_sub(deltap, v3_smul(ddp, dir));
	const float discriminant = radius * radius - v3_dot(remedy_term, remedy_term);

	if (!(discriminant > 0.0f))
		return false;

	const float sqrt_val = sqrtf(discriminant);
	const float q = ddp > 0.0f ? ddp + sqrt_val : ddp - sqrt_val;

	float t0 = q;
	float t1 = (deltapdot - radius * radius) / q;
	if (t1 < t0) { const float s = t0; t0 = t1; t1 = s; }

	float t;
	if (t0 > 0.0f) {
		t = t0;
	} else {
		if (t1 <= 0.0f)
			return false;
		t = t1;
	}

	const vec3 point = ora_ray_at(ray, t);
	vec3 normal = v3_divs(v3_sub(point, center), radius);
	bool is_out = true;
	if (v3_dot(normal, dir) > 0.0f) {
		is_out = false;
		normal = v3_neg(normal);
	}

	out->hit.t = t;
	out->hit.point = point;
	out->hit.error = v3_smul(ORA_EPSILON, v3_one());
	out->hit.normal = normal;
	/* get_uv (sphere.rs:106-117) only runs when material.requires_uv(); no material overrides
	 * the trait default `false` (rt_core/src/material.rs:8-10), so uv is always None. */
	out->hit.has_uv = false;
	out->hit.uv.x = 0.0f;
	out->hit.uv.y = 0.0f;
	out->hit.out = is_out;
	out->material = p->material;
	return true;
}

/* ---- primitives/triangle.rs:105-216  triangle_intersection ---- */
static inline void swap_z(vec3 *v, int axis) /* primitives/mod.rs:72-82 : X and Y both swap x<->z */
{
	if (axis == 0 || axis == 1) {
		const float t = v->x;
		v->x = v->z;
		v->z = t;
	}
}
static inline int max_abs_axis(vec3 v) /* primitives/mod.rs:62-70 */
{
	if (fabsf(v.x) > fabsf(v.y) && fabsf(v.x) > fabsf(v.z))
		return 0;
	if (fabsf(v.y) > fabsf(v.z))
		return 1;
	return 2;
}

static bool triangle_intersection(const ora_scene *s, const ora_primitive *p, const ora_ray *ray, ora_si *out)
{
	const vec3 P0 = tri_point(s, p, 0), P1 = tri_point(s, p, 1), P2 = tri_point(s, p, 2);
	vec3 p0t = v3_sub(P0, ray->origin);
	vec3 p1t = v3_sub(P1, ray->origin);
	vec3 p2t = v3_sub(P2, ray->origin);

	const int axis = max_abs_axis(ray->direction);
	swap_z(&p0t, axis);
	swap_z(&p1t, axis);
	swap_z(&p2t, axis);

	p0t.x += ray->shear.x * p0t.z;
	p0t.y += ray->shear.y * p0t.z;
	p1t.x += ray->shear.x * p1t.z;
	p1t.y += ray->shear.y * p1t.z;
	p2t.x += ray->shear.x * p2t.z;
	p2t.y += ray->shear.y * p2t.z;

	float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
	float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
	float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
	if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) { /* :128-132 f64 recompute */
		e0 = (float)((double)p1t.x * (double)p2t.y - (double)p1t.y * (double)p2t.x);
		e1 = (float)((double)p2t.x * (double)p0t.y - (double)p2t.y * (double)p0t.x);
		e2 = (float)((double)p0t.x * (double)p1t.y - (double)p0t.y * (double)p1t.x);
	}

	if ((e0 < 0.0f || e1 < 0.0f || e2 < 0.0f) && (e0 > 0.0f || e1 > 0.0f || e2 > 0.0f))
		return false;

	const float det = e0 + e1 + e2;
	if (det == 0.0f)
		return false;

	p0t = v3_muls(p0t, ray->shear.z);
	p1t = v3_muls(p1t, ray->shear.z);
	p2t = v3_muls(p2t, ray->shear.z);

	const float t_scaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
	if ((det < 0.0f && t_scaled >= 0.0f) || (det > 0.0f && t_scaled <= 0.0f))
		return false;

	const float inv_det = 1.0f / det;
	const float b0 = e0 * inv_det;
	const float b1 = e1 * inv_det;
	const float b2 = e2 * inv_det;
	const float t = inv_det * t_scaled;

	const float max_z_t = v3_component_max(v3(fabsf(p0t.z), fabsf(p1t.z), fabsf(p2t.z)));
	const float delta_z = ora_gamma(3) * max_z_t;

	const float max_x_t = v3_component_max(v3(fabsf(p0t.x), fabsf(p1t.x), fabsf(p2t.x)));
	const float max_y_t = v3_component_max(v3(fabsf(p0t.y), fabsf(p1t.y), fabsf(p2t.y)));
	const float delta_x = ora_gamma(5) * (max_x_t + max_z_t);
	const float delta_y = ora_gamma(5) * (max_y_t + max_z_t);

	const float delta_e = 2.0f * (ora_gamma(2) * max_x_t * max_y_t + delta_y * max_x_t + delta_x * max_y_t);
	const float max_e = v3_component_max(v3(fabsf(e0), fabsf(e1), fabsf(e2)));
	const float delta_t =
	    3.0f * (ora_gamma(3) * max_e * max_z_t + delta_e * max_z_t + delta_z * max_e) * fabsf(inv_det);

	if (t < delta_t)
		return false;

	/* uv = b0*(0,0) + b1*(1,0) + b2*(1,1) */
	vec2 uv;
	uv.x = b0 * 0.0f + b1 * 1.0f + b2 * 1.0f;
	uv.y = b0 * 0.0f + b1 * 0.0f + b2 * 1.0f;

	const vec3 N0 = tri_normal(s, p, 0), N1 = tri_normal(s, p, 1), N2 = tri_normal(s, p, 2);
	vec3 normal = v3_add(v3_add(v3_smul(b0, N0), v3_smul(b1, N1)), v3_smul(b2, N2));
	const bool is_out = ora_check_side(&normal, ray->direction);

	const float x_abs_sum = fabsf(b0 * P0.x) + fabsf(b1 * P1.x) + fabsf(b2 * P2.x);
	const float y_abs_sum = fabsf(b0 * P0.y) + fabsf(b1 * P1.y) + fabsf(b2 * P2.y);
	const float z_abs_sum = fabsf(b0 * P0.z) + fabsf(b1 * P1.z) + fabsf(b2 * P2.z);

	const vec3 point_error = v3_add(v3_smul(ora_gamma(7), v3(x_abs_sum, y_abs_sum, z_abs_sum)),
	                                v3_smul(ora_gamma(6), v3(b2 * P2.x, b2 * P2.y, b2 * P2.z)));

	const vec3 point = v3_add(v3_add(v3_smul(b0, P0), v3_smul(b1, P1)), v3_smul(b2, P2));

	out->hit.t = t;
	out->hit.point = point;
	out->hit.error = point_error;
	out->hit.normal = normal;
	out->hit.uv = uv;
	out->hit.has_uv = true;
	out->hit.out = is_out;
	out->material = p->material;
	return true;
}

/* enum dispatch: #[derive(Primitive)]  proc/src/lib.rs:126-187 */
bool ora_prim_get_int(const ora_scene *s, const ora_primitive *p, const ora_ray *ray, ora_si *out, ora_ctx *ctx)
{
	if (p->type == RT_PRIM_SPHERE) {
		ctx->c.sphere_tests++;
		return sphere_get_int(p, ray, out);
	}
	ctx->c.triangle_tests++;
	return triangle_intersection(s, p, ray, out);
}

/* ---- area: sphere.rs:167-169, triangle.rs:226-230,253-262 ---- */
float ora_prim_area(const ora_scene *s, const ora_primitive *p)
{
	if (p->type == RT_PRIM_SPHERE)
		return 4.0f * ORA_PI * p->radius * p->radius;
	const vec3 p0 = tri_point(s, p, 0), p1 = tri_point(s, p, 1), p2 = tri_point(s, p, 2);
	return 0.5f * v3_mag(v3_cross(v3_sub(p1, p0), v3_sub(p2, p0)));
}

/* ---- sphere.rs:118-123  Sphere::get_sample ---- */
static vec3 sphere_get_sample(const ora_primitive *p, ora_ctx *ctx)
{
	const float z = 1.0f - 2.0f * ora_random_float(ctx);
	const float a = sqrtf(f_max(1.0f - z * z, 0.0f));
	const float b = 2.0f * ORA_PI * ora_random_float(ctx);
	return v3_add(p->center, v3_smul(p->radius, v3(a * rt_cosf(b), a * rt_sinf(b), z)));
}

/* ---- sample_visible_from_point: sphere.rs:124-154, triangle.rs:231-241,263-277 ---- */
vec3 ora_prim_sample_visible_from_point(const ora_scene *s, const ora_primitive *p, vec3 in_point, ora_ctx *ctx)
{
	if (p->type == RT_PRIM_SPHERE) {
		const float radius = p->radius;
		const float distance_sq = v3_mag_sq(v3_sub(in_point, p->center));
		vec3 point;
		if (distance_sq <= radius * radius) {
			point = sphere_get_sample(p, ctx);
		} else {
			const float distance = sqrtf(distance_sq);
			const float sin_theta_max_sq = radius * radius / distance_sq;
			const float cost_theta_max = sqrtf(f_max(1.0f - sin_theta_max_sq, 0.0f));
			const float r1 = ora_random_float(ctx);
			const float cos_theta = (1.0f - r1) + r1 * cost_theta_max;
			const float sin_theta = sqrtf(f_max(1.0f - cos_theta * cos_theta, 0.0f));
			const float phi = 2.0f * ora_random_float(ctx) * ORA_PI;

			const float ds = distance * cos_theta -
			                 sqrtf(f_max(radius * radius - distance_sq * sin_theta * sin_theta, 0.0f));
			const float cos_alpha = (distance_sq + radius * radius - ds * ds) / (2.0f * distance * radius);
			const float sin_alpha = sqrtf(f_max(1.0f - cos_alpha * cos_alpha, 0.0f));

			const ora_coord coord_system = ora_coord_new_from_z(v3_normalised(v3_sub(in_point, p->center)));
			vec3 vec = v3(sin_alpha * rt_cosf(phi), sin_alpha * rt_sinf(phi), cos_alpha);
			vec = ora_coord_to_coord(&coord_system, vec);
			point = v3_add(p->center, v3_smul(radius, vec));
		}
		return v3_normalised(v3_sub(point, in_point));
	}
	/* Triangle: uv = (1-sqrt(r1), sqrt(r1)*r2); MeshTriangle: (1-sqrt(r1), sqrt(r1)*sqrt(r2)) (P-hazard 6) */
	const float su = sqrtf(ora_random_float(ctx));
	const float u0 = 1.0f - su;
	float r2 = ora_random_float(ctx);
	if (p->type == RT_PRIM_MESH_TRIANGLE)
		r2 = sqrtf(r2);
	const float u1 = su * r2;
	const vec3 p0 = tri_point(s, p, 0), p1 = tri_point(s, p, 1), p2 = tri_point(s, p, 2);
	const vec3 point = v3_add(v3_add(v3_smul(u0, p0), v3_smul(u1, p1)), v3_smul(1.0f - u0 - u1, p2));
	return v3_normalised(v3_sub(point, in_point));
}

/* ---- scattering_pdf: sphere.rs:155-166, triangle.rs:242-244,278-280 ---- */
float ora_prim_scattering_pdf(const ora_scene *s, const ora_primitive *p, vec3 hit_point, vec3 wi,
                              const ora_hit *sampled_hit)
{
	if (p->type == RT_PRIM_SPHERE) {
		const float rsq = p->radius * p->radius;
		const float dsq = v3_mag_sq(v3_sub(hit_point, p->center));
		if (dsq <= rsq)
			return v3_mag_sq(v3_sub(sampled_hit->point, hit_point)) /
			       (fabsf(v3_dot(wi, sampled_hit->normal)) * ora_prim_area(s, p));
		const float sin_theta_max_sq = rsq / dsq;
		const float cos_theta_max = sqrtf(f_max(1.0f - sin_theta_max_sq, 0.0f));
		return 1.0f / (2.0f * ORA_PI * (1.0f - cos_theta_max));
	}
	return v3_mag_sq(v3_sub(sampled_hit->point, hit_point)) /
	       (fabsf(v3_dot(sampled_hit->normal, wi)) * ora_prim_area(s, p));
}
