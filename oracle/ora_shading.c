/*
 * ora_shading.c -- CPU oracle: textures, piecewise-constant distributions, the sky, the five
 * materials and their BxDF math.  TEST INFRASTRUCTURE (see ora_internal.h).
 * Restates crates/implementations/src/{textures/mod.rs, statistics/distributions.rs, sky.rs,
 * materials/, statistics/bxdfs/}.
 */
#include "ora_internal.h"
#include <stdlib.h>
#include <string.h>

/* `as usize` on a float: saturating, NaN -> 0 */
static inline uint64_t f32_as_usize(float f)
{
	if (!(f > 0.0f))
		return 0;
	if (f >= 1.8446744e19f)
		return UINT64_MAX;
	return (uint64_t)f;
}
/* `as i32` on a float: saturating, NaN -> 0 */
static inline int32_t f32_as_i32(float f)
{
	if (f != f)
		return 0;
	if (f >= 2147483648.0f)
		return INT32_MAX;
	if (f <= -2147483648.0f)
		return INT32_MIN;
	return (int32_t)f;
}
static inline uint64_t clamp_u64(uint64_t v, uint64_t lo, uint64_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ================= textures/mod.rs ================= */

/* Perlin::noise / trilinear_lerp  :110-169 */
static float perlin_noise(const ora_texture *t, vec3 point)
{
	const float u = point.x - floorf(point.x);
	const float v = point.y - floorf(point.y);
	const float w = point.z - floorf(point.z);
	const int32_t i = f32_as_i32(floorf(point.x));
	const int32_t j = f32_as_i32(floorf(point.y));
	const int32_t k = f32_as_i32(floorf(point.z));
	vec3 c[8];
	for (int index = 0; index < 8; ++index) {
		const int32_t di = index / 4, dj = (index / 2) % 2, dk = index % 2;
		const uint32_t a = t->perm_x[(uint32_t)(i + di) & 255u] ^ t->perm_y[(uint32_t)(j + dj) & 255u] ^
		                   t->perm_z[(uint32_t)(k + dk) & 255u];
		c[index] = t->ran_vecs[a & 255u];
	}
	const float uu = u * u * (3.0f - 2.0f * u);
	const float vv = v * v * (3.0f - 2.0f * v);
	const float ww = w * w * (3.0f - 2.0f * w);
	float value = 0.0f;
	for (int index = 0; index < 8; ++index) {
		const int ii = index / 4, jj = (index / 2) % 2, kk = index % 2;
		const float fi = (float)ii, fj = (float)jj, fk = (float)kk;
		const vec3 weight = v3(u - fi, v - fj, w - fk);
		value += (fi * uu + (1.0f - fi) * (1.0f - uu)) * (fj * vv + (1.0f - fj) * (1.0f - vv)) *
		         (fk * ww + (1.0f - fk) * (1.0f - ww)) * v3_dot(c[ii * 4 + jj * 2 + kk], weight);
	}
	return value;
}

/* Texture::colour_value, enum-dispatched (#[derive(Texture)] proc/src/lib.rs:68-124) */
vec3 ora_texture_colour_value(const ora_scene *s, uint32_t tex, vec3 direction, vec3 point)
{
	const ora_texture *t = &s->textures[tex];
	switch (t->type) {
	case RT_TEX_CHECKERED: { /* :61-73 */
		const float sign = rt_sinf(10.0f * point.x) * rt_sinf(10.0f * point.y) * rt_sinf(10.0f * point.z);
		return sign > 0.0f ? t->colour_one : t->colour_two;
	}
	case RT_TEX_SOLID: /* :193-200 */
		return t->colour_one;
	case RT_TEX_IMAGE: { /* :251-262 */
		const float phi = rt_atan2f(direction.y, direction.x) + ORA_PI;
		const float theta = rt_acosf(direction.z);
		const float uvx = phi / (2.0f * ORA_PI);
		const float uvy = theta / ORA_PI;
		const uint64_t x_pixel = f32_as_usize((float)t->dim_x * uvx);
		const uint64_t y_pixel = f32_as_usize((float)t->dim_y * uvy);
		uint64_t index = y_pixel * (t->dim_x + 1) + x_pixel;
		const uint64_t n = (t->dim_x + 1) * (t->dim_y + 1);
		if (index >= n) /* the reference would panic on the out-of-range index */
			index = n - 1;
		return t->image[index];
	}
	case RT_TEX_LERP: { /* :283-291 */
		const float tt = direction.z * 0.5f + 0.5f;
		return v3_add(v3_muls(t->colour_one, tt), v3_muls(t->colour_two, 1.0f - tt));
	}
	case RT_TEX_PERLIN: /* :171-179 : 0.5 * Vec3::one() * (1.0 + noise) */
		return v3_muls(v3_smul(0.5f, v3_one()), 1.0f + perlin_noise(t, point));
	default:
		return v3_one(); /* trait default :10-12 */
	}
}

/* ================= statistics/distributions.rs ================= */

/* Distribution1D::new  :12-44 */
static void dist1d_new(ora_dist1d *d, const float *values, uint64_t n)
{
	d->n = n;
	d->cdf = (float *)malloc((n + 1) * sizeof(float));
	d->pdf = (float *)malloc((n ? n : 1) * sizeof(float));
	d->cdf[0] = 0.0f;
	for (uint64_t i = 1; i <= n; ++i)
		d->cdf[i] = d->cdf[i - 1] + values[i - 1];
	const float c = d->cdf[n];
	if (c != 0.0f)
		for (uint64_t i = 0; i <= n; ++i)
			d->cdf[i] /= c;
	float last = 0.0f;
	for (uint64_t i = 0; i < n; ++i) {
		d->pdf[i] = d->cdf[i + 1] - last;
		last = d->cdf[i + 1];
	}
}
static void dist1d_free(ora_dist1d *d)
{
	free(d->pdf);
	free(d->cdf);
}

/* Distribution1D::sample  :51-72 */
uint64_t ora_dist1d_sample(const ora_dist1d *d, ora_ctx *ctx)
{
	const float num = ora_random_float(ctx);
	uint64_t first = 0;
	uint64_t len = d->n + 1;
	while (len > 0) {
		const uint64_t half = len >> 1;
		const uint64_t middle = first + half;
		if (d->cdf[middle] <= num) {
			first = middle + 1;
			len -= half + 1;
		} else {
			len = half;
		}
	}
	/* (first - 1).clamp(0, cdf.len() - 2); cdf[0] == 0 <= num so first >= 1 */
	return clamp_u64(first - 1, 0, d->n - 1);
}

/* Distribution2D::new  :83-99 */
static void dist2d_new(ora_dist2d *d, const float *values, uint64_t n_values, uint64_t width)
{
	const uint64_t height = n_values / width;
	d->dim_x = width;
	d->dim_y = height;
	d->x_distributions = (ora_dist1d *)malloc(height * sizeof(ora_dist1d));
	float *y_values = (float *)malloc(height * sizeof(float));
	for (uint64_t r = 0; r < height; ++r) {
		const float *row = values + r * width;
		dist1d_new(&d->x_distributions[r], row, width);
		float row_sum = 0.0f; /* vec_x.iter().sum() */
		for (uint64_t i = 0; i < width; ++i)
			row_sum += row[i];
		y_values[r] = row_sum;
	}
	dist1d_new(&d->y_distribution, y_values, height);
	free(y_values);
}
/* Distribution2D::pdf  :105-110 */
static float dist2d_pdf(const ora_dist2d *d, float u, float v)
{
	const uint64_t ui = clamp_u64(f32_as_usize((float)d->dim_x * u), 0, d->dim_x - 1);
	const uint64_t vi = clamp_u64(f32_as_usize((float)d->dim_y * v), 0, d->dim_y - 1);
	return d->y_distribution.pdf[vi] * d->x_distributions[vi].pdf[ui];
}

/* ================= sky.rs ================= */

/* generate_values  textures/mod.rs:32-50 + Sky::new  sky.rs:22-39 */
int ora_sky_build(ora_scene *s)
{
	ora_sky *sky = &s->sky;
	sky->has_distribution = false;
	const uint64_t rx = sky->res_x, ry = sky->res_y;
	if ((rx | ry) == 0) /* `sampler_res.0 | sampler_res.1 != 0` parses as (a|b) != 0 (P-hazard 5) */
		return RT_OK;
	if (rx == 0 || ry == 0)
		return RT_ERR_INVALID_ARGUMENT; /* the reference panics in Distribution2D::new */
	float *values = (float *)malloc(rx * ry * sizeof(float));
	const float step0 = 1.0f / (float)rx, step1 = 1.0f / (float)ry;
	uint64_t k = 0;
	for (uint64_t y = 0; y < ry; ++y) {
		for (uint64_t x = 0; x < rx; ++x) {
			const float u = ((float)x + 0.5f) * step0;
			const float v = ((float)y + 0.5f) * step1;
			const float phi = u * 2.0f * ORA_PI;
			const float theta = v * ORA_PI;
			const float sin_theta = rt_sinf(theta);
			const vec3 direction = v3(rt_cosf(phi) * sin_theta, rt_sinf(phi) * sin_theta, rt_cosf(theta));
			const vec3 col = ora_texture_colour_value(s, sky->texture, direction, v3_zero());
			values[k++] = (0.2126f * col.x + 0.7152f * col.y + 0.0722f * col.z) * sin_theta;
		}
	}
	dist2d_new(&sky->distribution, values, rx * ry, rx);
	sky->has_distribution = true;
	free(values);
	return RT_OK;
}
void ora_sky_free(ora_scene *s)
{
	if (!s->sky.has_distribution)
		return;
	for (uint64_t r = 0; r < s->sky.distribution.dim_y; ++r)
		dist1d_free(&s->sky.distribution.x_distributions[r]);
	free(s->sky.distribution.x_distributions);
	dist1d_free(&s->sky.distribution.y_distribution);
}

bool ora_sky_can_sample(const ora_scene *s) { return (s->sky.res_x | s->sky.res_y) != 0; } /* :61-63 */

float ora_sky_pdf(const ora_scene *s, vec3 wi, ora_ctx *ctx) /* :43-60 */
{
	ctx->c.sky_ops++;
	const float sin_theta = sqrtf(1.0f - wi.z * wi.z);
	if (sin_theta <= 0.0f)
		return 0.0f;
	const float theta = rt_acosf(wi.z);
	float phi = rt_atan2f(wi.y, wi.x);
	if (phi < 0.0f)
		phi += 2.0f * ORA_PI;
	const float u = phi / (2.0f * ORA_PI);
	const float v = theta / ORA_PI;
	return (float)s->sky.res_x * (float)s->sky.res_y * dist2d_pdf(&s->sky.distribution, u, v) /
	       (sin_theta * ORA_TAU * ORA_PI);
}

vec3 ora_sky_sample(const ora_scene *s, ora_ctx *ctx) /* :64-78 */
{
	ctx->c.sky_ops++;
	const ora_dist2d *d = &s->sky.distribution;
	/* Distribution2D::sample  distributions.rs:100-104 */
	const uint64_t sv = ora_dist1d_sample(&d->y_distribution, ctx);
	const uint64_t su = ora_dist1d_sample(&d->x_distributions[sv], ctx);

	const float u = ora_next_float((float)su + ora_random_float(ctx)) / (float)s->sky.res_x;
	const float v = ora_next_float((float)sv + ora_random_float(ctx)) / (float)s->sky.res_y;

	const float phi = u * 2.0f * ORA_PI;
	const float theta = v * ORA_PI;
	return v3_from_spherical(rt_sinf(theta), rt_cosf(theta), rt_sinf(phi), rt_cosf(phi));
}

ora_si ora_sky_get_si(const ora_scene *s) /* :79-92 */
{
	ora_si si;
	si.hit.t = 0.0f;
	si.hit.point = v3_zero();
	si.hit.error = v3_zero();
	si.hit.normal = v3_zero();
	si.hit.uv.x = si.hit.uv.y = 0.0f;
	si.hit.has_uv = false;
	si.hit.out = false;
	si.material = s->sky.material;
	return si;
}

/* ================= statistics/bxdfs ================= */

/* lambertian.rs:5-22 */
vec3 ora_lambertian_sample(vec3 incoming, vec3 normal, ora_ctx *ctx)
{
	(void)incoming;
	const float cos_theta = sqrtf(1.0f - ora_random_float(ctx));
	const float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
	const float phi = 2.0f * ORA_PI * ora_random_float(ctx);
	const vec3 local = v3(rt_cosf(phi) * sin_theta, rt_sinf(phi) * sin_theta, cos_theta);
	const ora_coord c = ora_coord_new_from_z(normal);
	return ora_coord_to_coord(&c, local);
}
float ora_lambertian_pdf(vec3 incoming, vec3 outgoing, vec3 normal)
{
	(void)incoming;
	return f_max(v3_dot(outgoing, normal), 0.0f) / ORA_PI;
}

/* trowbridge_reitz.rs:14-21 */
static float tr_d(float alpha, float cos_theta)
{
	if (cos_theta <= 0.0f)
		return 0.0f;
	const float a_sq = alpha * alpha;
	const float tmp = cos_theta * cos_theta * (a_sq - 1.0f) + 1.0f;
	return a_sq / (ORA_PI * tmp * tmp);
}
/* trowbridge_reitz.rs:61-78 */
static float tr_g2(float alpha, vec3 normal, vec3 h, vec3 incoming, vec3 outgoing)
{
	if (v3_dot(incoming, h) / v3_dot(incoming, normal) <= 0.0f ||
	    v3_dot(outgoing, h) / v3_dot(outgoing, normal) <= 0.0f)
		return 0.0f;
	const float alpha_sq = alpha * alpha;
	const float one_minus_alpha_sq = 1.0f - alpha_sq;
	const float cos_i = v3_dot(normal, incoming);
	const float cos_i_sq = cos_i * cos_i;
	const float tmp_a = alpha_sq + one_minus_alpha_sq * cos_i_sq;
	const float cos_o = v3_dot(normal, outgoing);
	const float cos_o_sq = cos_o * cos_o;
	const float tmp_b = alpha_sq + one_minus_alpha_sq * cos_o_sq;
	return 2.0f * cos_i * cos_o / (cos_o * sqrtf(tmp_a) + cos_i * sqrtf(tmp_b));
}
/* trowbridge_reitz.rs:80-89 */
static float tr_g1(float alpha, vec3 normal, vec3 h, vec3 v)
{
	if (v3_dot(v, h) / v3_dot(v, normal) <= 0.0f)
		return 0.0f;
	const float c = v3_dot(normal, v);
	const float cos_sq = c * c;
	const float alpha_sq = alpha * alpha;
	const float tmp = alpha_sq + (1.0f - alpha_sq) * cos_sq;
	return 2.0f * c / (sqrtf(tmp) + c);
}
/* test hooks: d / g1 / g2 exactly as the shading code above calls them, over arrays, so that tests/ can restate the
 * reference's quadrature identities (statistics/bxdfs/trowbridge_reitz.rs:128-230) on THESE functions */
int ora_tr_d_many(float alpha, const float *cos_theta, uint64_t n, float *out)
{
	for (uint64_t i = 0; i < n; ++i)
		out[i] = tr_d(alpha, cos_theta[i]);
	return RT_OK;
}
int ora_tr_g1_many(float alpha, const float normal[3], const float *h, const float v[3], uint64_t n, float *out)
{
	for (uint64_t i = 0; i < n; ++i)
		out[i] = tr_g1(alpha, v3_from(normal), v3_from(&h[3 * i]), v3_from(v));
	return RT_OK;
}
int ora_tr_g2_many(float alpha, const float normal[3], const float *h, const float incoming[3], const float *outgoing, uint64_t n,
                   float *out)
{
	for (uint64_t i = 0; i < n; ++i)
		out[i] = tr_g2(alpha, v3_from(normal), v3_from(&h[3 * i]), v3_from(incoming), v3_from(&outgoing[3 * i]));
	return RT_OK;
}
/* test hook: Distribution2D::new + n draws of Distribution2D::sample (distributions.rs:83-104), and the discrete pdf
 * y_distribution.pdf[y] * x_distributions[y].pdf[x] the reference's random_2d tests compare against (:206-213) */
int ora_dist2d_sample_many(const float *values, uint64_t n_values, uint64_t width, uint64_t seed, uint64_t n, uint32_t *out_x,
                           uint32_t *out_y, float *out_pdf)
{
	if (!values || width == 0 || n_values == 0 || n_values % width != 0)
		return RT_ERR_INVALID_ARGUMENT;
	ora_dist2d d;
	dist2d_new(&d, values, n_values, width);
	ora_ctx ctx;
	ora_ctx_init(&ctx);
	rt_rng_seed(&ctx.rng, seed, 0, 0);
	for (uint64_t i = 0; i < n; ++i) {
		const uint64_t sv = ora_dist1d_sample(&d.y_distribution, &ctx); /* Distribution2D::sample :100-104 */
		const uint64_t su = ora_dist1d_sample(&d.x_distributions[sv], &ctx);
		out_x[i] = (uint32_t)su;
		out_y[i] = (uint32_t)sv;
	}
	if (out_pdf)
		for (uint64_t y = 0; y < d.dim_y; ++y)
			for (uint64_t x = 0; x < d.dim_x; ++x)
				out_pdf[y * d.dim_x + x] = d.y_distribution.pdf[y] * d.x_distributions[y].pdf[x];
	ora_ctx_free(&ctx);
	for (uint64_t r = 0; r < d.dim_y; ++r)
		dist1d_free(&d.x_distributions[r]);
	free(d.x_distributions);
	dist1d_free(&d.y_distribution);
	return RT_OK;
}

/* trowbridge_reitz_vndf.rs:9-15  isotropic::vndf */
static float tr_vndf(float a, vec3 h, vec3 incoming)
{
	if (h.z < 0.0f)
		return 0.0f;
	return tr_g1(a, v3(0.0f, 0.0f, 1.0f), h, incoming) * f_max(v3_dot(incoming, h), 0.0f) * tr_d(a, h.z) /
	       incoming.z;
}
/* trowbridge_reitz_vndf.rs:80-108  ansiotropic::sample_vndf (isotropic calls it with a_x = a_y) */
static vec3 tr_sample_vndf(float a_x, float a_y, vec3 incoming, ora_ctx *ctx)
{
	const vec3 v_hemisphere = v3_normalised(v3(a_x * incoming.x, a_y * incoming.y, incoming.z));
	const float len_sq = v_hemisphere.x * v_hemisphere.x + v_hemisphere.y * v_hemisphere.y;
	const vec3 basis_two =
	    len_sq > 0.0f ? v3_divs(v3(-v_hemisphere.y, v_hemisphere.x, 0.0f), sqrtf(len_sq)) : v3(1.0f, 0.0f, 0.0f);
	const vec3 basis_three = v3_cross(v_hemisphere, basis_two);

	const float r = sqrtf(ora_random_float(ctx));
	const float phi = ORA_TAU * ora_random_float(ctx);
	float tx = r * rt_cosf(phi);
	float ty = r * rt_sinf(phi);
	const float sv = 0.5f * (1.0f + v_hemisphere.z);
	ty = (1.0f - sv) * sqrtf(1.0f - tx * tx) + sv * ty;

	const vec3 h_hemisphere =
	    v3_add(v3_add(v3_smul(tx, basis_two), v3_smul(ty, basis_three)),
	           v3_smul(sqrtf(f_max(1.0f - tx * tx - ty * ty, 0.0f)), v_hemisphere));

	return v3_normalised(v3(a_x * h_hemisphere.x, a_y * h_hemisphere.y, f_max(h_hemisphere.z, 0.0f)));
}
/* the two halves of the reference's `isotropic_h` test (trowbridge_reitz_vndf.rs:157-165): isotropic::sample_vndf and
 * isotropic::vndf on half vectors, local frame */
vec3 ora_tr_vndf_sample_h(float alpha, vec3 incoming, ora_ctx *ctx) { return tr_sample_vndf(alpha, alpha, incoming, ctx); }
float ora_tr_vndf_h(float alpha, vec3 h, vec3 incoming) { return tr_vndf(alpha, h, incoming); }
/* trowbridge_reitz_vndf.rs:37-42  isotropic::sample */
vec3 ora_tr_vndf_sample(float alpha, vec3 incoming, vec3 normal, ora_ctx *ctx)
{
	const ora_coord coord = ora_coord_new_from_z(normal);
	const ora_coord inverse = ora_coord_create_inverse(&coord);
	const vec3 h = ora_coord_to_coord(&coord, tr_sample_vndf(alpha, alpha, ora_coord_to_coord(&inverse, incoming), ctx));
	return v3_reflected(incoming, h);
}
/* trowbridge_reitz_vndf.rs:44-54  isotropic::pdf */
float ora_tr_vndf_pdf(float alpha, vec3 incoming, vec3 outgoing, vec3 normal)
{
	const ora_coord coord = ora_coord_new_from_z(normal);
	const ora_coord inverse = ora_coord_create_inverse(&coord);
	incoming = ora_coord_to_coord(&inverse, incoming);
	outgoing = ora_coord_to_coord(&inverse, outgoing);
	vec3 h = v3_normalised(v3_add(outgoing, incoming));
	if (h.z < 0.0f)
		h = v3_neg(h);
	const float vndf = tr_vndf(alpha, h, incoming);
	return vndf / (4.0f * v3_dot(incoming, h));
}

/* ================= materials ================= */

/* refract.rs:59-61 */
static vec3 fresnel(float c, vec3 f0)
{
	return v3_add(f0, v3_muls(v3_ssub(1.0f, f0), rt_pow5f(1.0f - c)));
}
/* trowbridge_reitz.rs:26-31 + lerp :89-91 */
static vec3 tr_fresnel(const ora_scene *s, const ora_material *m, const ora_hit *hit, vec3 wo, vec3 wi, vec3 h)
{
	vec3 f0 = v3_abs(v3_div(v3_ssub(1.0f, m->ior), v3_sadd(1.0f, m->ior)));
	f0 = v3_mul(f0, f0);
	const vec3 tex = ora_texture_colour_value(s, m->texture, wi, hit->point);
	f0 = v3_add(v3_smul(1.0f - m->metallic, f0), v3_smul(m->metallic, tex));
	return fresnel(v3_dot(wo, h), f0);
}

/* reflect.rs:25-35 */
static bool reflect_scatter(float fuzz, ora_ray *ray, const ora_hit *hit, ora_ctx *ctx)
{
	vec3 direction = v3_neg(ray->direction);
	direction = v3_reflected(direction, hit->normal);
	const vec3 point = ora_offset_ray_v(hit->point, hit->normal, hit->error, true);
	const vec3 ruv = ora_random_unit_vector(&ctx->rng);
	*ray = ora_ray_new(point, v3_add(direction, v3_smul(fuzz, ruv)), ray->time);
	return false;
}

bool ora_mat_scatter_ray(const ora_scene *s, uint32_t mat, ora_ray *ray, const ora_hit *hit, ora_ctx *ctx)
{
	const ora_material *m = &s->materials[mat];
	switch (m->type) {
	case RT_MAT_LAMBERTIAN: { /* lambertian.rs:30-41 */
		const vec3 direction = ora_lambertian_sample(ray->direction, hit->normal, ctx);
		const vec3 point = ora_offset_ray_v(hit->point, hit->normal, hit->error, true);
		*ray = ora_ray_new(point, direction, ray->time);
		return false;
	}
	case RT_MAT_REFLECT:
		return reflect_scatter(m->param, ray, hit, ctx);
	case RT_MAT_REFRACT: { /* refract.rs:26-50 */
		const float eta = m->param;
		float eta_fraction = 1.0f / eta;
		if (!hit->out)
			eta_fraction = eta;
		const float cos_theta = f_min(v3_dot(v3_neg(ray->direction), hit->normal), 1.0f);
		const float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
		const bool cannot_refract = eta_fraction * sin_theta > 1.0f;
		float f0s = (1.0f - eta_fraction) / (1.0f + eta_fraction);
		const vec3 f0 = v3_smul(f0s * f0s, v3_one());
		if (cannot_refract || fresnel(cos_theta, f0).x > ora_random_float(ctx))
			return reflect_scatter(0.0f, ray, hit, ctx); /* Reflect::new(texture, 0.0).scatter_ray */
		const vec3 perp = v3_smul(eta_fraction, v3_add(ray->direction, v3_smul(cos_theta, hit->normal)));
		const vec3 para = v3_smul(-1.0f * sqrtf(fabsf(1.0f - v3_mag_sq(perp))), hit->normal);
		const vec3 direction = v3_add(perp, para);
		const vec3 point = ora_offset_ray_v(hit->point, hit->normal, hit->error, false);
		*ray = ora_ray_new(point, direction, ray->time);
		return false;
	}
	case RT_MAT_TROWBRIDGE_REITZ: { /* trowbridge_reitz.rs:38-51 */
		const vec3 direction = ora_tr_vndf_sample(m->param, v3_neg(ray->direction), hit->normal, ctx);
		const vec3 point = ora_offset_ray_v(hit->point, hit->normal, hit->error, true);
		*ray = ora_ray_new(point, direction, ray->time);
		return false;
	}
	case RT_MAT_EMIT: /* emissive.rs:36-38 */
	default:          /* trait default material.rs:5-7 */
		return true;
	}
}

bool ora_mat_is_light(const ora_scene *s, uint32_t mat) { return s->materials[mat].type == RT_MAT_EMIT; }
bool ora_mat_is_delta(const ora_scene *s, uint32_t mat)
{
	const int32_t t = s->materials[mat].type;
	return t == RT_MAT_REFLECT || t == RT_MAT_REFRACT;
}

float ora_mat_scattering_pdf(const ora_scene *s, uint32_t mat, const ora_hit *hit, vec3 wo, vec3 wi)
{
	const ora_material *m = &s->materials[mat];
	switch (m->type) {
	case RT_MAT_LAMBERTIAN: /* lambertian.rs:42-44 */
		return ora_lambertian_pdf(wo, wi, hit->normal);
	case RT_MAT_TROWBRIDGE_REITZ: { /* trowbridge_reitz.rs:52-60 */
		const float a = ora_tr_vndf_pdf(m->param, v3_neg(wo), wi, hit->normal);
		return a == 0.0f ? INFINITY : a;
	}
	default: /* trait default 0.0 (Reflect, Refract); Emit is unreachable!() */
		return 0.0f;
	}
}

vec3 ora_mat_eval(const ora_scene *s, uint32_t mat, const ora_hit *hit, vec3 wo, vec3 wi)
{
	const ora_material *m = &s->materials[mat];
	switch (m->type) {
	case RT_MAT_LAMBERTIAN: /* lambertian.rs:45-47 */
		return v3_divs(v3_muls(v3_muls(ora_texture_colour_value(s, m->texture, wo, hit->point), m->param),
		                       f_max(v3_dot(hit->normal, wi), 0.0f)),
		               ORA_PI);
	case RT_MAT_REFLECT: /* reflect.rs:36-38 */
	case RT_MAT_REFRACT: /* refract.rs:51-53 */
		return ora_texture_colour_value(s, m->texture, wo, hit->point);
	case RT_MAT_TROWBRIDGE_REITZ: { /* trowbridge_reitz.rs:61-74 */
		const vec3 wom = v3_neg(wo);
		const vec3 h = v3_normalised(v3_add(wi, wom));
		if (v3_dot(wi, hit->normal) < 0.0f || v3_dot(h, wom) < 0.0f)
			return v3_zero();
		const vec3 f = tr_fresnel(s, m, hit, wom, wi, h);
		const float g = tr_g2(m->param, hit->normal, h, wom, wi);
		const float d = tr_d(m->param, v3_dot(hit->normal, h));
		return v3_divs(v3_muls(v3_muls(f, g), d), 4.0f * fabsf(v3_dot(wom, hit->normal)) * v3_dot(wi, hit->normal));
	}
	default: /* Emit: unreachable!() in the reference */
		return v3_zero();
	}
}

vec3 ora_mat_eval_over_scattering_pdf(const ora_scene *s, uint32_t mat, const ora_hit *hit, vec3 wo, vec3 wi)
{
	const ora_material *m = &s->materials[mat];
	switch (m->type) {
	case RT_MAT_LAMBERTIAN: /* lambertian.rs:48-50 */
		return v3_muls(ora_texture_colour_value(s, m->texture, wo, hit->point), m->param);
	case RT_MAT_TROWBRIDGE_REITZ: { /* trowbridge_reitz.rs:75-87 */
		const vec3 wom = v3_neg(wo);
		const vec3 h = v3_normalised(v3_add(wi, wom));
		if (v3_dot(wom, h) < 0.0f || v3_dot(wi, hit->normal) < 0.0f)
			return v3_zero();
		const vec3 f = tr_fresnel(s, m, hit, wom, wi, h);
		const float g = tr_g2(m->param, hit->normal, h, wom, wi);
		return v3_divs(v3_muls(f, g), tr_g1(m->param, hit->normal, h, wom));
	}
	default: /* trait default material.rs:24-26: eval / scattering_pdf */
		return v3_divs(ora_mat_eval(s, mat, hit, wo, wi), ora_mat_scattering_pdf(s, mat, hit, wo, wi));
	}
}

vec3 ora_mat_get_emission(const ora_scene *s, uint32_t mat, const ora_hit *hit, vec3 wo)
{
	const ora_material *m = &s->materials[mat];
	if (m->type == RT_MAT_EMIT) { /* emissive.rs:23-26 */
		const vec3 point = ora_offset_ray_v(hit->point, hit->normal, hit->error, true);
		return v3_smul(m->param, ora_texture_colour_value(s, m->texture, wo, point));
	}
	return v3_zero(); /* trait default material.rs:27-29 */
}
