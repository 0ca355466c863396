// rt_shade.h -- the per-bounce shade/sample step on the device: textures, the importance-sampled
// sky, the five materials and their BxDF math.
//
// Reference (paths under crates/implementations/src/): textures/mod.rs:9-291,
// statistics/distributions.rs:51-72,100-110, sky.rs:43-92, materials/{emissive,lambertian,reflect,
// refract,trowbridge_reitz}.rs, statistics/bxdfs/{lambertian,trowbridge_reitz,
// trowbridge_reitz_vndf}.rs, utility/{mod,coord}.rs.  The enum dispatch the reference generates with
// proc macros (proc/src/lib.rs) is a switch on the type tag here.
#pragma once

#include "rt_intersect.h"

namespace rt {

// sqrtf for arguments that are provably zero, negative, NaN or >= 2^-96 (each use says why): the short form of rt_lean.h
__device__ __forceinline__ float sqrt_unit_(float x)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RT_NO_LEAN)
	return lean_sqrt(x);
#else
	return sqrtf(x);
#endif
}
// n / d for a tame d and an n that is provably zero, infinite, NaN or tame (each use says why)
__device__ __forceinline__ float div_tame_fix_(float n, float d)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RT_NO_LEAN)
	return lean_div_fix(n, d);
#else
	return n / d;
#endif
}

// ---- utility/mod.rs ----
__device__ __forceinline__ float next_float(float f) // :51-65
{
	if (f == INFINITY)
		return f;
	if (f == 0.0f) // catches -0.0 too; the reference sets it to +0.0
		f = 0.0f;
	const uint32_t bits = __float_as_uint(f);
	return __uint_as_float(f >= 0.0f ? bits + 1u : bits - 1u);
}
__device__ __forceinline__ float previous_float(float f) // :67-81
{
	if (f == -INFINITY)
		return f;
	if (f == 0.0f)
		f = -0.0f;
	const uint32_t bits = __float_as_uint(f);
	return __uint_as_float(f <= 0.0f ? bits + 1u : bits - 1u);
}
// offset_ray(origin, normal, error, is_brdf) with offset_val = dot(|normal|, error) already formed (Hit::err_dot)
__device__ __forceinline__ V3 offset_ray(V3 origin, V3 normal, float offset_val, bool is_brdf) // :88-117
{
	V3 offset = offset_val * normal;
	if (!is_brdf)
		offset = -offset;
	V3 o = origin + offset;
	o.x = offset.x > 0.0f ? next_float(o.x) : previous_float(o.x);
	o.y = offset.y > 0.0f ? next_float(o.y) : previous_float(o.y);
	o.z = offset.z > 0.0f ? next_float(o.z) : previous_float(o.z);
	return o;
}
__device__ __forceinline__ V3 random_unit_vector(rt_rng &rng) // :15-25
{
	float x = 1.0f, y = 1.0f, z = 1.0f;
	while (x * x + y * y + z * z > 1.0f) {
		x = rt_rng_range_f32(&rng, -1.0f, 1.0f);
		y = rt_rng_range_f32(&rng, -1.0f, 1.0f);
		z = rt_rng_range_f32(&rng, -1.0f, 1.0f);
	}
	return normalised(v3(x, y, z));
}

// ---- utility/coord.rs:9-31 ----
struct Coord {
	V3 x, y, z;
};
__device__ __forceinline__ Coord coord_from_z(V3 z)
{
	Coord c;
	if (fabsf(z.x) > fabsf(z.y))
		c.x = v3(-z.z, 0.0f, z.x) / sqrtf(z.x * z.x + z.z * z.z);
	else
		c.x = v3(0.0f, z.z, -z.y) / sqrtf(z.y * z.y + z.z * z.z);
	c.y = cross(c.x, z);
	c.z = z;
	return c;
}
// The same frame with the zero component of the numerator not divided: 0 / s for s = sqrtf(...) in [0, inf] or NaN is +0 when
// s > 0 and NaN otherwise (0 / 0, 0 / NaN) -- a compare and a select instead of the eleven instructions of an IEEE division.
// Same bits; used by the kernels with triangles, where it measured faster (config 3 -0.5 %).  The spheres-only kernels keep the
// form above: there it measured 0.4 - 1.5 % SLOWER depending on how it was spelled (profiles/r04j_*, r04m_*: these kernels sit
// on a scheduling knife edge where the text of a function that computes the same thing moves a launch by a millisecond).
__device__ __forceinline__ Coord coord_from_z_zsel(V3 z)
{
	Coord c;
	if (fabsf(z.x) > fabsf(z.y)) {
		const float s = sqrtf(z.x * z.x + z.z * z.z);
		c.x = v3(-z.z / s, s > 0.0f ? 0.0f : __builtin_nanf(""), z.x / s);
	} else {
		const float s = sqrtf(z.y * z.y + z.z * z.z);
		c.x = v3(s > 0.0f ? 0.0f : __builtin_nanf(""), z.z / s, -z.y / s);
	}
	c.y = cross(c.x, z);
	c.z = z;
	return c;
}
__device__ __forceinline__ Coord coord_inverse(const Coord &c)
{
	Coord r;
	r.x = v3(c.x.x, c.y.x, c.z.x);
	r.y = v3(c.x.y, c.y.y, c.z.y);
	r.z = v3(c.x.z, c.y.z, c.z.z);
	return r;
}
__device__ __forceinline__ V3 to_coord(const Coord &c, V3 v) { return v.x * c.x + v.y * c.y + v.z * c.z; }

// Rust `as usize` / `as i32` on floats saturate and map NaN to 0
__device__ __forceinline__ uint32_t f32_as_index(float f)
{
	if (!(f > 0.0f))
		return 0u;
	if (f >= 4294967040.0f)
		return 0xFFFFFFFFu;
	return (uint32_t)f;
}
__device__ __forceinline__ int32_t f32_as_i32(float f)
{
	if (f != f)
		return 0;
	if (f >= 2147483648.0f)
		return 2147483647;
	if (f <= -2147483648.0f)
		return (-2147483647 - 1);
	return (int32_t)f;
}

// ---- textures/mod.rs ----
__device__ inline float perlin_noise(const DevTexture &t, V3 point) // :110-169
{
	const float u = point.x - floorf(point.x);
	const float v = point.y - floorf(point.y);
	const float w = point.z - floorf(point.z);
	const int32_t i = f32_as_i32(floorf(point.x));
	const int32_t j = f32_as_i32(floorf(point.y));
	const int32_t k = f32_as_i32(floorf(point.z));
	const float uu = u * u * (3.0f - 2.0f * u);
	const float vv = v * v * (3.0f - 2.0f * v);
	const float ww = w * w * (3.0f - 2.0f * w);
	float value = 0.0f;
	for (int index = 0; index < 8; ++index) {
		const int ii = index / 4, jj = (index / 2) % 2, kk = index % 2;
		const uint32_t a = t.perlin_perm[(uint32_t)(i + ii) & 255u] ^ t.perlin_perm[256 + ((uint32_t)(j + jj) & 255u)] ^
		                   t.perlin_perm[512 + ((uint32_t)(k + kk) & 255u)];
		const float *rv = t.perlin_vecs + 3 * (a & 255u);
		const V3 c = v3(rv[0], rv[1], rv[2]);
		const float fi = (float)ii, fj = (float)jj, fk = (float)kk;
		const V3 weight = v3(u - fi, v - fj, w - fk);
		value += (fi * uu + (1.0f - fi) * (1.0f - uu)) * (fj * vv + (1.0f - fj) * (1.0f - vv)) *
		         (fk * ww + (1.0f - fk) * (1.0f - ww)) * dot(c, weight);
	}
	return value;
}

template <class F> __device__ __forceinline__ V3 texture_colour(const DevScene &S, uint32_t tex, V3 direction, V3 point)
{
	const DevTexture &t = S.textures[tex];
	const int type = t.type;
	if (type == 1) // SolidColour :193-200
		return v3(t.c1[0], t.c1[1], t.c1[2]);
	if (type == 3) { // Lerp :283-291
		const float tt = direction.z * 0.5f + 0.5f;
		return v3(t.c1[0], t.c1[1], t.c1[2]) * tt + v3(t.c2[0], t.c2[1], t.c2[2]) * (1.0f - tt);
	}
	if (!F::ctex)
		return v3s(1.0f); // unreachable: the host launches a ctex variant when such textures exist
	if (type == 0) { // CheckeredTexture :61-73
		const float sign = rt_sinf(10.0f * point.x) * rt_sinf(10.0f * point.y) * rt_sinf(10.0f * point.z);
		return sign > 0.0f ? v3(t.c1[0], t.c1[1], t.c1[2]) : v3(t.c2[0], t.c2[1], t.c2[2]);
	}
	if (type == 2) { // ImageTexture :251-262
		const float phi = lean_atan2(direction.y, direction.x) + kPi;
		const float theta = lean_acos_dev(direction.z);
		const float uvx = phi / (2.0f * kPi);
		const float uvy = theta / kPi;
		const uint32_t x_pixel = f32_as_index((float)t.dim_x * uvx);
		const uint32_t y_pixel = f32_as_index((float)t.dim_y * uvy);
		uint64_t index = (uint64_t)y_pixel * (t.dim_x + 1u) + x_pixel;
		const uint64_t n = (uint64_t)(t.dim_x + 1u) * (t.dim_y + 1u);
		if (index >= n) // the reference would panic here
			index = n - 1;
		const float *px = t.image + 3 * index;
		return v3(px[0], px[1], px[2]);
	}
	if (type == 4) // Perlin :171-179
		return (0.5f * v3s(1.0f)) * (1.0f + perlin_noise(t, point));
	return v3s(1.0f);
}

// What a caller may know about a material record without reading it.  The kernels compiled for a scene whose primitives all
// carry a Lambertian with a SolidColour texture and whose sky is an Emit (F::known_materials: FeatPair, which the host picks
// per launch only for such a scene, rt_api.cpp) pass the type that follows from WHAT WAS HIT -- a primitive: Lambertian, the
// sky: Emit.  The record's type, its texture's type and the loads that only served to find them out are then gone: a material
// evaluation was a chain of three dependent per-lane loads (type -> texture type -> colours), now it is one round of loads
// (a Lambertian's colour and albedo) or none (scattering, pdf, "is it a light"), and the sky's record, the same for every
// lane, arrives through scalar loads (sky_emission_uniform).  The arithmetic on what is read is untouched.
enum : int { kMatRead = -1, kMatEmit = 0, kMatLambertian = 1 };
// (`mat` is a handle, rt_types.h kMatHandleShift: the type rides in it)
template <class F> __device__ __forceinline__ int mat_type_(uint32_t mat, int known)
{
	if (F::known_materials && known != kMatRead)
		return known;
	return mat_handle_type(mat);
}
__device__ __forceinline__ const DevMaterial &mat_record(const DevScene &S, uint32_t mat) { return S.materials[mat_handle_index(mat)]; }

// the texture of material record `m`: SolidColour and Lerp are answered from the copy inside the record
template <class F> __device__ __forceinline__ V3 material_texture_colour(const DevScene &S, const DevMaterial &m, uint32_t mat, V3 direction, V3 point)
{
	const int type = mat_handle_tex_type(mat);
	if (type == 1)
		return v3(m.tex_c1[0], m.tex_c1[1], m.tex_c1[2]);
	if (type == 3) {
		const float tt = direction.z * 0.5f + 0.5f;
		return v3(m.tex_c1[0], m.tex_c1[1], m.tex_c1[2]) * tt + v3(m.tex_c2[0], m.tex_c2[1], m.tex_c2[2]) * (1.0f - tt);
	}
	if (!F::ctex)
		return v3s(1.0f); // unreachable: see texture_colour
	return texture_colour<F>(S, m.texture, direction, point);
}

// texture colour and albedo of a Lambertian: read from its record, or -- F::known_materials -- selected between the two
// records in the kernel arguments by the slot the pair kernels carry as "material" (rt_intersect.h make_hit, rt_types.h
// DevPairScene): SGPR operands and two selects per value instead of a per-lane load
struct LambertRec {
	V3 colour;
	float albedo;
};
template <class F> __device__ __forceinline__ LambertRec lambert_record(const DevScene &S, const DevMaterial &m, uint32_t mat, V3 direction, V3 point, const DevPairScene &ps)
{
	LambertRec L;
	if constexpr (F::known_materials) {
		const bool first = mat == ps.slot0;
		L.colour = first ? v3(ps.lambert[0][0], ps.lambert[0][1], ps.lambert[0][2]) : v3(ps.lambert[1][0], ps.lambert[1][1], ps.lambert[1][2]);
		L.albedo = first ? ps.lambert[0][3] : ps.lambert[1][3];
	} else {
		L.colour = material_texture_colour<F>(S, m, mat, direction, point);
		L.albedo = m.param;
	}
	return L;
}

// A wave-uniform value the optimiser must not see through: conversions and products of it are then computed where they are used
// (one or two instructions) instead of being hoisted out of the persistent loop into a VGPR that lives -- or is spilled to
// scratch and reloaded on the hot path -- for the whole kernel.
__device__ __forceinline__ uint32_t here_(uint32_t uniform)
{
#if defined(__HIP_DEVICE_COMPILE__)
	asm volatile("" : "+s"(uniform));
#endif
	return uniform;
}

// x / (float)n for a wave-uniform count n (the light-selection divisor, acceleration/mod.rs:299-318).  For a power of two --
// one light or none besides the sky: the reference's scenes -- division and multiplication by the exact reciprocal are the
// same correctly rounded operation on the same real number, whatever x is (zero, subnormal, infinite, NaN included).
__device__ __forceinline__ float div_by_count(float x, uint32_t n)
{
	if ((n & (n - 1u)) == 0u && n != 0u) // (wave-uniform)
		return x * __uint_as_float((127u - (uint32_t)__builtin_ctz(n)) << 23);
	return x / (float)n;
}

// ---- statistics/distributions.rs ----
// Distribution1D::sample :51-72 over a cdf of n+1 entries
// `guide` (may be null; wave-uniform): guide_k upper-bound indices for this cdf, see DevSky in rt_types.h
__device__ __forceinline__ uint32_t dist1d_sample(const float *cdf, uint32_t n, const uint8_t *guide, uint32_t guide_k, rt_rng &rng)
{
	const float num = rt_rng_f32(&rng);
	uint32_t first;
	if (guide != nullptr) {
		// the upper bound of num lies at or right of the upper bound of floor(num * K) / K
		first = guide[(uint32_t)(num * (float)guide_k)];
		// `while (first <= n && cdf[first] <= num) ++first;` -- two entries per round of reads instead of one: the CDF is
		// non-decreasing (the guide tables are built only then), so the entries that are <= num form a prefix and the scan may
		// look ahead; one round of two INDEPENDENT reads usually ends it (four at a time cost the config-2 kernel 14 spilled registers), where the one-at-a-time loop paid a dependent
		// LDS round trip per step
		for (;;) {
			const float c0 = cdf[first > n ? n : first], c1 = cdf[first + 1u > n ? n : first + 1u];
			const bool b0 = first <= n && c0 <= num;
			const bool b1 = b0 && first + 1u <= n && c1 <= num;
			first += (b0 ? 1u : 0u) + (b1 ? 1u : 0u);
			if (!b1)
				break;
		}
	} else {
		first = 0;
		uint32_t len = n + 1;
		while (len > 0) {
			const uint32_t half = len >> 1;
			const uint32_t middle = first + half;
			if (cdf[middle] <= num) {
				first = middle + 1;
				len -= half + 1;
			} else {
				len = half;
			}
		}
	}
	const uint32_t v = first - 1u; // cdf[0] = 0 <= num, so first >= 1
	return v > n - 1u ? n - 1u : v;
}

// (inv_res_ok, inv_res_x, inv_res_y) of DevSky, read from the kernel arguments WHERE THEY ARE USED: through `S` they would be
// loaded in the prologue and hold three scalar registers for the life of the persistent loop (rt_render.hip, RenderArgs)
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) uint32_t *KWords;
#else
typedef const uint32_t *KWords;
#endif
struct SkyTables {
	const float *row_cdf;      // [res_y][res_x + 1]
	const float *marginal_cdf; // [res_y + 1]
	const uint8_t *guide;      // [res_y + 1][guide_k] or null
	uint32_t guide_k;
	KWords inv_res;            // -> DevSky::inv_res_ok in the kernarg segment (null: plain division)
};

__device__ __forceinline__ bool sky_can_sample(const DevScene &S) { return (S.sky.res_x | S.sky.res_y) != 0u; } // sky.rs:61-63

// Sky::pdf  sky.rs:43-60 with Distribution2D::pdf distributions.rs:105-110
__device__ __forceinline__ float sky_pdf(const DevScene &S, const SkyTables &T, V3 wi)
{
	// 1 - z * z is zero, negative, NaN or >= 2^-24: inside the range of the short square root (rt_lean.h)
	const float sin_theta = sqrt_unit_(1.0f - wi.z * wi.z);
	if (sin_theta <= 0.0f)
		return 0.0f;
	const float theta = lean_acos_dev(wi.z);
	float phi = lean_atan2(wi.y, wi.x);
	if (phi < 0.0f)
		phi += 2.0f * kPi;
	// u = phi / (2 pi) and v = theta / pi only pick a table cell below, and the divisions are by constants: the verified
	// two-fma form (rt_lean.h div_by_verified; kRcpTau, kRcpPi) returns the bits of the division for phi, theta = 0, NaN or
	// >= 2^-60 -- acos returns zero, NaN or at least 2 asin(2^-12.5) = 3.4e-4; atan2 (plus 2 pi when negative) lies in [0, 2 pi]
	// and can be a tiny positive number, for which either form returns less than 2^-58: with a table of at most 2^24 cells the
	// index below is 0 both ways.  (2.0f * kPi is kTau exactly: doubling a float is exact.)
	const float u = div_by_verified(phi, kTau, kRcpTau);
	const float v = div_by_verified(theta, kPi, kRcpPi);
	const uint32_t rx = here_(S.sky.res_x), ry = here_(S.sky.res_y);
	uint32_t ui = f32_as_index((float)rx * u);
	uint32_t vi = f32_as_index((float)ry * v);
	ui = ui > rx - 1u ? rx - 1u : ui;
	vi = vi > ry - 1u ? ry - 1u : vi;
	// pdf[i] = cdf[i+1] - cdf[i]: the subtraction Distribution1D::new performs (:32-38)
	const float ypdf = T.marginal_cdf[vi + 1u] - T.marginal_cdf[vi];
	const float *row = T.row_cdf + (size_t)vi * (rx + 1u);
	const float xpdf = row[ui + 1u] - row[ui];
	return (float)rx * (float)ry * (ypdf * xpdf) / (sin_theta * kTau * kPi);
}

// Sky::sample  sky.rs:64-78
__device__ __forceinline__ V3 sky_sample(const DevScene &S, const SkyTables &T, rt_rng &rng)
{
	const uint32_t rx = here_(S.sky.res_x), ry = here_(S.sky.res_y);
	const bool guided = T.guide_k != 0u;
	const uint32_t sv = dist1d_sample(T.marginal_cdf, ry, guided ? T.guide + (size_t)ry * T.guide_k : nullptr, T.guide_k, rng);
	const uint32_t su = dist1d_sample(T.row_cdf + (size_t)sv * (rx + 1u), rx, guided ? T.guide + (size_t)sv * T.guide_k : nullptr, T.guide_k, rng);
	// next_float(cell + r) is the smallest subnormal (cell = r = 0) or lies in [2^-24, 256]: the verified two-fma division applies
	// (for the subnormal both forms return it unchanged when the table has one cell and zero otherwise: 2^-149 / n rounds to 0 for
	// n >= 2, and x * rc underflows to 0 with a zero correction)
	const float nu = next_float((float)su + rt_rng_f32(&rng)), nv = next_float((float)sv + rt_rng_f32(&rng));
	float u, v;
	KWords inv = T.inv_res;
#if defined(__HIP_DEVICE_COMPILE__)
	asm volatile("" : "+s"(inv)); // (a pointer the optimiser cannot see through: the loads below stay here)
#endif
	if (inv != nullptr && inv[0] != 0u) { // (wave-uniform)
		u = div_by_verified(nu, (float)rx, __uint_as_float(inv[1]));
		v = div_by_verified(nv, (float)ry, __uint_as_float(inv[2]));
	} else {
		u = nu / (float)rx;
		v = nv / (float)ry;
	}
	const float phi = u * 2.0f * kPi;
	const float theta = v * kPi;
	// u and v lie in (0, 1 + 2^-20]: angles far inside lean_sincos's domain
	float st, ct, sp, cp;
	lean_sincos(theta, st, ct);
	lean_sincos(phi, sp, cp);
	return v3(st * cp, st * sp, ct); // Vec3::from_spherical  vec.rs:155-163
}

// ---- statistics/bxdfs ----
__device__ __forceinline__ V3 lambertian_sample(V3 normal, rt_rng &rng) // lambertian.rs:5-18
{
	// 1 - r with r in [0, 1 - 2^-24] is >= 2^-24; 1 - c * c with c in [2^-12, 1] is zero or >= 2^-24 (rt_lean.h)
	const float cos_theta = sqrt_unit_(1.0f - rt_rng_f32(&rng));
	const float sin_theta = sqrt_unit_(1.0f - cos_theta * cos_theta);
	const float phi = 2.0f * kPi * rt_rng_f32(&rng);
	float sin_phi, cos_phi;
	lean_sincos(phi, sin_phi, cos_phi);
	const V3 local = v3(cos_phi * sin_theta, sin_phi * sin_theta, cos_theta);
	const Coord c = coord_from_z(normal);
	return to_coord(c, local);
}
// (the kernels with triangles: see coord_from_z_zsel)
template <bool ZSEL = false> __device__ __forceinline__ V3 lambertian_sample_(V3 normal, rt_rng &rng) // lambertian.rs:5-18
{
	// 1 - r with r in [0, 1 - 2^-24] is >= 2^-24; 1 - c * c with c in [2^-12, 1] is zero or >= 2^-24 (rt_lean.h)
	const float cos_theta = sqrt_unit_(1.0f - rt_rng_f32(&rng));
	const float sin_theta = sqrt_unit_(1.0f - cos_theta * cos_theta);
	const float phi = 2.0f * kPi * rt_rng_f32(&rng);
	float sin_phi, cos_phi;
	lean_sincos(phi, sin_phi, cos_phi);
	const V3 local = v3(cos_phi * sin_theta, sin_phi * sin_theta, cos_theta);
	const Coord c = ZSEL ? coord_from_z_zsel(normal) : coord_from_z(normal);
	return to_coord(c, local);
}
__device__ __forceinline__ float tr_d(float alpha, float cos_theta) // trowbridge_reitz.rs:14-21
{
	if (cos_theta <= 0.0f)
		return 0.0f;
	const float a_sq = alpha * alpha;
	const float tmp = cos_theta * cos_theta * (a_sq - 1.0f) + 1.0f;
	return a_sq / (kPi * tmp * tmp);
}
__device__ __forceinline__ float tr_g2(float alpha, V3 normal, V3 h, V3 incoming, V3 outgoing) // :61-78
{
	if (dot(incoming, h) / dot(incoming, normal) <= 0.0f || dot(outgoing, h) / dot(outgoing, normal) <= 0.0f)
		return 0.0f;
	const float alpha_sq = alpha * alpha;
	const float one_minus_alpha_sq = 1.0f - alpha_sq;
	const float cos_i = dot(normal, incoming);
	const float tmp_a = alpha_sq + one_minus_alpha_sq * (cos_i * cos_i);
	const float cos_o = dot(normal, outgoing);
	const float tmp_b = alpha_sq + one_minus_alpha_sq * (cos_o * cos_o);
	return 2.0f * cos_i * cos_o / (cos_o * sqrtf(tmp_a) + cos_i * sqrtf(tmp_b));
}
__device__ __forceinline__ float tr_g1(float alpha, V3 normal, V3 h, V3 v) // :80-89
{
	if (dot(v, h) / dot(v, normal) <= 0.0f)
		return 0.0f;
	const float c = dot(normal, v);
	const float alpha_sq = alpha * alpha;
	const float tmp = alpha_sq + (1.0f - alpha_sq) * (c * c);
	return 2.0f * c / (sqrtf(tmp) + c);
}
__device__ __forceinline__ float tr_vndf(float a, V3 h, V3 incoming) // trowbridge_reitz_vndf.rs:9-15
{
	if (h.z < 0.0f)
		return 0.0f;
	return tr_g1(a, v3(0.0f, 0.0f, 1.0f), h, incoming) * fmax_(dot(incoming, h), 0.0f) * tr_d(a, h.z) / incoming.z;
}
__device__ inline V3 tr_sample_vndf(float a_x, float a_y, V3 incoming, rt_rng &rng) // :80-108
{
	const V3 vh = normalised(v3(a_x * incoming.x, a_y * incoming.y, incoming.z));
	const float len_sq = vh.x * vh.x + vh.y * vh.y;
	const V3 basis_two = len_sq > 0.0f ? v3(-vh.y, vh.x, 0.0f) / sqrtf(len_sq) : v3(1.0f, 0.0f, 0.0f);
	const V3 basis_three = cross(vh, basis_two);
	const float r = sqrtf(rt_rng_f32(&rng));
	const float phi = kTau * rt_rng_f32(&rng);
	float sin_phi, cos_phi;
	lean_sincos(phi, sin_phi, cos_phi);
	const float tx = r * cos_phi;
	float ty = r * sin_phi;
	const float s = 0.5f * (1.0f + vh.z);
	ty = (1.0f - s) * sqrtf(1.0f - tx * tx) + s * ty;
	const V3 hh = tx * basis_two + ty * basis_three + sqrtf(fmax_(1.0f - tx * tx - ty * ty, 0.0f)) * vh;
	return normalised(v3(a_x * hh.x, a_y * hh.y, fmax_(hh.z, 0.0f)));
}
__device__ inline V3 tr_sample(float alpha, V3 incoming, V3 normal, rt_rng &rng) // isotropic::sample :37-42
{
	const Coord coord = coord_from_z(normal);
	const Coord inverse = coord_inverse(coord);
	const V3 h = to_coord(coord, tr_sample_vndf(alpha, alpha, to_coord(inverse, incoming), rng));
	return reflected(incoming, h);
}
__device__ inline float tr_pdf(float alpha, V3 incoming, V3 outgoing, V3 normal) // isotropic::pdf :44-54
{
	const Coord coord = coord_from_z(normal);
	const Coord inverse = coord_inverse(coord);
	incoming = to_coord(inverse, incoming);
	outgoing = to_coord(inverse, outgoing);
	V3 h = normalised(outgoing + incoming);
	if (h.z < 0.0f)
		h = -h;
	return tr_vndf(alpha, h, incoming) / (4.0f * dot(incoming, h));
}

// ---- materials ----
__device__ __forceinline__ V3 fresnel(float c, V3 f0) { return f0 + (1.0f - f0) * rt_pow5f(1.0f - c); } // refract.rs:59-61
template <class F> __device__ inline V3 tr_fresnel(const DevScene &S, const DevMaterial &m, uint32_t mat, const Hit &hit, V3 wo, V3 wi, V3 h) // trowbridge_reitz.rs:26-31
{
	const V3 ior = v3(m.ior[0], m.ior[1], m.ior[2]);
	V3 f0 = vabs((1.0f - ior) / (1.0f + ior));
	f0 = f0 * f0;
	const V3 tex = material_texture_colour<F>(S, m, mat, wi, hit.point);
	f0 = (1.0f - m.metallic) * f0 + m.metallic * tex; // lerp :89-91
	return fresnel(dot(wo, h), f0);
}

// May `cosine / pi` and `(colour x albedo) x cosine / pi` of a Lambertian (lambertian.rs:42-47) take the verified two-fma division
// (rt_lean.h div_by_verified: exact for numerators that are zero or in [2^-60, 2^60])?  The cosine is max(dot, 0) >= 0 (NaN
// becomes 0).  With DevScene::lambert_tame -- the host has checked that every Lambertian of the scene has a SolidColour whose
// (colour x albedo) components are zero or in [2^-30, 2^30] and that every vertex normal is finite and below 2^20, so a cosine is
// below 2^22 -- the numerators are tame whenever the cosine is zero or at least 2^-30.  The rare rest takes the plain operator.
// (F::known_materials: the host launches those kernels only for scenes that pass the check -- rt_api.cpp, pair_tree)
template <class F> __device__ __forceinline__ bool cosine_is_tame_(const DevScene &S, float c)
{
	return (F::known_materials || S.lambert_tame != 0u) && !(c < 0x1p-30f && c > 0.0f);
}

__device__ __forceinline__ bool mat_is_light(const DevScene &S, uint32_t mat) { return mat_handle_type(mat) == 0; }
template <class F> __device__ __forceinline__ bool mat_is_light(const DevScene &S, uint32_t mat, int known) { return mat_type_<F>(mat, known) == 0; }
template <class F> __device__ __forceinline__ bool mat_is_delta(const DevScene &S, uint32_t mat)
{
	if (!F::cmat)
		return false;
	const int t = mat_handle_type(mat);
	return t == 3 || t == 4;
}

template <class F> __device__ __forceinline__ bool reflect_scatter(float fuzz, Ray &ray, const Hit &hit, rt_rng &rng) // reflect.rs:25-35
{
	V3 direction = -ray.d;
	direction = reflected(direction, hit.normal);
	const V3 point = offset_ray(hit.point, hit.normal, hit.err_dot, true);
	const V3 ruv = random_unit_vector(rng);
	ray = ray_new<F>(point, direction + fuzz * ruv);
	return false;
}

// Scatter::scatter_ray; returns `exit`
template <class F> __device__ __forceinline__ bool mat_scatter_ray(const DevScene &S, uint32_t mat, Ray &ray, const Hit &hit, rt_rng &rng, int known = kMatRead)
{
	const DevMaterial &m = mat_record(S, mat);
	const int type = mat_type_<F>(mat, known);
	if (type == 1) { // Lambertian  lambertian.rs:30-41
		const V3 direction = F::tri ? lambertian_sample_<true>(hit.normal, rng) : lambertian_sample(hit.normal, rng);
		const V3 point = offset_ray(hit.point, hit.normal, hit.err_dot, true);
		ray = ray_new<F>(point, direction);
		return false;
	}
	if (type == 0) // Emit  emissive.rs:36-38
		return true;
	if (!F::cmat)
		return true; // unreachable: a cmat variant is launched when such materials exist
	if (type == 3) // Reflect
		return reflect_scatter<F>(m.param, ray, hit, rng);
	if (type == 4) { // Refract  refract.rs:26-50
		const float eta = m.param;
		float eta_fraction = 1.0f / eta;
		if (!hit.out)
			eta_fraction = eta;
		const float cos_theta = fmin_(dot(-ray.d, hit.normal), 1.0f);
		const float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
		const bool cannot_refract = eta_fraction * sin_theta > 1.0f;
		const float f0s = (1.0f - eta_fraction) / (1.0f + eta_fraction);
		const V3 f0 = (f0s * f0s) * v3s(1.0f);
		if (cannot_refract || fresnel(cos_theta, f0).x > rt_rng_f32(&rng))
			return reflect_scatter<F>(0.0f, ray, hit, rng);
		const V3 perp = eta_fraction * (ray.d + cos_theta * hit.normal);
		const V3 para = (-1.0f * sqrtf(fabsf(1.0f - mag_sq(perp)))) * hit.normal;
		const V3 point = offset_ray(hit.point, hit.normal, hit.err_dot, false);
		ray = ray_new<F>(point, perp + para);
		return false;
	}
	if (type == 2) { // TrowbridgeReitz  trowbridge_reitz.rs:38-51
		const V3 direction = tr_sample(m.param, -ray.d, hit.normal, rng);
		const V3 point = offset_ray(hit.point, hit.normal, hit.err_dot, true);
		ray = ray_new<F>(point, direction);
		return false;
	}
	return true;
}

template <class F> __device__ __forceinline__ float mat_scattering_pdf(const DevScene &S, uint32_t mat, const Hit &hit, V3 wo, V3 wi, int known = kMatRead)
{
	const DevMaterial &m = mat_record(S, mat);
	const int type = mat_type_<F>(mat, known);
	if (type == 1) { // lambertian.rs:42-44 -> bxdfs::lambertian::pdf
		const float c = fmax_(dot(wi, hit.normal), 0.0f);
		if (__builtin_expect(cosine_is_tame_<F>(S, c), 1))
			return div_by_verified(c, kPi, kRcpPi);
		return c / kPi;
	}
	if (F::cmat && type == 2) { // trowbridge_reitz.rs:52-60
		const float a = tr_pdf(m.param, -wo, wi, hit.normal);
		return a == 0.0f ? INFINITY : a;
	}
	return 0.0f; // trait default (Reflect, Refract)
}

template <class F> __device__ __forceinline__ V3 mat_eval(const DevScene &S, uint32_t mat, const Hit &hit, V3 wo, V3 wi, int known = kMatRead, const DevPairScene &ps = kNoPairScene)
{
	const DevMaterial &m = mat_record(S, mat);
	const int type = mat_type_<F>(mat, known);
	if (type == 1) { // lambertian.rs:45-47
		const LambertRec L = lambert_record<F>(S, m, mat, wo, hit.point, ps);
		const float c = fmax_(dot(hit.normal, wi), 0.0f);
		const V3 num = L.colour * L.albedo * c;
		if (__builtin_expect(cosine_is_tame_<F>(S, c), 1)) // numerators zero or in [2^-60, 2^60]: see cosine_is_tame_
			return v3(div_by_verified(num.x, kPi, kRcpPi), div_by_verified(num.y, kPi, kRcpPi), div_by_verified(num.z, kPi, kRcpPi));
		return num / kPi;
	}
	if (!F::cmat)
		return v3s(0.0f);
	if (type == 3 || type == 4) // reflect.rs:36-38, refract.rs:51-53
		return material_texture_colour<F>(S, m, mat, wo, hit.point);
	if (type == 2) { // trowbridge_reitz.rs:61-74
		const V3 wom = -wo;
		const V3 h = normalised(wi + wom);
		if (dot(wi, hit.normal) < 0.0f || dot(h, wom) < 0.0f)
			return v3s(0.0f);
		const V3 f = tr_fresnel<F>(S, m, mat, hit, wom, wi, h);
		const float g = tr_g2(m.param, hit.normal, h, wom, wi);
		const float d = tr_d(m.param, dot(hit.normal, h));
		return f * g * d / (4.0f * fabsf(dot(wom, hit.normal)) * dot(wi, hit.normal));
	}
	return v3s(0.0f); // Emit::eval is unreachable!() in the reference
}

template <class F> __device__ __forceinline__ V3 mat_eval_over_pdf(const DevScene &S, uint32_t mat, const Hit &hit, V3 wo, V3 wi, int known = kMatRead, const DevPairScene &ps = kNoPairScene)
{
	const DevMaterial &m = mat_record(S, mat);
	const int type = mat_type_<F>(mat, known);
	if (type == 1) { // lambertian.rs:48-50
		const LambertRec L = lambert_record<F>(S, m, mat, wo, hit.point, ps);
		return L.colour * L.albedo;
	}
	if (!F::cmat)
		return v3s(0.0f); // unreachable without Reflect/Refract/TrowbridgeReitz
	if (type == 2) { // trowbridge_reitz.rs:75-87
		const V3 wom = -wo;
		const V3 h = normalised(wi + wom);
		if (dot(wom, h) < 0.0f || dot(wi, hit.normal) < 0.0f)
			return v3s(0.0f);
		const V3 f = tr_fresnel<F>(S, m, mat, hit, wom, wi, h);
		const float g = tr_g2(m.param, hit.normal, h, wom, wi);
		return f * g / tr_g1(m.param, hit.normal, h, wom);
	}
	// trait default: eval / scattering_pdf  (rt_core/src/material.rs:24-26)
	return mat_eval<F>(S, mat, hit, wo, wi) / mat_scattering_pdf<F>(S, mat, hit, wo, wi);
}

template <class F> __device__ __forceinline__ V3 mat_get_emission(const DevScene &S, uint32_t mat, const Hit &hit, V3 wo)
{
	const DevMaterial &m = mat_record(S, mat);
	if (mat_handle_type(mat) == 0) { // emissive.rs:23-26
		const V3 point = offset_ray(hit.point, hit.normal, hit.err_dot, true);
		return m.param * material_texture_colour<F>(S, m, mat, wo, point);
	}
	return v3s(0.0f);
}
// Emit::get_emission of the SKY's material when that record is known to be an Emit over a SolidColour or a Lerp
// (F::known_materials): strength, texture type and colours come from the kernel arguments (rt_types.h DevPairScene) and the
// colour is formed from SGPR operands.  Neither texture reads the point, so offset_ray (emissive.rs:24) has nothing to feed.
__device__ __forceinline__ V3 sky_emission_uniform(const DevPairScene &ps, V3 wo)
{
	const float param = ps.sky_param;
	const V3 c1 = v3(ps.sky_c1[0], ps.sky_c1[1], ps.sky_c1[2]);
	if (ps.sky_tex_type == 1) // SolidColour (wave-uniform branch)
		return param * c1;
	const V3 c2 = v3(ps.sky_c2[0], ps.sky_c2[1], ps.sky_c2[2]);
	const float tt = wo.z * 0.5f + 0.5f; // Lerp  textures/mod.rs:283-291
	return param * (c1 * tt + c2 * (1.0f - tt));
}
// get_emission of what a path ray hit: a primitive's material or the sky's
template <class F> __device__ __forceinline__ V3 emission_of_hit(const DevScene &S, const DevPairScene &ps, uint32_t mat, bool is_sky, const Hit &hit, V3 wo)
{
	if constexpr (F::known_materials) {
		if (!is_sky)
			return v3s(0.0f); // a Lambertian
		return sky_emission_uniform(ps, wo);
	} else {
		return mat_get_emission<F>(S, mat, hit, wo);
	}
}

// ---- light-sampling geometry of primitives ----
template <class F> __device__ __forceinline__ float prim_area(const PrimGeom &g) // sphere.rs:167-169, triangle.rs:226-230
{
	if (!F::tri || g.type == kPrimSphere)
		return 4.0f * kPi * g.p1.x * g.p1.x;
	return 0.5f * mag(cross(g.p1 - g.p0, g.p2 - g.p0));
}
// sample_visible_from_point  sphere.rs:118-154, triangle.rs:231-241,263-277
template <class F> __device__ inline V3 prim_sample_visible_from_point(const PrimGeom &g, V3 in_point, rt_rng &rng)
{
	if (!F::tri || g.type == kPrimSphere) {
		const V3 center = g.p0;
		const float radius = g.p1.x;
		const float distance_sq = mag_sq(in_point - center);
		V3 point;
		if (distance_sq <= radius * radius) { // Sphere::get_sample
			const float z = 1.0f - 2.0f * rt_rng_f32(&rng);
			const float a = sqrtf(fmax_(1.0f - z * z, 0.0f));
			const float b = 2.0f * kPi * rt_rng_f32(&rng);
			float sin_b, cos_b;
			lean_sincos(b, sin_b, cos_b);
			point = center + radius * v3(a * cos_b, a * sin_b, z);
		} else {
			// (max(1 - x, 0) of a float x: zero, NaN or at least 2^-24 -- inside the short square root's range, rt_lean.h;
			// the other arguments are tested)
			const float distance = sqrt_from_(distance_sq);
			const float sin_theta_max_sq = radius * radius / distance_sq;
			const float cost_theta_max = sqrt_unit_(fmax_(1.0f - sin_theta_max_sq, 0.0f));
			const float r1 = rt_rng_f32(&rng);
			const float cos_theta = (1.0f - r1) + r1 * cost_theta_max;
			const float sin_theta = sqrt_unit_(fmax_(1.0f - cos_theta * cos_theta, 0.0f));
			const float phi = 2.0f * rt_rng_f32(&rng) * kPi;
			const float ds = distance * cos_theta - sqrt_from_(fmax_(radius * radius - distance_sq * sin_theta * sin_theta, 0.0f));
			const float cos_alpha = (distance_sq + radius * radius - ds * ds) / (2.0f * distance * radius);
			const float sin_alpha = sqrt_unit_(fmax_(1.0f - cos_alpha * cos_alpha, 0.0f));
			const Coord cs = F::tri ? coord_from_z_zsel(normalised(in_point - center)) : coord_from_z(normalised(in_point - center));
			float sin_phi, cos_phi;
			lean_sincos(phi, sin_phi, cos_phi);
			const V3 vec = to_coord(cs, v3(sin_alpha * cos_phi, sin_alpha * sin_phi, cos_alpha));
			point = center + radius * vec;
		}
		return normalised(point - in_point);
	}
	const float su = sqrtf(rt_rng_f32(&rng));
	const float u0 = 1.0f - su;
	float r2 = rt_rng_f32(&rng);
	if (g.type == kPrimMeshTriangle) // MeshTriangle takes a second sqrt (triangle.rs:270)
		r2 = sqrtf(r2);
	const float u1 = su * r2;
	const V3 point = u0 * g.p0 + u1 * g.p1 + (1.0f - u0 - u1) * g.p2;
	return normalised(point - in_point);
}
// Primitive::scattering_pdf  sphere.rs:155-166, triangle.rs:242-244,278-280
template <class F> __device__ __forceinline__ float prim_scattering_pdf(const PrimGeom &g, V3 hit_point, V3 wi, const Hit &sampled_hit)
{
	if (!F::tri || g.type == kPrimSphere) {
		const float rsq = g.p1.x * g.p1.x;
		const float dsq = mag_sq(hit_point - g.p0);
		if (dsq <= rsq)
			return mag_sq(sampled_hit.point - hit_point) / (fabsf(dot(wi, sampled_hit.normal)) * prim_area<F>(g));
		const float sin_theta_max_sq = rsq / dsq;
		const float cos_theta_max = sqrt_unit_(fmax_(1.0f - sin_theta_max_sq, 0.0f));
		return 1.0f / (2.0f * kPi * (1.0f - cos_theta_max));
	}
	return mag_sq(sampled_hit.point - hit_point) / (fabsf(dot(sampled_hit.normal, wi)) * prim_area<F>(g));
}

} // namespace rt
