/*
 * rt_detmath.h -- the deterministic arithmetic CONTRACT of the rt_hip boundary.
 *
 * The reference renderer (nonl4331/raytracing-rust) takes every random number from an
 * OS-seeded generator (`thread_rng()` / `SmallRng::from_rng(thread_rng())`:
 * crates/implementations/src/samplers/random_sampler.rs:48,
 * crates/implementations/src/utility/mod.rs:41-44) and every transcendental from the
 * platform libm through Rust `std` (f32::{sin,cos,acos,atan2,tan,powf}).  Neither is
 * pinned by the reference (no Cargo.lock, no seed), so an `rt_render(seed=…)` boundary has
 * to DEFINE both before "same scene, same seed => same pixels" can mean anything.  This
 * header is that definition.  It plays the role the `rand` crate and libm play for the
 * reference: a third-party arithmetic dependency shared by every implementation of the
 * boundary (the HIP product in raytracing-rust_amd/csrc and the CPU checker in oracle/).
 * It contains NO path-tracing logic; everything the reference itself implements is
 * written twice, independently, on the two sides.
 *
 * Rules that make host (gcc, x86-64 SSE) and device (hipcc, gfx950) agree bit for bit:
 *   - only IEEE-754 binary32 +,-,*,/,sqrt and fmaf (all correctly rounded on both sides);
 *   - compile every translation unit that includes this file with -ffp-contract=off and
 *     without fast-math; fused operations appear ONLY as explicit fmaf() calls;
 *   - no libm transcendental is called from here.
 *
 * One header, three compilers: gcc (C99), g++ (C++17), hipcc (host + device).
 */
#ifndef RT_DETMATH_H
#define RT_DETMATH_H

#include <stdint.h>
#include <math.h>
#include <string.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ __forceinline__
#elif defined(__cplusplus)
#define RT_HD static inline
#else
#define RT_HD static inline
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

/* ------------------------------------------------------------------------------------ */
/* bit casts                                                                             */
/* ------------------------------------------------------------------------------------ */
RT_HD uint32_t rt_f32_bits(float f)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __float_as_uint(f);
#else
	uint32_t u;
	memcpy(&u, &f, 4);
	return u;
#endif
}
RT_HD float rt_bits_f32(uint32_t u)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __uint_as_float(u);
#else
	float f;
	memcpy(&f, &u, 4);
	return f;
#endif
}

/* ------------------------------------------------------------------------------------ */
/* Random stream.                                                                        */
/*                                                                                       */
/* One independent stream per (seed, pixel index, sample index).  The stream is seeded   */
/* by one Philox4x32-10 block (Salmon et al., SC'11; counter = pixel/sample, key = seed) */
/* and advanced by xoshiro128++ (Blackman & Vigna) -- 32-bit words only, no multiplies   */
/* on the per-draw path, 4 registers of state per lane.  Draw k of a path is therefore   */
/* a pure function of (seed, pixel, sample, k), whatever lane or thread evaluates it.    */
/* ------------------------------------------------------------------------------------ */
typedef struct rt_rng {
	uint32_t s0, s1, s2, s3;
} rt_rng;

RT_HD uint32_t rt_mulhi_u32(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __umulhi(a, b);
#else
	return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
#endif
}

RT_HD void rt_philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RT_PHILOX_ROLLED)
	/* (unrolled on the device: left to itself hipcc keeps a five-trip loop whose counter, compare and branch are scalar
	 * instructions that take vector issue slots from the SIMD -- profiles/r04_valu_issue.txt) */
	_Pragma("unroll")
#endif
	for (int round = 0; round < 10; ++round) {
		/* the full 64-bit products: on gfx950 ONE v_mad_u64_u32 each instead of a v_mul_hi_u32 / v_mul_lo_u32 pair */
		const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c[0];
		const uint64_t p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c[2];
		const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
		const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
		const uint32_t n0 = hi1 ^ c[1] ^ k0;
		const uint32_t n1 = lo1;
		const uint32_t n2 = hi0 ^ c[3] ^ k1;
		const uint32_t n3 = lo0;
		c[0] = n0;
		c[1] = n1;
		c[2] = n2;
		c[3] = n3;
		k0 += 0x9E3779B9u;
		k1 += 0xBB67AE85u;
	}
}

RT_HD void rt_rng_seed(rt_rng *r, uint64_t seed, uint64_t pixel, uint64_t sample)
{
	uint32_t c[4];
	c[0] = (uint32_t)pixel;
	c[1] = (uint32_t)(pixel >> 32);
	c[2] = (uint32_t)sample;
	c[3] = (uint32_t)(sample >> 32);
	rt_philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
	if ((c[0] | c[1] | c[2] | c[3]) == 0u)
		c[0] = 1u; /* xoshiro must not start from the all-zero state */
	r->s0 = c[0];
	r->s1 = c[1];
	r->s2 = c[2];
	r->s3 = c[3];
}

RT_HD uint32_t rt_rotl32(uint32_t x, int k)
{
	return (x << k) | (x >> (32 - k));
}

/* xoshiro128++ 1.0 */
RT_HD uint32_t rt_rng_u32(rt_rng *r)
{
	const uint32_t result = rt_rotl32(r->s0 + r->s3, 7) + r->s0;
	const uint32_t t = r->s1 << 9;
	r->s2 ^= r->s0;
	r->s3 ^= r->s1;
	r->s1 ^= r->s2;
	r->s0 ^= r->s3;
	r->s2 ^= t;
	r->s3 = rt_rotl32(r->s3, 11);
	return result;
}

/* rand 0.8 `Standard` for f32 (what `rng.gen::<f32>()` / `random_float()` produce):
 * 24 random bits times 2^-24, in [0,1).  Call sites: utility/mod.rs:41-44,
 * statistics/bxdfs/lambertian.rs:6-8, integrators/mis.rs:75, statistics/distributions.rs:52. */
RT_HD float rt_rng_f32(rt_rng *r)
{
	return (float)(rt_rng_u32(r) >> 8) * 5.9604644775390625e-08f;
}

/* rand 0.8 `UniformFloat<f32>::sample_single` (what `rng.gen_range(low..high)` produces):
 * 23 random mantissa bits -> [1,2) -> minus 1 -> times (high-low) plus low.  The crate's
 * retry loop (taken only when rounding lands on `high`) cannot trigger for the two ranges
 * the reference uses, 0.0..1.0 (samplers/random_sampler.rs:55,58) and -1.0..1.0
 * (utility/mod.rs:19-21): the largest result is high - 2^-23*(high-low) < high. */
RT_HD float rt_rng_range_f32(rt_rng *r, float low, float high)
{
	const float scale = high - low;
	const float value1_2 = rt_bits_f32((rt_rng_u32(r) >> 9) | 0x3F800000u);
	const float value0_1 = value1_2 - 1.0f;
	return value0_1 * scale + low;
}

/* rand 0.8 `UniformInt::sample_single` (what `rng.gen_range(0..n)` produces), restated on
 * 32-bit words: widening multiply with the crate's `zone` rejection.  Call sites:
 * integrators/mis.rs:140-142,147-149.  Requires 1 <= n < 2^31. */
RT_HD uint32_t rt_rng_below(rt_rng *r, uint32_t n)
{
	if (n == 0u)
		return 0u;
#if defined(__HIP_DEVICE_COMPILE__)
	/* n = 2^k (one light, or none, besides the sky: the reference's scenes): the crate's widening multiply is two shifts --
	 * hi = v >> (32 - k), lo = v << k, zone = 2^31 - 1, so a draw is accepted iff bit (31 - k) of v is clear (rand 0.8's
	 * sample_single rejects HALF of all draws for a power-of-two range, and a wave loops until its unluckiest lane accepts:
	 * about seven trips, each without the two half-rate 32-bit multiplies now).  Same draws, same results. */
	if ((n & (n - 1u)) == 0u) {
		const uint32_t k = 31u - (uint32_t)__builtin_clz(n);
		for (;;) {
			const uint32_t v = rt_rng_u32(r);
			if (((v << k) & 0x80000000u) == 0u)
				return k == 0u ? 0u : v >> (32u - k);
		}
	}
#endif
	uint32_t lz = 0;
	for (uint32_t m = n; (m & 0x80000000u) == 0u; m <<= 1)
		++lz;
	const uint32_t zone = (n << lz) - 1u;
	for (;;) {
		const uint32_t v = rt_rng_u32(r);
		const uint32_t hi = rt_mulhi_u32(v, n);
		const uint32_t lo = v * n;
		if (lo <= zone)
			return hi;
	}
}

/* ------------------------------------------------------------------------------------ */
/* Deterministic f32 elementary functions (replacement for the platform libm the        */
/* reference reaches through Rust std).  Accuracy is tested against glibc in            */
/* tests/test_detmath.py (sin, cos <= 2 ulp, acos <= 2.5, atan2 <= 4, tan <= 4).              */
/* ------------------------------------------------------------------------------------ */
#define RT_PI 3.14159274101257324219f      /* std::f32::consts::PI  */
#define RT_TAU 6.28318548202514648438f     /* std::f32::consts::TAU */
#define RT_FRAC_PI_2 1.57079637050628662109f
#define RT_FRAC_PI_4 0.785398185253143310547f
#define RT_F32_EPSILON 1.1920928955078125e-07f /* f32::EPSILON */

/* x - k*pi/2 with a three-term Cody-Waite split of pi/2 (24+24+24 bits) and fmaf: exact to ~1e-15
 * for |x| up to 2^22.  Beyond that (never reached by the path's own arguments 2*pi*r, pi*v and
 * 10*coordinate) x is first folded into [0, 2*pi) in IEEE double arithmetic -- deterministic on
 * both sides, accuracy |x| * 2^-53 -- and from 2^50 on the result is NaN (no Payne-Hanek). */
RT_HD float rt_reduce_pio2(float x, int *quadrant)
{
	if (fabsf(x) > 4194304.0f) {
		const double two_pi = 6.283185307179586476925286766559;
		const double xd = (double)x;
		x = (float)(xd - two_pi * floor(xd / two_pi));
	}
	const float k = rintf(x * 0.636619746685028076172f);
	float r = fmaf(-k, 1.57079637050628662109f, x);
	r = fmaf(-k, -4.37113882867379282984e-08f, r);
	r = fmaf(-k, -1.71512451000588185479e-15f, r);
	*quadrant = (int)k; /* |k| <= 2^22 * 2/pi: always representable */
	return r;
}

/* minimax kernels on [-pi/4, pi/4] (Cephes single-precision coefficient sets) */
RT_HD float rt_sin_kernel(float r)
{
	const float z = r * r;
	float p = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
	p = fmaf(p, z, -1.6666654611e-1f);
	return fmaf(p * z, r, r);
}
RT_HD float rt_cos_kernel(float r)
{
	const float z = r * r;
	float p = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
	p = fmaf(p, z, 4.166664568298827e-2f);
	return fmaf(p * z, z, fmaf(-0.5f, z, 1.0f));
}

RT_HD float rt_sinf(float x)
{
	if (!(fabsf(x) < 1.125899906842624e15f))
		return (x - x) / (x - x); /* inf, nan, |x| >= 2^50 -> nan */
	int q;
	const float r = rt_reduce_pio2(x, &q);
	const float s = (q & 1) ? rt_cos_kernel(r) : rt_sin_kernel(r);
	return (q & 2) ? -s : s;
}
RT_HD float rt_cosf(float x)
{
	if (!(fabsf(x) < 1.125899906842624e15f))
		return (x - x) / (x - x);
	int q;
	const float r = rt_reduce_pio2(x, &q);
	const float c = (q & 1) ? rt_sin_kernel(r) : rt_cos_kernel(r);
	return ((q + 1) & 2) ? -c : c;
}
RT_HD float rt_tanf(float x)
{
	return rt_sinf(x) / rt_cosf(x);
}

/* asin on [-0.5, 0.5] (Cephes asinf polynomial) */
RT_HD float rt_asin_kernel(float x)
{
	const float z = x * x;
	float p = fmaf(4.2163199048e-2f, z, 2.4181311049e-2f);
	p = fmaf(p, z, 4.5470025998e-2f);
	p = fmaf(p, z, 7.4953002686e-2f);
	p = fmaf(p, z, 1.6666752422e-1f);
	return fmaf(p * z, x, x);
}
/* acos; |x| > 1 -> NaN (as f32::acos) */
RT_HD float rt_acosf(float x)
{
	if (x > 0.5f)
		return 2.0f * rt_asin_kernel(sqrtf(0.5f * (1.0f - x)));
	if (x < -0.5f)
		return RT_PI - 2.0f * rt_asin_kernel(sqrtf(0.5f * (1.0f + x)));
	return RT_FRAC_PI_2 - rt_asin_kernel(x);
}

/* atan on [0, 1]: one octant split at tan(pi/8), Cephes atanf polynomial */
RT_HD float rt_atan_unit(float t)
{
	float base = 0.0f;
	if (t > 0.414213567972183227539f) {
		t = (t - 1.0f) / (t + 1.0f);
		base = RT_FRAC_PI_4;
	}
	const float z = t * t;
	float p = fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
	p = fmaf(p, z, 1.99777106478e-1f);
	p = fmaf(p, z, -3.33329491539e-1f);
	return base + fmaf(p * z, t, t);
}
/* atan2 with the IEEE special cases f32::atan2 has */
RT_HD float rt_atan2f(float y, float x)
{
	if (x != x || y != y)
		return x + y;
	const float ax = fabsf(x), ay = fabsf(y);
	float r;
	if (ax == 0.0f && ay == 0.0f) {
		r = 0.0f;
	} else if (ax == ay) {
		r = RT_FRAC_PI_4; /* also inf/inf */
	} else if (ay > ax) {
		r = RT_FRAC_PI_2 - rt_atan_unit(ax / ay);
	} else {
		r = rt_atan_unit(ay / ax);
	}
	if (rt_f32_bits(x) & 0x80000000u)
		r = RT_PI - r;
	return (rt_f32_bits(y) & 0x80000000u) ? -r : r;
}

/* x^5, the only power on the render path (materials/refract.rs:59-61, `powf(5.0)`) */
RT_HD float rt_pow5f(float x)
{
	const float x2 = x * x;
	return (x2 * x2) * x;
}

/* f32::to_radians: self * (PI / 180) evaluated in f32 */
/* x^y for the output stage's `val.powf(1.0 / gamma)` (crates/output/src/lib.rs:92-95).  Computed in binary64 from IEEE
 * +, -, *, / and explicit fma only -- log2 by the atanh series on a mantissa in [sqrt(1/2), sqrt(2)), exp2 by a Taylor
 * polynomial on [-1/2, 1/2] -- so host and device agree bit for bit; the f64 result carries ~1e-15 relative error before the
 * single rounding to f32 (within 1 ulp of libm's powf; tests/test_detmath.py).  Special cases follow powf. */
RT_HD double rt_bits_f64(uint64_t u)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __longlong_as_double((long long)u);
#else
	double d;
	memcpy(&d, &u, sizeof d);
	return d;
#endif
}
RT_HD float rt_powf(float x, float y)
{
	if (y == 0.0f || x == 1.0f)
		return 1.0f;
	if (x != x || y != y)
		return x + y; /* NaN */
	/* negative base (and -0): powf is defined for INTEGER exponents -- the sign of the result is the parity of y -- and NaN
	 * otherwise.  1 / gamma is an integer for gamma = 1, 0.5, 0.25, ...; every float of magnitude >= 2^24 is an even integer. */
	int negate = 0;
	if (rt_f32_bits(x) & 0x80000000u) {
		const float ay = fabsf(y);
		const int y_is_int = (ay >= 16777216.0f) || (ay == floorf(ay));
		const int y_is_odd = (ay < 16777216.0f) && y_is_int && (((uint32_t)ay & 1u) != 0u);
		if (x == -INFINITY)
			return y > 0.0f ? (y_is_odd ? -INFINITY : INFINITY) : (y_is_odd ? -0.0f : 0.0f);
		if (y == INFINITY || y == -INFINITY) { /* pow(x, +-inf) looks at |x| only */
			if (x == -1.0f)
				return 1.0f;
			return ((-x > 1.0f) == (y > 0.0f)) ? INFINITY : 0.0f;
		}
		if (x == 0.0f)
			return y > 0.0f ? (y_is_odd ? -0.0f : 0.0f) : (y_is_odd ? -INFINITY : INFINITY);
		if (!y_is_int)
			return rt_bits_f32(0x7FC00000u);
		negate = y_is_odd;
		x = -x;
	}
	if (x == 0.0f)
		return y > 0.0f ? 0.0f : INFINITY;
	if (x == INFINITY)
		return y > 0.0f ? INFINITY : 0.0f;
	if (y == INFINITY || y == -INFINITY)
		return ((x > 1.0f) == (y > 0.0f)) ? INFINITY : 0.0f;
	if (x == 1.0f)
		return negate ? -1.0f : 1.0f;
	/* x = m * 2^e, m in [sqrt(1/2), sqrt(2)) */
	uint32_t bits = rt_f32_bits(x);
	int e = 0;
	if ((bits >> 23) == 0u) { /* subnormal: scale by 2^24 (exact) */
		bits = rt_f32_bits(x * 16777216.0f);
		e = -24;
	}
	e += (int)(bits >> 23) - 127;
	float m = rt_bits_f32((bits & 0x007FFFFFu) | 0x3F800000u); /* [1, 2) */
	if (m > 1.41421353816986083984f) {
		m *= 0.5f;
		e += 1;
	}
	const double md = (double)m;
	const double f = (md - 1.0) / (md + 1.0); /* |f| <= 0.1716 */
	const double f2 = f * f;
	double p = 1.0 / 25.0;
	p = fma(p, f2, 1.0 / 23.0);
	p = fma(p, f2, 1.0 / 21.0);
	p = fma(p, f2, 1.0 / 19.0);
	p = fma(p, f2, 1.0 / 17.0);
	p = fma(p, f2, 1.0 / 15.0);
	p = fma(p, f2, 1.0 / 13.0);
	p = fma(p, f2, 1.0 / 11.0);
	p = fma(p, f2, 1.0 / 9.0);
	p = fma(p, f2, 1.0 / 7.0);
	p = fma(p, f2, 1.0 / 5.0);
	p = fma(p, f2, 1.0 / 3.0);
	p = fma(p, f2, 1.0);
	const double ln_m = 2.0 * f * p;
	const double log2_x = (double)e + ln_m * 1.4426950408889634074; /* log2(e) */
	double z = (double)y * log2_x;
	if (z > 200.0)
		z = 200.0; /* beyond f32 range either way: the final conversion gives inf / 0 */
	if (z < -200.0)
		z = -200.0;
	const double n = floor(z + 0.5);
	const double r = (z - n) * 0.69314718055994530942; /* |r| <= 0.3466 */
	double q = 1.0 / 87178291200.0; /* 1/14! */
	q = fma(q, r, 1.0 / 6227020800.0);
	q = fma(q, r, 1.0 / 479001600.0);
	q = fma(q, r, 1.0 / 39916800.0);
	q = fma(q, r, 1.0 / 3628800.0);
	q = fma(q, r, 1.0 / 362880.0);
	q = fma(q, r, 1.0 / 40320.0);
	q = fma(q, r, 1.0 / 5040.0);
	q = fma(q, r, 1.0 / 720.0);
	q = fma(q, r, 1.0 / 120.0);
	q = fma(q, r, 1.0 / 24.0);
	q = fma(q, r, 1.0 / 6.0);
	q = fma(q, r, 0.5);
	q = fma(q, r, 1.0);
	q = fma(q, r, 1.0);
	const double scale = rt_bits_f64((uint64_t)((long long)n + 1023) << 52); /* 2^n, n in [-200, 200] */
	const float result = (float)(q * scale);
	return negate ? -result : result;
}

/* save_data_to_image's pixel conversion (crates/output/src/lib.rs:92-95): (val.powf(1.0 / gamma) * 255.999) as u8.
 * Rust's float -> u8 `as` saturates and maps NaN to 0. */
RT_HD uint8_t rt_quantise_u8(float val, float inv_gamma)
{
	const float v = rt_powf(val, inv_gamma) * 255.999f;
	return !(v > 0.0f) ? (uint8_t)0 : (v >= 255.0f ? (uint8_t)255 : (uint8_t)v);
}

RT_HD float rt_to_radians(float deg)
{
	return deg * (RT_PI / 180.0f);
}

#endif /* RT_DETMATH_H */
