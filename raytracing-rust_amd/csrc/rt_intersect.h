// rt_intersect.h -- device rays, the AABB slab test, ray/sphere and ray/triangle intersection and
// BVH traversal for gfx950.
//
// What is computed follows the reference exactly (same predicates, same operation order):
//   Ray::new                      rt_core/src/ray.rs:13-46
//   AABB::does_int                acceleration/aabb.rs:22-57
//   Sphere::get_int               primitives/sphere.rs:34-105
//   triangle_intersection         primitives/triangle.rs:105-216
//   Bvh::check_hit / check_hit_index   acceleration/mod.rs:226-298
// HOW the tree is walked does not: the reference collects every AABB-hit leaf breadth-first into a
// heap Vec and then tests them all (mod.rs:199-224).  Here each lane walks depth-first with a
// per-lane stack in LDS, two child boxes per 64-byte node fetch, near child first, optional
// t-pruning.  The reference's result is "smallest t > 0; among equal t the primitive that comes
// first in BFS-leaf order" (strict `<` at mod.rs:282); the same winner is selected here by
// comparing (t, prim_rank) where prim_rank is the primitive's position in that BFS order.
#pragma once

#include "rt_types.h"
#include "rt_lean.h"

namespace rt {

struct Ray {
	V3 o, d, inv, shear;
};

// Ray::new  ray.rs:13-46 (time is carried nowhere: it is never read on the render path)
// Six to nine divisions and a square root per ray.  When every component of the direction is non-zero with magnitude in
// [2^-60, 2^20] and the squared length lies in [2^-40, 2^40] (every ray a render produces, bar the exactly axis-parallel
// ones) all operands below are tame in the sense of rt_lean.h -- length in [2^-20, 2^20], normalised components in
// [2^-81, 1], numerators 1 or such components -- so the short forms return the bits of the plain operators; any other
// direction takes the plain operators.
template <class F> __device__ __forceinline__ Ray ray_new(V3 origin, V3 direction)
{
	Ray r;
	r.o = origin;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RT_NO_LEAN)
	const float m2 = dot(direction, direction);
	const float smallest = fminf(fminf(fabsf(direction.x), fabsf(direction.y)), fabsf(direction.z));
	// (deciding for the WAVE instead -- one untame lane sends all through the plain operators, a scalar branch instead of an
	// exec-mask region -- measured 0.3 % slower on configs 2 and 3: profiles/r04j_sums_coord_waveguard_ab.log)
	if (__builtin_expect(smallest >= 0x1p-60f && m2 >= 0x1p-40f && m2 <= 0x1p40f, 1)) {
		direction = lean_div3(direction, lean_sqrt(m2));
		r.d = direction;
		r.inv = v3(lean_inv(direction.x), lean_inv(direction.y), lean_inv(direction.z));
		if (F::tri) {
			const float ax = fabsf(direction.x), ay = fabsf(direction.y), az = fabsf(direction.z);
			const bool swap = (ax > ay && ax > az) || (ay > az);
			const float sx = swap ? direction.z : direction.x;
			const float sz = swap ? direction.x : direction.z;
			const float rz = lean_rcp_refined(sz);
			r.shear = v3(lean_div_core(-sx, sz, rz), lean_div_core(-direction.y, sz, rz), lean_div_core(1.0f, sz, rz));
		} else {
			r.shear = v3s(0.0f);
		}
		return r;
	}
#endif
	direction = direction / mag(direction);
	r.d = direction;
	r.inv = v3(1.0f / direction.x, 1.0f / direction.y, 1.0f / direction.z);
	if (F::tri) {
		const float ax = fabsf(direction.x), ay = fabsf(direction.y), az = fabsf(direction.z);
		// max_axis 0 (x dominant) and 1 (y dominant) BOTH swap x<->z (ray.rs:26-33)
		const bool swap = (ax > ay && ax > az) || (ay > az);
		const float sx = swap ? direction.z : direction.x;
		const float sz = swap ? direction.x : direction.z;
		r.shear = v3(-sx / sz, -direction.y / sz, 1.0f / sz);
	} else {
		r.shear = v3s(0.0f); // only triangle_intersection reads the shear (triangle.rs:113-123)
	}
	return r;
}
__device__ __forceinline__ bool ray_swaps_xz(const Ray &r) // Axis::get_max_abs_axis + swap_z  primitives/mod.rs:62-82
{
	const float ax = fabsf(r.d.x), ay = fabsf(r.d.y), az = fabsf(r.d.z);
	return (ax > ay && ax > az) || (ay > az);
}

// AABB::does_int  aabb.rs:22-57.  Returns the reference's predicate; tmin_out is the entry
// distance (used only for ordering / pruning).
__device__ __forceinline__ bool aabb_does_int(const float bmin[3], const float bmax[3], const Ray &r, float &tmin_out)
{
	constexpr float widen = 1.0f + 2.0f * gamma_n(3);
	float t1 = (bmin[0] - r.o.x) * r.inv.x;
	float t2 = (bmax[0] - r.o.x) * r.inv.x;
	if (t1 > t2) { const float s = t1; t1 = t2; t2 = s; }
	t2 *= widen;
	float tmin = fmin_(t1, t2);
	float tmax = fmax_(t1, t2);

	t1 = (bmin[1] - r.o.y) * r.inv.y;
	t2 = (bmax[1] - r.o.y) * r.inv.y;
	if (t1 > t2) { const float s = t1; t1 = t2; t2 = s; }
	t2 *= widen;
	tmin = fmax_(tmin, fmin_(t1, t2));
	tmax = fmin_(tmax, fmax_(t1, t2));

	t1 = (bmin[2] - r.o.z) * r.inv.z;
	t2 = (bmax[2] - r.o.z) * r.inv.z;
	if (t1 > t2) { const float s = t1; t1 = t2; t2 = s; }
	t2 *= widen;
	tmin = fmax_(tmin, fmin_(t1, t2));
	tmax = fmin_(tmax, fmax_(t1, t2));

	tmin_out = tmin;
	return tmax > fmax_(tmin, 0.0f);
}

// sqrtf; the short form of rt_lean.h where the argument is in its range (>= 2^-96; zero, negatives and NaN too), the plain
// operator for the tiny positives in between
__device__ __forceinline__ float sqrt_from_(float x)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RT_NO_LEAN)
	if (__builtin_expect(!(x < kLeanSqrtMin) || !(x > 0.0f), 1))
		return lean_sqrt(x);
#endif
	return sqrtf(x);
}

// Sphere::get_int up to the choice of t  sphere.rs:34-77
__device__ __forceinline__ bool sphere_t(V3 center, float radius, const Ray &r, float &t)
{
	const V3 deltap = center - r.o;
	const float ddp = dot(r.d, deltap);
	const float deltapdot = dot(deltap, deltap);
	const V3 remedy_term = deltap - ddp * r.d;
	const float discriminant = radius * radius - dot(remedy_term, remedy_term);
	if (!(discriminant > 0.0f))
		return false;
	const float sqrt_val = sqrtf(discriminant); // (the tested short form, sqrt_from_, measured no gain here: its branch costs what it saves)
	const float q = ddp > 0.0f ? ddp + sqrt_val : ddp - sqrt_val;
	float t0 = q;
	float t1 = (deltapdot - radius * radius) / q;
	if (t1 < t0) { const float s = t0; t0 = t1; t1 = s; }
	if (t0 > 0.0f) {
		t = t0;
		return true;
	}
	if (t1 <= 0.0f)
		return false;
	t = t1;
	return true;
}

// triangle_intersection up to the barycentrics and t  triangle.rs:105-177
__device__ __forceinline__ bool triangle_t(V3 P0, V3 P1, V3 P2, const Ray &r, float &t, float &b0, float &b1, float &b2)
{
	V3 p0t = P0 - r.o;
	V3 p1t = P1 - r.o;
	V3 p2t = P2 - r.o;
	if (ray_swaps_xz(r)) {
		float s;
		s = p0t.x; p0t.x = p0t.z; p0t.z = s;
		s = p1t.x; p1t.x = p1t.z; p1t.z = s;
		s = p2t.x; p2t.x = p2t.z; p2t.z = s;
	}
	p0t.x += r.shear.x * p0t.z;
	p0t.y += r.shear.y * p0t.z;
	p1t.x += r.shear.x * p1t.z;
	p1t.y += r.shear.y * p1t.z;
	p2t.x += r.shear.x * p2t.z;
	p2t.y += r.shear.y * p2t.z;

	float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
	float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
	float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
	if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) { // f64 recompute  triangle.rs:128-132
		e0 = (float)((double)p1t.x * (double)p2t.y - (double)p1t.y * (double)p2t.x);
		e1 = (float)((double)p2t.x * (double)p0t.y - (double)p2t.y * (double)p0t.x);
		e2 = (float)((double)p0t.x * (double)p1t.y - (double)p0t.y * (double)p1t.x);
	}
	if ((e0 < 0.0f || e1 < 0.0f || e2 < 0.0f) && (e0 > 0.0f || e1 > 0.0f || e2 > 0.0f))
		return false;
	const float det = e0 + e1 + e2;
	if (det == 0.0f)
		return false;

	p0t = p0t * r.shear.z;
	p1t = p1t * r.shear.z;
	p2t = p2t * r.shear.z;

	const float t_scaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
	if ((det < 0.0f && t_scaled >= 0.0f) || (det > 0.0f && t_scaled <= 0.0f))
		return false;

	const float inv_det = 1.0f / det;
	b0 = e0 * inv_det;
	b1 = e1 * inv_det;
	b2 = e2 * inv_det;
	t = inv_det * t_scaled;

	const float max_z_t = component_max(v3(fabsf(p0t.z), fabsf(p1t.z), fabsf(p2t.z)));
	const float delta_z = gamma_n(3) * max_z_t;
	const float max_x_t = component_max(v3(fabsf(p0t.x), fabsf(p1t.x), fabsf(p2t.x)));
	const float max_y_t = component_max(v3(fabsf(p0t.y), fabsf(p1t.y), fabsf(p2t.y)));
	const float delta_x = gamma_n(5) * (max_x_t + max_z_t);
	const float delta_y = gamma_n(5) * (max_y_t + max_z_t);
	const float delta_e = 2.0f * (gamma_n(2) * max_x_t * max_y_t + delta_y * max_x_t + delta_x * max_y_t);
	const float max_e = component_max(v3(fabsf(e0), fabsf(e1), fabsf(e2)));
	const float delta_t = 3.0f * (gamma_n(3) * max_e * max_z_t + delta_e * max_z_t + delta_z * max_e) * fabsf(inv_det);
	if (t < delta_t)
		return false;
	return true;
}

// ---- primitive records ----
struct PrimGeom {
	uint32_t type, material;
	V3 p0, p1, p2; // sphere: p0 = centre, p1.x = radius
};
template <class F> __device__ __forceinline__ PrimGeom load_prim(const DevScene &S, uint32_t slot)
{
	const float4 *q = reinterpret_cast<const float4 *>(&S.prims[slot]);
	const float4 a = q[0];
	const float4 b = q[1];
	PrimGeom g;
	const uint32_t meta = __float_as_uint(a.w);
	g.type = meta & 3u;
	g.material = meta >> 2;
	g.p0 = v3(a.x, a.y, a.z);
	g.p1 = v3(b.x, b.y, b.z);
	if (F::tri && g.type != kPrimSphere) {
		const float4 c = q[2];
		g.p2 = v3(c.x, c.y, c.z);
	} else {
		g.p2 = v3(0.0f, 0.0f, 0.0f);
	}
	return g;
}
// Primitive::get_int reduced to (hit?, t)
template <class F> __device__ __forceinline__ bool prim_t(const PrimGeom &g, const Ray &r, float &t)
{
	if (!F::tri || g.type == kPrimSphere)
		return sphere_t(g.p0, g.p1.x, r, t);
	float b0, b1, b2;
	return triangle_t(g.p0, g.p1, g.p2, r, t, b0, b1, b2);
}

// ---- Hit  rt_core/src/primitive.rs:3-10 ----
struct Hit {
	float t;
	V3 point, error, normal;
	// dot(|normal|, error): all that offset_ray (utility/mod.rs:88-117) ever reads of `error`.  Formed once where the hit is
	// made, so the paths carry one register from bounce to bounce instead of three (same operands, same operation: same bits)
	float err_dot;
	float uvx, uvy;
	bool has_uv, out;
};

// utility::check_side  utility/mod.rs:6-13
__device__ __forceinline__ bool check_side(V3 &normal, V3 ray_direction)
{
	if (dot(normal, ray_direction) > 0.0f) {
		normal = -normal;
		return false;
	}
	return true;
}

// the rest of Sphere::get_int (sphere.rs:79-101) / triangle_intersection (triangle.rs:179-215)
// for the primitive that won the traversal.  `t` is the value the traversal computed for it; a
// sphere needs nothing else, a triangle re-runs its intersector for the barycentrics (same code and
// inputs, so the same t).
// The sphere's record, with the radius' reciprocal at hand (verified by the host: DevPrim::b[1], DevPairScene::inv_radius; 0 = not
// verified): the normal's three divisions become three plain operations each (rt_lean.h div_by_verified) when every component of
// p - c lies in [2^-60, 2^60] by magnitude -- the sequence's domain minus the zeros, whose sign it does not keep; otherwise the
// plain divisions.  NaN components pass the test or not as min / max ignore them: either way they come out NaN, which is all the
// filter of the sample asks.  Config 2, same box: 59.74 -> 57.9 ms (profiles/r04ai_sphere_normal_ab.log).
__device__ __forceinline__ void make_sphere_hit_by_reciprocal(V3 centre, float radius, float inv_radius, const Ray &r, float t, Hit &h) // sphere.rs:79-101
{
	const V3 point = r.o + r.d * t;
	const V3 x = point - centre;
	const float lo = fminf(fminf(fabsf(x.x), fabsf(x.y)), fabsf(x.z)), hi = fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), fabsf(x.z));
	V3 normal;
	if (inv_radius != 0.0f && lo >= 0x1p-60f && hi <= 0x1p60f) // (0: this radius was not verified -- never so under FeatPair's contract)
		normal = v3(div_by_verified(x.x, radius, inv_radius), div_by_verified(x.y, radius, inv_radius), div_by_verified(x.z, radius, inv_radius));
	else
		normal = x / radius;
	bool out = true;
	if (dot(normal, r.d) > 0.0f) {
		out = false;
		normal = -normal;
	}
	h.t = t;
	h.point = point;
	h.error = kEpsilon * v3s(1.0f);
	h.normal = normal;
	h.err_dot = dot(vabs(normal), h.error);
	h.uvx = h.uvy = 0.0f;
	h.has_uv = false; // no material overrides Scatter::requires_uv (rt_core/src/material.rs:8-10)
	h.out = out;
}
template <class F> __device__ __forceinline__ void make_hit(const DevScene &S, uint32_t slot, const Ray &r, float t_known, Hit &h, uint32_t &material,
                                                            const DevPairScene &ps = kNoPairScene)
{
	if constexpr (F::pair) {
		// the sphere's record from the kernel arguments (rt_types.h DevPairScene), selected by the slot; the "material" the
		// pair kernels carry along a path is the slot itself: all they ever want of it is the Lambertian record, which
		// lies next to the sphere's (rt_shade.h known_lambert)
		const bool first = slot == ps.slot0;
		const V3 centre = first ? v3(ps.sphere[0][0], ps.sphere[0][1], ps.sphere[0][2]) : v3(ps.sphere[1][0], ps.sphere[1][1], ps.sphere[1][2]);
		const float radius = first ? ps.sphere[0][3] : ps.sphere[1][3];
		material = slot;
		make_sphere_hit_by_reciprocal(centre, radius, first ? ps.inv_radius[0] : ps.inv_radius[1], r, t_known, h);
		return;
	}
	const PrimGeom g = load_prim<F>(S, slot);
	material = g.material;
	if (!F::tri || g.type == kPrimSphere) {
		make_sphere_hit_by_reciprocal(g.p0, g.p1.x, g.p1.y, r, t_known, h); // (DevPrim::b[1]: the radius' verified reciprocal, or 0)
		return;
	}
	float t = 0.0f, b0 = 0.0f, b1 = 0.0f, b2 = 0.0f;
	(void)triangle_t(g.p0, g.p1, g.p2, r, t, b0, b1, b2);
	const float4 *q = reinterpret_cast<const float4 *>(&S.shade[slot]);
	const float4 n0 = q[0], n1 = q[1], n2 = q[2];
	V3 normal = b0 * v3(n0.x, n0.y, n0.z) + b1 * v3(n1.x, n1.y, n1.z) + b2 * v3(n2.x, n2.y, n2.z);
	const bool out = check_side(normal, r.d);
	const float x_abs_sum = fabsf(b0 * g.p0.x) + fabsf(b1 * g.p1.x) + fabsf(b2 * g.p2.x);
	const float y_abs_sum = fabsf(b0 * g.p0.y) + fabsf(b1 * g.p1.y) + fabsf(b2 * g.p2.y);
	const float z_abs_sum = fabsf(b0 * g.p0.z) + fabsf(b1 * g.p1.z) + fabsf(b2 * g.p2.z);
	h.t = t;
	h.point = b0 * g.p0 + b1 * g.p1 + b2 * g.p2;
	h.error = gamma_n(7) * v3(x_abs_sum, y_abs_sum, z_abs_sum) + gamma_n(6) * v3(b2 * g.p2.x, b2 * g.p2.y, b2 * g.p2.z);
	h.normal = normal;
	h.err_dot = dot(vabs(normal), h.error);
	h.uvx = b0 * 0.0f + b1 * 1.0f + b2 * 1.0f; // b0*(0,0) + b1*(1,0) + b2*(1,1)
	h.uvy = b0 * 0.0f + b1 * 0.0f + b2 * 1.0f;
	h.has_uv = true;
	h.out = out;
}

// Sky::get_si  sky.rs:79-92
__device__ __forceinline__ void make_sky_hit(const DevScene &S, Hit &h, uint32_t &material)
{
	h.t = 0.0f;
	h.point = h.error = h.normal = v3s(0.0f);
	h.err_dot = 0.0f;
	h.uvx = h.uvy = 0.0f;
	h.has_uv = false;
	h.out = false;
	material = S.sky.material;
}
// The same for a feature set whose sky is KNOWN to carry an Emit material over a texture of the DIRECTION alone (FeatPair's
// contract, rt_types.h): the path ends there and nothing ever reads the zeroed position / normal / error of the sky "hit", so they
// are left as whatever the registers hold -- well-defined values the compiler need not produce (eight v_mov per shading arm of the
// config-2 loop otherwise).  The flags and t, which control flow could look at, stay defined.  (Any other feature set may meet a
// sky whose material scatters -- tests/test_gpu_parity.py "a lambertian as the sky's material" -- and then the zeros ARE the hit.)
template <class F> __device__ __forceinline__ void make_sky_hit_lean(const DevScene &S, Hit &h, uint32_t &material)
{
	if (F::ctex || !F::known_materials) {
		make_sky_hit(S, h, material);
		return;
	}
	h.t = 0.0f;
#if defined(__HIP_DEVICE_COMPILE__)
#define RT_ANY_VALUE(x) asm volatile("" : "=v"(x)) // "some value": an empty statement that defines its operand, no instruction (volatile: one
                                                   // definition per use, or the compiler shares one register and copies it)
#else
#define RT_ANY_VALUE(x) x = 0.0f
#endif
	RT_ANY_VALUE(h.point.x); RT_ANY_VALUE(h.point.y); RT_ANY_VALUE(h.point.z);
	RT_ANY_VALUE(h.error.x); RT_ANY_VALUE(h.error.y); RT_ANY_VALUE(h.error.z);
	RT_ANY_VALUE(h.normal.x); RT_ANY_VALUE(h.normal.y); RT_ANY_VALUE(h.normal.z);
	RT_ANY_VALUE(h.err_dot); RT_ANY_VALUE(h.uvx); RT_ANY_VALUE(h.uvy);
#undef RT_ANY_VALUE
	h.has_uv = false;
	h.out = false;
	material = S.sky.material;
}

// ---- traversal ----
// "while-while" walk (Aila & Laine, HPG 2009) for 64-lane wavefronts.  A lane descends through
// INNER nodes until the next thing it has to look at is a LEAF; only then does it fall out of the
// inner loop.  The wave reconverges there, so the (long) primitive intersection code runs with
// most lanes holding a leaf, instead of once per iteration for the one or two lanes that happen
// to be at a leaf.  Leaf references are ordinary 32-bit stack entries:
//   ref >= 0                inner node index
//   ref <  0 (bit 31 set)   leaf: bits 26-30 primitive count (1..31), bits 0-25 first slot;
//                           count field 0 = index into DevScene::big_leaves (first, count)
// Per-lane stack in LDS: entry e of lane l lives at stack[e * 64 + l]; `stk` already points at
// the lane's column, so consecutive lanes hit consecutive banks (conflict-free ds_read/ds_write_b32).
constexpr int kStackStride = 64;
// Where a lane's stack lives: the first `cap` entries in its LDS column, the rest (rare: the host sizes `cap` from the
// occupancy it wants, the worst case of a wide tree is far above what walks reach) in a global overflow area
// [thread][ovf_depth].  All fields are wave-uniform.  The overflow address is recomputed from the LDS column pointer when
// needed, so the common path carries no extra per-lane register.
struct StackMem {
	uint32_t cap;       // entries per lane in LDS
	uint32_t ovf_depth; // entries per lane in the overflow area (0: none, cap covers the worst case)
	uint32_t *ovf;
	uint32_t *region;   // LDS: start of this workgroup's stack region (wave w's columns start at region + w * cap * 64)
};
__device__ __forceinline__ uint32_t *stack_overflow_slot(const StackMem &M, const uint32_t *stk, int sp)
{
	const uint32_t idx = (uint32_t)(stk - M.region); // wave * cap * 64 + lane
	const uint32_t wave = idx / (M.cap * (uint32_t)kStackStride), lane = idx & 63u;
	const size_t thread = (size_t)blockIdx.x * blockDim.x + wave * 64u + lane;
	return M.ovf + thread * M.ovf_depth + ((uint32_t)sp - M.cap);
}
// OVF = false: the caller knows the LDS part covers the worst case (ovf_depth == 0): plain LDS accesses
template <bool OVF> __device__ __forceinline__ void stack_store(const StackMem &M, uint32_t *stk, int sp, uint32_t v)
{
	if (!OVF || sp < (int)M.cap)
		stk[sp * kStackStride] = v;
	else
		*stack_overflow_slot(M, stk, sp) = v;
}
template <bool OVF> __device__ __forceinline__ uint32_t stack_load(const StackMem &M, const uint32_t *stk, int sp)
{
	if (!OVF || sp < (int)M.cap)
		return stk[sp * kStackStride];
	return *stack_overflow_slot(M, stk, sp);
}
constexpr uint32_t kRefDone = 0x7FFFFFFFu; // sentinel: traversal finished
// Slack used by the pruned walk: a child whose entry distance exceeds the current best t by more
// than this cannot contain a primitive that beats or ties it (the slab test and the intersectors
// agree to a few ulp; 3e-5 relative is ~250 ulp).  The exhaustive walk (prune = false) needs none.
constexpr float kPruneSlack = 3.0e-5f;

struct NodeView {
	float c0min[3], c0max[3], c1min[3], c1max[3];
	uint32_t c0, c1;
};
__device__ __forceinline__ NodeView load_node(const DevScene &S, uint32_t node)
{
	const float4 *q = reinterpret_cast<const float4 *>(&S.nodes[node]);
	const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
	NodeView n;
	n.c0min[0] = q0.x; n.c0min[1] = q0.y; n.c0min[2] = q0.z;
	n.c0max[0] = q0.w; n.c0max[1] = q1.x; n.c0max[2] = q1.y;
	n.c1min[0] = q1.z; n.c1min[1] = q1.w; n.c1min[2] = q2.x;
	n.c1max[0] = q2.y; n.c1max[1] = q2.z; n.c1max[2] = q2.w;
	n.c0 = __float_as_uint(q3.x);
	n.c1 = __float_as_uint(q3.y);
	return n;
}
__device__ __forceinline__ float box_extent_l1(const float mn[3], const float mx[3])
{
	return (mx[0] - mn[0]) + (mx[1] - mn[1]) + (mx[2] - mn[2]);
}
__device__ __forceinline__ bool ref_is_leaf(uint32_t ref) { return (ref & 0x80000000u) != 0u; }
__device__ __forceinline__ void leaf_range(const DevScene &S, uint32_t ref, uint32_t &first, uint32_t &count)
{
	count = (ref >> 26) & 31u;
	first = ref & 0x03FFFFFFu;
	if (count == 0u) { // a leaf of 32+ primitives (coincident centroids, acceleration/mod.rs:129-134)
		const uint2 e = S.big_leaves[first];
		first = e.x;
		count = e.y;
	}
}
// can a box entered at distance t_entry still hold a primitive that beats or ties t_best?
__device__ __forceinline__ bool beyond(float t_entry, float t_best, const float mn[3], const float mx[3])
{
	return t_entry - kPruneSlack * (fabsf(t_entry) + fabsf(t_best) + box_extent_l1(mn, mx)) > t_best;
}

// One step of the inner-node descent shared by both walks: fetch `node`, test both child boxes
// with the reference's predicate, choose where to go next.  limit_valid/t_limit: prune children
// entered beyond t_limit (PRUNE only).
template <bool PRUNE, bool OVF>
__device__ __forceinline__ uint32_t descend_view(const NodeView &n, const StackMem &M, const Ray &r, uint32_t *stk, int &sp, bool limit_valid, float t_limit)
{
	float t0, t1;
	bool h0 = aabb_does_int(n.c0min, n.c0max, r, t0);
	bool h1 = aabb_does_int(n.c1min, n.c1max, r, t1);
	if (PRUNE && limit_valid) {
		if (h0 && beyond(t0, t_limit, n.c0min, n.c0max))
			h0 = false;
		if (h1 && beyond(t1, t_limit, n.c1min, n.c1max))
			h1 = false;
	}
	if (h0 && h1) {
		uint32_t near = n.c0, far = n.c1;
		if (PRUNE && t1 < t0) { // nearer child first, so the farther one can be pruned later
			near = n.c1;
			far = n.c0;
		}
		stack_store<OVF>(M, stk, sp, far);
		++sp;
		return near;
	}
	if (h0)
		return n.c0;
	if (h1)
		return n.c1;
	if (sp == 0)
		return kRefDone;
	--sp;
	return stack_load<OVF>(M, stk, sp);
}
template <bool PRUNE, bool OVF>
__device__ __forceinline__ uint32_t descend(const DevScene &S, const StackMem &M, const Ray &r, uint32_t node, uint32_t *stk, int &sp,
                                            bool limit_valid, float t_limit)
{
	return descend_view<PRUNE, OVF>(load_node(S, node), M, r, stk, sp, limit_valid, t_limit);
}

// ---- the wide walk ----
// WHY ANOTHER TREE RETURNS THE SAME HITS.  The reference tests a leaf's primitives iff the leaf's box and every
// ancestor's box pass AABB::does_int (mod.rs:199-224).  A node's box is the exact union of its primitives' boxes
// (mod.rs:105-108), so an ancestor's box contains the leaf's box exactly, and for a REGULAR ray -- origin and
// 1/direction finite and tame, which makes every (bound - origin) * inverse in the slab test a finite, non-NaN value --
// the slab test is monotone in the bounds: each IEEE operation in it is monotone, and on a box that is hit every axis
// has near = t1 and far = t2 * widen, so a box containing a hit box has tmin no larger and tmax no smaller and is hit
// too.  Hence for regular rays "leaf and all its ancestors hit" == "the leaf's OWN box is hit": the reference's
// candidate leaves are exactly the leaves whose own box passes does_int, whatever hierarchy leads to them.
// The wide tree (rt_types.h DevNodeQ4) keeps the reference's leaves, regroups the inner nodes four to a 64-byte record
// and stores CONSERVATIVE child boxes (8-bit grid, rounded outward).  Its walk therefore
//   1. reaches every leaf whose exact box the ray hits: a stored box contains the leaf's exact box as a set, and the
//      inner test below is padded by more than the rounding of both tests together (derivation at kWidePad);
//   2. tests a reached leaf's EXACT box (DevScene::leaf_box) with the reference's own predicate before touching its
//      primitives, so the primitives tested are exactly the reference's; the winner is chosen by the same (t, BFS rank)
//      rule.
// Irregular rays (a zero direction component gives an infinite inverse, and 0 * inf = NaN is IGNORED by Rust's min/max,
// which breaks monotonicity for flat boxes; huge components could overflow the padded test) take the two-child walk,
// which tests every ancestor as the reference does.  Scenes with non-finite or huge bounds get no wide tree at all.
__device__ __forceinline__ bool ray_is_regular(const Ray &r)
{
	const float mi = fmaxf(fmaxf(fabsf(r.inv.x), fabsf(r.inv.y)), fabsf(r.inv.z));
	const float mo = fmaxf(fmaxf(fabsf(r.o.x), fabsf(r.o.y)), fabsf(r.o.z));
	// 2^60: with bounds <= 2^60 (host check) every product in the walk stays far below FLT_MAX.  NaN anywhere -> false
	// (fmaxf drops a NaN operand, so test the components' sum for NaN as well)
	const float nan_probe = (r.inv.x + r.inv.y + r.inv.z) + (r.o.x + r.o.y + r.o.z);
	return mi <= 0x1p60f && mo <= 0x1p60f && nan_probe == nan_probe;
}

// Padding of the inner-node test, relative to M = max|b| + 255 max|a| (the node's distance and extent in ray
// parameter units; t = q * a + b with a = step / direction, b = (origin - ray origin) / direction per axis).
//   * this test: b carries two roundings, the fma one: |t_computed - t| <= 2^-23 |b| + 2^-24 |t| <= 2.4e-7 M;
//   * the reference's test on a box inside this one: tmax is inflated by (1 + 2 gamma(3)) and three roundings,
//     tmin deflated by two roundings: a box it accepts has tfar - tnear >= -1.1e-6 M in exact arithmetic and tfar > 0.
// So an accepted descendant implies tfar_computed - tnear_computed >= -1.6e-6 M and tfar_computed >= -2.4e-7 M here;
// the pad is 1e-5 M on both.  (A pad that is too large only costs visits; too small would lose hits.)
constexpr float kWidePad = 1.0e-5f;

// One step of the wide descent: fetch the 64-byte node (four dwordx4), test its (up to) four child boxes
// conservatively, prune, go to the nearest surviving child and push the others farthest first.
template <bool PRUNE, bool OVF>
__device__ __forceinline__ uint32_t descend4(const DevScene &S, const StackMem &M, const Ray &r, uint32_t node, uint32_t *stk, int &sp, bool limit_valid,
                                             float t_limit)
{
	const uint4 *q = reinterpret_cast<const uint4 *>(&S.nodes4[node]);
	const uint4 q0 = q[0], q1 = q[1], q2 = q[2]; // the compact node: three pieces (rt_types.h)
	const float sx = __uint_as_float((q0.w & 0xFFu) << 23), sy = __uint_as_float(((q0.w >> 8) & 0xFFu) << 23),
	            sz = __uint_as_float(((q0.w >> 16) & 0xFFu) << 23);
	const float ax = sx * r.inv.x, ay = sy * r.inv.y, az = sz * r.inv.z; // exact: the steps are powers of two
	const float bx = (__uint_as_float(q0.x) - r.o.x) * r.inv.x, by = (__uint_as_float(q0.y) - r.o.y) * r.inv.y,
	            bz = (__uint_as_float(q0.z) - r.o.z) * r.inv.z;
	// grid coordinates of the near and far planes per axis, by the sign of the direction; the four children's bytes
	// ride in one dword, so one select serves all four
	const bool negx = r.inv.x < 0.0f, negy = r.inv.y < 0.0f, negz = r.inv.z < 0.0f;
	const uint32_t nx = negx ? q1.w : q1.x, fx = negx ? q1.x : q1.w;
	const uint32_t ny = negy ? q2.x : q1.y, fy = negy ? q1.y : q2.x;
	const uint32_t nz = negz ? q2.y : q1.z, fz = negz ? q1.z : q2.y;
	const float big = fmaxf(fmaxf(fabsf(bx), fabsf(by)), fabsf(bz)) + 255.0f * fmaxf(fmaxf(fabsf(ax), fabsf(ay)), fabsf(az));
	const float pad = kWidePad * big;
	// prune: nothing inside a child is nearer than te = max(tnear - pad, 0); skip it when te exceeds the limit by the
	// slack of the two-child walk (kPruneSlack, with the node's whole extent standing in for the child's)
	const float ext = 255.0f * (sx + sy + sz);
	const float cut = (PRUNE && limit_valid) ? t_limit + kPruneSlack * (fabsf(t_limit) + ext) : INFINITY;
	uint32_t key[4]; // entry distance as ordered bits with the child slot in the two low bits; 0xFFFFFFFF = do not visit
#pragma unroll
	for (int c = 0; c < 4; ++c) {
		const int sh = 8 * c;
		const float tn = fmaxf(fmaxf(fmaf((float)((nx >> sh) & 0xFFu), ax, bx), fmaf((float)((ny >> sh) & 0xFFu), ay, by)),
		                       fmaf((float)((nz >> sh) & 0xFFu), az, bz));
		const float tf = fminf(fminf(fmaf((float)((fx >> sh) & 0xFFu), ax, bx), fmaf((float)((fy >> sh) & 0xFFu), ay, by)),
		                       fmaf((float)((fz >> sh) & 0xFFu), az, bz));
		const float te = fmaxf(tn - pad, 0.0f);
		bool h = (tf - tn >= -2.0f * pad) && (tf >= -pad) && (te - kPruneSlack * te <= cut);
		// An ABSENT child needs its own bit.  Its stored interval is inverted on every axis (tf - tn <= -255 max|a|), but the pad
		// grows with the node's distance: once the node's extent falls below ~2e-5 of it (a small, finely tessellated object far
		// away; a node of coincident primitives at any distance) the inverted interval passes the padded test, ref_of decodes a
		// phantom child and the walk re-enters a node it has been to.  Children 0 and 1 always exist (a wide node is a collapsed
		// reference node: two children at least; the host checks it), so only slots 2 and 3 are looked up: bits 28, 29 of q2.w.
		if (c >= 2)
			h = h && ((q2.w >> (26u + (uint32_t)c)) & 1u) != 0u;
		key[c] = h ? ((__float_as_uint(te) & ~3u) | (uint32_t)c) : 0xFFFFFFFFu;
	}
#define RT_CSWAP(a, b) { const uint32_t lo_ = min(key[a], key[b]); key[b] = max(key[a], key[b]); key[a] = lo_; }
	RT_CSWAP(0, 1) RT_CSWAP(2, 3) RT_CSWAP(0, 2) RT_CSWAP(1, 3) RT_CSWAP(1, 2)
#undef RT_CSWAP
	// child c of the compact node: inner node q2.z + offset or leaf index q2.w + offset, by bit c of the leaf mask
	auto ref_of = [&](uint32_t k) {
		const uint32_t c = k & 3u;
		const uint32_t delta = (q0.w >> (24u + 2u * c)) & 3u;
		const bool is_leaf = ((q2.z >> (26u + c)) & 1u) != 0u;
		return is_leaf ? (kLeafFlag | ((q2.w & kLeafSlotMask) + delta)) : ((q2.z & kLeafSlotMask) + delta);
	};
	if (key[0] == 0xFFFFFFFFu) { // nothing to descend into: back to the nearest pending sibling
		if (sp == 0)
			return kRefDone;
		--sp;
		return stack_load<OVF>(M, stk, sp);
	}
	if (key[3] != 0xFFFFFFFFu) { stack_store<OVF>(M, stk, sp, ref_of(key[3])); ++sp; }
	if (key[2] != 0xFFFFFFFFu) { stack_store<OVF>(M, stk, sp, ref_of(key[2])); ++sp; }
	if (key[1] != 0xFFFFFFFFu) { stack_store<OVF>(M, stk, sp, ref_of(key[1])); ++sp; }
	return ref_of(key[0]);
}

// the candidate test of the wide walk: does the ray hit this leaf's EXACT reference box (reference predicate)?  `leaf` is
// a wide-tree leaf reference (kLeafFlag | leaf index); `ref` receives the leaf's own reference -- (count, first slot) or a
// big-leaf index, what leaf_range / closest_in_leaf / any_in_leaf take -- from the spare word of the box record
__device__ __forceinline__ bool wide_leaf_hit(const DevScene &S, uint32_t leaf, const Ray &r, uint32_t &ref)
{
	const float4 *q = reinterpret_cast<const float4 *>(&S.leaf_box[leaf & kLeafSlotMask]);
	const float4 a = q[0], b = q[1];
	ref = __float_as_uint(a.w);
	const float lo[3] = {a.x, a.y, a.z}, hi[3] = {b.x, b.y, b.z};
	float t;
	return aabb_does_int(lo, hi, r, t);
}

// The root test of Bvh::get_intersection_candidates (mod.rs:203-210).  It decides something only when the
// root is a leaf: an inner root's bounds contain both children's, and the slab test is monotone in the
// bounds (each t interval of the larger box contains the smaller box's; a NaN from 0 * inf is ignored by
// min/max on either side), so "root missed" implies "both children missed" and the walk below finds
// nothing anyway.  The branch is wave-uniform.
__device__ __forceinline__ bool root_box_misses(const DevScene &S, const Ray &r)
{
	if (!ref_is_leaf(S.root_ref))
		return false;
	float tm;
	return !aabb_does_int(S.root_min, S.root_max, r, tm);
}

// ---- wave-uniform loads, for the walk of a two-leaf tree (below): the node and the primitives of its leaves are the same
// for every lane, so ONE scalar load brings a record into SGPRs for the whole wave.  The compiler emits scalar loads by
// itself only through `const __restrict__` kernel arguments (a pointer that was itself loaded from memory might alias
// the frame), and three more kernel arguments cost the fine kernels 10 % (rt_render.hip); hence the instruction is written
// out.  `p` must be wave-uniform and point at global memory nothing writes during the kernel. ----
typedef uint32_t u32x16_t __attribute__((ext_vector_type(16)));
typedef uint32_t u32x8_t __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
// (the pointer IS wave-uniform; where the compiler's divergence analysis cannot tell, this puts it into SGPRs -- and folds away where it can)
template <class T> __device__ __forceinline__ const T *as_scalar_pointer(const T *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
	const uint64_t a = reinterpret_cast<uint64_t>(p);
	const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
	return reinterpret_cast<const T *>(((uint64_t)hi << 32) | lo);
#else
	return p;
#endif
}
__device__ __forceinline__ NodeView load_node_uniform(const DevNode *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
	p = as_scalar_pointer(p);
	u32x16_t q;
	asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(q) : "s"(p) : "memory");
	NodeView n;
	n.c0min[0] = __uint_as_float(q[0]); n.c0min[1] = __uint_as_float(q[1]); n.c0min[2] = __uint_as_float(q[2]);
	n.c0max[0] = __uint_as_float(q[3]); n.c0max[1] = __uint_as_float(q[4]); n.c0max[2] = __uint_as_float(q[5]);
	n.c1min[0] = __uint_as_float(q[6]); n.c1min[1] = __uint_as_float(q[7]); n.c1min[2] = __uint_as_float(q[8]);
	n.c1max[0] = __uint_as_float(q[9]); n.c1max[1] = __uint_as_float(q[10]); n.c1max[2] = __uint_as_float(q[11]);
	n.c0 = q[12];
	n.c1 = q[13];
	return n;
#else
	NodeView n{};
	(void)p;
	return n;
#endif
}
template <class F> __device__ __forceinline__ PrimGeom load_prim_uniform(const DevPrim *p)
{
	PrimGeom g;
#if defined(__HIP_DEVICE_COMPILE__)
	p = as_scalar_pointer(p);
	u32x8_t ab;
	asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ab) : "s"(p) : "memory");
	const uint32_t meta = ab[3];
	g.type = meta & 3u;
	g.material = meta >> 2;
	g.p0 = v3(__uint_as_float(ab[0]), __uint_as_float(ab[1]), __uint_as_float(ab[2]));
	g.p1 = v3(__uint_as_float(ab[4]), __uint_as_float(ab[5]), __uint_as_float(ab[6]));
	g.p2 = v3(0.0f, 0.0f, 0.0f);
	if (F::tri && g.type != kPrimSphere) { // (wave-uniform branch)
		u32x4_t c;
		asm volatile("s_load_dwordx4 %0, %1, 0x20\n\ts_waitcnt lgkmcnt(0)" : "=s"(c) : "s"(p) : "memory");
		g.p2 = v3(__uint_as_float(c[0]), __uint_as_float(c[1]), __uint_as_float(c[2]));
	}
#else
	(void)p;
	g.type = g.material = 0u;
	g.p0 = g.p1 = g.p2 = v3(0.0f, 0.0f, 0.0f);
#endif
	return g;
}

// ... two records in ONE round of scalar loads (a two-leaf tree whose leaves hold one primitive each: rtweekend1).  Measured
// same-box: 27.4 -> 25.5 ms on config 2 at 256 spp.  Fetching the node in that same round as well (slots supplied by the host)
// was measured too: no further gain, three more VGPR spills; so was handing the winner's record (selected from these registers)
// on to make_hit instead of its own fetch: 25.9 -> 27.1 ms, twenty more spilled VGPRs.
template <class F> __device__ __forceinline__ void load_prim_pair_uniform(const DevPrim *pa, const DevPrim *pb, PrimGeom &ga, PrimGeom &gb)
{
#if defined(__HIP_DEVICE_COMPILE__)
	pa = as_scalar_pointer(pa);
	pb = as_scalar_pointer(pb);
	u32x8_t a, b;
	u32x4_t ca = {0u, 0u, 0u, 0u}, cb = {0u, 0u, 0u, 0u};
	if (F::tri)
		asm volatile("s_load_dwordx8 %0, %4, 0x0\n\ts_load_dwordx8 %1, %5, 0x0\n\ts_load_dwordx4 %2, %4, 0x20\n\ts_load_dwordx4 %3, %5, 0x20\n\ts_waitcnt lgkmcnt(0)"
		             : "=&s"(a), "=&s"(b), "=&s"(ca), "=&s"(cb) : "s"(pa), "s"(pb) : "memory");
	else
		asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dwordx8 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(b) : "s"(pa), "s"(pb) : "memory");
	auto unpack = [](const u32x8_t &q, const u32x4_t &c, PrimGeom &g) {
		g.type = q[3] & 3u;
		g.material = q[3] >> 2;
		g.p0 = v3(__uint_as_float(q[0]), __uint_as_float(q[1]), __uint_as_float(q[2]));
		g.p1 = v3(__uint_as_float(q[4]), __uint_as_float(q[5]), __uint_as_float(q[6]));
		g.p2 = (F::tri && g.type != kPrimSphere) ? v3(__uint_as_float(c[0]), __uint_as_float(c[1]), __uint_as_float(c[2])) : v3(0.0f, 0.0f, 0.0f);
	};
	unpack(a, ca, ga);
	unpack(b, cb, gb);
#else
	(void)pa; (void)pb;
	ga.type = ga.material = gb.type = gb.material = 0u;
	ga.p0 = ga.p1 = ga.p2 = gb.p0 = gb.p1 = gb.p2 = v3(0.0f, 0.0f, 0.0f);
#endif
}
// the selection rule of Bvh::check_hit (mod.rs:270-293) for one candidate: smallest t > 0, ties to the first in BFS-leaf order
__device__ __forceinline__ bool leaf_holds_one(uint32_t ref) { return ((ref >> 26) & 31u) == 1u; }
__device__ __forceinline__ void consider_closest(const DevScene &S, uint32_t slot, float t, float &best_t, uint32_t &best_prim)
{
	bool take;
	if (best_prim == kNoPrim)
		take = true;
	else if (t < best_t)
		take = true;
	else if (t == best_t)
		take = S.prim_rank[slot] < S.prim_rank[best_prim];
	else
		take = false;
	if (take) {
		best_t = t;
		best_prim = slot;
	}
}

// the primitive loop of Bvh::check_hit for one leaf: selection rule of mod.rs:270-293
template <class F, bool UNIFORM = false>
__device__ __forceinline__ void closest_in_leaf(const DevScene &S, const Ray &r, uint32_t leaf, float &best_t, uint32_t &best_prim)
{
	uint32_t first, count;
	leaf_range(S, leaf, first, count);
	for (uint32_t slot = first; slot < first + count; ++slot) {
		const PrimGeom g = UNIFORM ? load_prim_uniform<F>(&S.prims[slot]) : load_prim<F>(S, slot);
		float t;
		if (prim_t<F>(g, r, t) && t > 0.0f) {
			bool take;
			if (best_prim == kNoPrim)
				take = true;
			else if (t < best_t)
				take = true;
			else if (t == best_t)
				take = S.prim_rank[slot] < S.prim_rank[best_prim]; // reference order: first in BFS-leaf order wins
			else
				take = false;
			if (take) {
				best_t = t;
				best_prim = slot;
			}
		}
	}
}
// ... and of the occlusion rule (see trace_any)
template <class F, bool UNIFORM = false>
__device__ __forceinline__ bool any_in_leaf(const DevScene &S, const Ray &r, uint32_t leaf, float t_limit, uint32_t skip)
{
	uint32_t first, count;
	leaf_range(S, leaf, first, count);
	for (uint32_t slot = first; slot < first + count; ++slot) {
		if (slot == skip)
			continue;
		const PrimGeom g = UNIFORM ? load_prim_uniform<F>(&S.prims[slot]) : load_prim<F>(S, slot);
		float t;
		if (prim_t<F>(g, r, t) && t > 0.0f && !(t >= t_limit))
			return true;
	}
	return false;
}
// A tree of one inner node over two leaves (rtweekend1: ground sphere + ball) needs no loop and no stack:
// both child boxes, then the leaves whose box was hit, in the reference's candidate order.  Wave-uniform.
__device__ __forceinline__ bool is_two_leaf_tree(const DevScene &S) { return S.n_nodes == 1u && !ref_is_leaf(S.root_ref); }

// Bvh::check_hit: smallest t > 0, ties to the primitive first in BFS-leaf order (mod.rs:265-298)
// FeatPair: the node's two boxes from the kernel arguments (rt_types.h DevPairScene)
struct PairBoxes {
	float c0min[3], c0max[3], c1min[3], c1max[3];
};
__device__ __forceinline__ PairBoxes pair_boxes(const DevPairScene &ps)
{
	PairBoxes b;
#pragma unroll
	for (int k = 0; k < 3; ++k) {
		b.c0min[k] = ps.c0min[k]; b.c0max[k] = ps.c0max[k];
		b.c1min[k] = ps.c1min[k]; b.c1max[k] = ps.c1max[k];
	}
	return b;
}

template <class F, bool PRUNE, bool OVF = false>
__device__ __forceinline__ void trace_closest(const DevScene &S, const DevScene &SU, const StackMem &M, const Ray &r, uint32_t *stk, float &best_t,
                                              uint32_t &best_prim, const DevPairScene &ps = kNoPairScene)
{
	best_t = 0.0f;
	best_prim = kNoPrim;
	if (!F::pair && root_box_misses(S, r)) // (FeatPair: the root is an inner node)
		return;
	if constexpr (F::pair) { // the host launches this set only for such a tree: nothing else is compiled in
		// both child boxes, then the leaves whose box was hit, in the reference's candidate order; the selection rule of
		// Bvh::check_hit (mod.rs:270-293) for two candidates: smallest t > 0, an exact tie to the first in BFS-leaf order
		const PairBoxes n = pair_boxes(ps);
		float t0, t1, t;
		const bool h0 = aabb_does_int(n.c0min, n.c0max, r, t0);
		const bool h1 = aabb_does_int(n.c1min, n.c1max, r, t1);
		if (h0 && sphere_t(v3(ps.sphere[0][0], ps.sphere[0][1], ps.sphere[0][2]), ps.sphere[0][3], r, t) && t > 0.0f) {
			best_t = t;
			best_prim = ps.slot0;
		}
		if (h1 && sphere_t(v3(ps.sphere[1][0], ps.sphere[1][1], ps.sphere[1][2]), ps.sphere[1][3], r, t) && t > 0.0f) {
			if (best_prim == kNoPrim || t < best_t || (t == best_t && ps.rank1 < ps.rank0)) {
				best_t = t;
				best_prim = ps.slot1;
			}
		}
		return;
	}
	if (!PRUNE && is_two_leaf_tree(S)) {
		// SU is the scene as the kernel received it (global memory): every address below is wave-uniform,
		// so the node and the primitives arrive through scalar loads and the tests read them from SGPRs
		const NodeView n = load_node_uniform(SU.nodes);
		float t0, t1;
		const bool h0 = aabb_does_int(n.c0min, n.c0max, r, t0);
		const bool h1 = aabb_does_int(n.c1min, n.c1max, r, t1);
		if (leaf_holds_one(n.c0) && leaf_holds_one(n.c1)) { // (wave-uniform) one primitive per leaf: both records in one round of loads
			const uint32_t s0 = n.c0 & kLeafSlotMask, s1 = n.c1 & kLeafSlotMask;
			PrimGeom g0, g1;
			load_prim_pair_uniform<F>(&SU.prims[s0], &SU.prims[s1], g0, g1);
			float t;
			if (h0 && prim_t<F>(g0, r, t) && t > 0.0f)
				consider_closest(SU, s0, t, best_t, best_prim);
			if (h1 && prim_t<F>(g1, r, t) && t > 0.0f)
				consider_closest(SU, s1, t, best_t, best_prim);
			return;
		}
#pragma unroll 1
		for (int c = 0; c < 2; ++c)
			if (c == 0 ? h0 : h1)
				closest_in_leaf<F, true>(SU, r, c == 0 ? n.c0 : n.c1, best_t, best_prim);
		return;
	}
	int sp = 0;
	const bool wide = PRUNE && S.nodes4 != nullptr && S.narrow_only == 0u && ray_is_regular(r);
	uint32_t node = wide ? S.root4_ref : S.root_ref;
	// The ROOT of the two-child tree is the same record for every lane of every walk: under the exhaustive walk (no lane takes
	// the wide tree) it arrives through one scalar load and its two box tests read it from SGPRs -- one per-lane LDS / L1 round
	// trip less at the head of each walk's chain of dependent fetches (config 3: three walks per sample)
	if (!PRUNE && !ref_is_leaf(S.root_ref)) // (wave-uniform)
		node = descend_view<PRUNE, OVF>(load_node_uniform(SU.nodes + S.root_ref), M, r, stk, sp, false, 0.0f);
	while (node != kRefDone) {
		while (!ref_is_leaf(node) && node != kRefDone)
			node = wide ? descend4<PRUNE, OVF>(S, M, r, node, stk, sp, best_prim != kNoPrim, best_t)
			            : descend<PRUNE, OVF>(S, M, r, node, stk, sp, best_prim != kNoPrim, best_t);
		if (node == kRefDone)
			break;
		uint32_t leaf = node;
		if (!wide || wide_leaf_hit(S, node, r, leaf))
			closest_in_leaf<F>(S, r, leaf, best_t, best_prim);
		if (sp == 0)
			break;
		--sp;
		node = stack_load<OVF>(M, stk, sp);
	}
}

// "is anything in the way": the occlusion rule shared by the sky shadow ray (Bvh::check_hit
// returning an index != usize::MAX, mis.rs:104-115) and Bvh::check_hit_index (mod.rs:244-261):
// some primitive other than `skip` has 0 < t and NOT (t >= t_limit).  t_limit = NaN means "no
// limit" (any t > 0 occludes).
template <class F, bool PRUNE, bool OVF = false>
__device__ __forceinline__ bool trace_any(const DevScene &S, const DevScene &SU, const StackMem &M, const Ray &r, uint32_t *stk, float t_limit,
                                          uint32_t skip, const DevPairScene &ps = kNoPairScene)
{
	if (!F::pair && root_box_misses(S, r))
		return false;
	const bool limited = !(t_limit != t_limit);
	if constexpr (F::pair) { // as in trace_closest
		const PairBoxes n = pair_boxes(ps);
		float t0, t1, t;
		const bool h0 = aabb_does_int(n.c0min, n.c0max, r, t0);
		const bool h1 = aabb_does_int(n.c1min, n.c1max, r, t1);
		if (h0 && ps.slot0 != skip && sphere_t(v3(ps.sphere[0][0], ps.sphere[0][1], ps.sphere[0][2]), ps.sphere[0][3], r, t) && t > 0.0f && !(t >= t_limit))
			return true;
		return h1 && ps.slot1 != skip && sphere_t(v3(ps.sphere[1][0], ps.sphere[1][1], ps.sphere[1][2]), ps.sphere[1][3], r, t) && t > 0.0f &&
		       !(t >= t_limit);
	}
	if (!PRUNE && is_two_leaf_tree(S)) {
		const NodeView n = load_node_uniform(SU.nodes);
		float t0, t1;
		const bool h0 = aabb_does_int(n.c0min, n.c0max, r, t0);
		const bool h1 = aabb_does_int(n.c1min, n.c1max, r, t1);
		if (leaf_holds_one(n.c0) && leaf_holds_one(n.c1)) { // (wave-uniform) as in trace_closest
			const uint32_t s0 = n.c0 & kLeafSlotMask, s1 = n.c1 & kLeafSlotMask;
			PrimGeom g0, g1;
			load_prim_pair_uniform<F>(&SU.prims[s0], &SU.prims[s1], g0, g1);
			float t;
			if (h0 && s0 != skip && prim_t<F>(g0, r, t) && t > 0.0f && !(t >= t_limit))
				return true;
			return h1 && s1 != skip && prim_t<F>(g1, r, t) && t > 0.0f && !(t >= t_limit);
		}
		bool occluded = false;
#pragma unroll 1
		for (int c = 0; c < 2; ++c)
			if (!occluded && (c == 0 ? h0 : h1))
				occluded = any_in_leaf<F, true>(SU, r, c == 0 ? n.c0 : n.c1, t_limit, skip);
		return occluded;
	}
	int sp = 0;
	const bool wide = PRUNE && S.nodes4 != nullptr && S.narrow_only == 0u && ray_is_regular(r);
	uint32_t node = wide ? S.root4_ref : S.root_ref;
	if (!PRUNE && !ref_is_leaf(S.root_ref)) // the root through a scalar load: see trace_closest
		node = descend_view<PRUNE, OVF>(load_node_uniform(SU.nodes + S.root_ref), M, r, stk, sp, false, 0.0f);
	while (node != kRefDone) {
		while (!ref_is_leaf(node) && node != kRefDone)
			node = wide ? descend4<PRUNE, OVF>(S, M, r, node, stk, sp, limited, t_limit)
			            : descend<PRUNE, OVF>(S, M, r, node, stk, sp, limited, t_limit);
		if (node == kRefDone)
			break;
		uint32_t leaf = node;
		if ((!wide || wide_leaf_hit(S, node, r, leaf)) && any_in_leaf<F>(S, r, leaf, t_limit, skip))
			return true;
		if (sp == 0)
			break;
		--sp;
		node = stack_load<OVF>(M, stk, sp);
	}
	return false;
}

} // namespace rt
