// rt_lean.h -- the SAME f32 results as include/rt_detmath.h and the IEEE operators, in fewer gfx950 instructions.
//
// The arithmetic contract (rt_detmath.h + correctly rounded +,-,*,/,sqrt) fixes WHAT every operation returns.  How hipcc
// gets there is expensive on this chip, and a third of the render kernels' vector instructions is spent on it:
//   * `a / b`       11 instructions: v_div_scale x2, v_rcp, 2 fma (reciprocal refinement), mul + 4 fma (quotient with two
//                   corrections), v_div_fmas, v_div_fixup;
//   * `sqrtf(x)`    16: a 2^32 pre-scale for tiny inputs, v_sqrt, two one-ulp corrections, the un-scale, a class fix-up;
//   * rt_sinf / rt_cosf / rt_acosf / rt_atan2f compile to DIVERGENT branches: both polynomial kernels run in turn
//     whenever the lanes of a wave disagree (they always do), sin and cos of one angle reduce the argument twice, and the
//     guards of the huge-argument paths are paid on every call.
// This header restates them so that the instructions which cannot change the result are not executed:
//   * lean_sincos / lean_acos / lean_atan2: the operations of rt_detmath.h on every input of their stated domain, one
//     polynomial evaluation per result, selects instead of branches.  Plain C++ over +,-,*,/,sqrt,fmaf: they compile on
//     the host too, where tests/test_lean_math.py checks them against rt_detmath.h bit for bit (exhaustively over the
//     float ranges the render path feeds them);
//   * lean_div* / lean_sqrt (device only): hipcc's own expansion MINUS the steps that are the identity on "tame" operands.
//     v_div_scale_f32 returns its operand unchanged and clears VCC unless an operand is zero / denormal / the quotient or
//     the reciprocal would leave the normal range (exponent differences >= 96 or <= -126, numerator below 2^-103);
//     v_div_fmas_f32 with VCC clear is v_fma_f32; v_div_fixup_f32 returns the quotient unchanged unless an operand is
//     zero, infinite or NaN or the exponents differ by more than the format allows.  So for operands whose magnitudes lie
//     in [2^-81, 2^41] and whose exponent difference stays below 96 the remaining instructions compute bit for bit what
//     `a / b` computes.  Every use states why its operands are tame, or tests it (one or two compares per ray) and takes
//     the plain operator otherwise.  rt_selftest_lean (rt_api.cpp) runs both forms side by side ON THE DEVICE over random
//     and edge-case operands; tests/test_gpu_parity.py::test_lean_arithmetic_matches_the_ieee_operators asserts zero
//     mismatches, and every parity test compares frames that went through them with the CPU checker's plain C.
#pragma once

#include "rt_vec.h"

namespace rt {

// ---- branch-free restatements of rt_detmath.h (host + device) ----

// (rt_sinf(x), rt_cosf(x)) for |x| <= 2^22 (rt_reduce_pio2's Cody-Waite range; finite, so neither the NaN guard nor the
// binary64 fold of rt_detmath.h can trigger).  Callers: angles 2*pi*r and pi*v with r, v in [0, 1 + 2^-20].
RT_FN void lean_sincos(float x, float &s, float &c)
{
	const float k = rintf(x * 0.636619746685028076172f);
	float r = fmaf(-k, 1.57079637050628662109f, x);
	r = fmaf(-k, -4.37113882867379282984e-08f, r);
	r = fmaf(-k, -1.71512451000588185479e-15f, r);
	const int q = (int)k;
	const float sk = rt_sin_kernel(r), ck = rt_cos_kernel(r);
	const bool odd = (q & 1) != 0;
	const float sv = odd ? ck : sk;
	const float cv = odd ? sk : ck;
	s = (q & 2) ? -sv : sv;
	c = ((q + 1) & 2) ? -cv : cv;
}

// rt_acosf(x) for every x (NaN and |x| > 1 give NaN there and here): 1 + x == 1 - |x| for x < 0, so the two outer arms
// share one square root and all three share one polynomial.
RT_FN float lean_acos_with(float x, float sqrt_half_one_minus_abs)
{
	const bool big = fabsf(x) > 0.5f;
	const float k = rt_asin_kernel(big ? sqrt_half_one_minus_abs : x);
	const float two_k = 2.0f * k;
	return big ? (x > 0.0f ? two_k : RT_PI - two_k) : RT_FRAC_PI_2 - k;
}
RT_FN float lean_acos(float x) { return lean_acos_with(x, sqrtf(0.5f * (1.0f - fabsf(x)))); }

// rt_atan_unit(t) for t in [0, 1] given q = (t - 1) / (t + 1) (used only where t > tan(pi/8))
RT_FN float lean_atan_unit_with(float t, float q)
{
	const bool upper = t > 0.414213567972183227539f;
	const float u = upper ? q : t;
	const float z = u * u;
	float p = fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
	p = fmaf(p, z, 1.99777106478e-1f);
	p = fmaf(p, z, -3.33329491539e-1f);
	return (upper ? RT_FRAC_PI_4 : 0.0f) + fmaf(p * z, u, u);
}
// rt_atan2f(y, x) for every (y, x): the quotient min/max is the quotient the taken arm of rt_atan2f forms
template <class DivA, class DivB> RT_FN float lean_atan2_with(float y, float x, DivA div_ratio, DivB div_octant)
{
	const float ax = fabsf(x), ay = fabsf(y);
	const bool swap = ay > ax;
	const float t = div_ratio(swap ? ax : ay, swap ? ay : ax);
	const float a = lean_atan_unit_with(t, div_octant(t - 1.0f, t + 1.0f));
	float r = swap ? RT_FRAC_PI_2 - a : a;
	r = (ax == ay) ? RT_FRAC_PI_4 : r;
	r = (ax == 0.0f && ay == 0.0f) ? 0.0f : r;
	if (rt_f32_bits(x) & 0x80000000u)
		r = RT_PI - r;
	r = (rt_f32_bits(y) & 0x80000000u) ? -r : r;
	return (x != x || y != y) ? x + y : r;
}
RT_FN float lean_atan2_portable(float y, float x)
{
	return lean_atan2_with(y, x, [](float n, float d) { return n / d; }, [](float n, float d) { return n / d; });
}

// ---- x / c for a divisor the launch knows beforehand, c in [2^-20, 2^32], rc = RN(1 / c) formed on the host, which has also
// VERIFIED (rt_build.cpp verified_reciprocal: every one of the 2^23 significands of x) that these three IEEE operations return the
// correctly rounded quotient.  The sequence is scale-invariant while nothing under- or overflows, so it holds for x = 0, NaN
// and 2^-60 <= |x| <= 2^60 (x = -0 comes back as +0 where the division returns -0: no use below feeds a -0 whose sign could
// reach a pixel); each use says why its x is in that set (or tests it).  Three plain instructions instead of eleven
// with a v_rcp_f32 among them: 7.5 issue cycles instead of 33 (profiles/r04_valu_issue.txt). ----
RT_FN float div_by_verified(float x, float c, float rc)
{
	const float q0 = x * rc;
	const float e = fmaf(-c, q0, x);
	return fmaf(e, rc, q0);
}
// the two constants the shading code divides by: verified like any other (tests/test_lean_math.py asks the library), their
// reciprocals are the compile-time IEEE quotients
constexpr float kRcpPi = 1.0f / RT_PI, kRcpTau = 1.0f / RT_TAU;

#if defined(__HIP_DEVICE_COMPILE__)
// ---- device: hipcc's f32 division and square root without the identity steps (see the header comment) ----

// r1 of the expansion: v_rcp_f32 and one Newton step.  d tame.
__device__ __forceinline__ float lean_rcp_refined(float d)
{
	const float r0 = __builtin_amdgcn_rcpf(d);
	const float e = __builtin_fmaf(-d, r0, 1.0f);
	return __builtin_fmaf(e, r0, r0);
}
// quotient from the refined reciprocal: mul, then two (residual, correction) pairs; the last fma is v_div_fmas with VCC = 0
__device__ __forceinline__ float lean_div_core(float n, float d, float r1)
{
	const float q0 = n * r1;
	const float e1 = __builtin_fmaf(-d, q0, n);
	const float q1 = __builtin_fmaf(e1, r1, q0);
	const float e2 = __builtin_fmaf(-d, q1, n);
	return __builtin_fmaf(e2, r1, q1);
}
// n / d, both tame (|n|, |d| in [2^-81, 2^41], exponent difference < 96)
__device__ __forceinline__ float lean_div(float n, float d) { return lean_div_core(n, d, lean_rcp_refined(d)); }
// n / d, d tame, n tame OR zero / infinite / NaN (v_div_fixup_f32 supplies the IEEE result for those, as in `n / d`)
__device__ __forceinline__ float lean_div_fix(float n, float d)
{
	return __builtin_amdgcn_div_fixupf(lean_div_core(n, d, lean_rcp_refined(d)), d, n);
}
// 1 / d, d tame: q0 = 1 * r1 is r1 itself
__device__ __forceinline__ float lean_inv(float d)
{
	const float r1 = lean_rcp_refined(d);
	const float e1 = __builtin_fmaf(-d, r1, 1.0f);
	const float q1 = __builtin_fmaf(e1, r1, r1);
	const float e2 = __builtin_fmaf(-d, q1, 1.0f);
	return __builtin_fmaf(e2, r1, q1);
}
// v / d componentwise with ONE refined reciprocal; v's components and d tame
__device__ __forceinline__ V3 lean_div3(V3 v, float d)
{
	const float r1 = lean_rcp_refined(d);
	return V3{lean_div_core(v.x, d, r1), lean_div_core(v.y, d, r1), lean_div_core(v.z, d, r1)};
}
// ... components tame or zero / infinite / NaN
__device__ __forceinline__ V3 lean_div3_fix(V3 v, float d)
{
	const float r1 = lean_rcp_refined(d);
	return V3{__builtin_amdgcn_div_fixupf(lean_div_core(v.x, d, r1), d, v.x), __builtin_amdgcn_div_fixupf(lean_div_core(v.y, d, r1), d, v.y),
	          __builtin_amdgcn_div_fixupf(lean_div_core(v.z, d, r1), d, v.z)};
}

// sqrtf(x) for x in {+-0} U [2^-96, +inf] U NaN U negatives: v_sqrt_f32 (1 ulp) and the expansion's two one-ulp corrections.
// The 2^32 pre-scale the expansion applies below 2^-96 and its final class test (which returns x itself for +-0 and +inf)
// are the identity there: for x = +-0 both corrections compare NaN / +-0 residuals and keep v_sqrt's +-0, for +inf they keep inf.
__device__ __forceinline__ float lean_sqrt(float x)
{
	const float s = __builtin_amdgcn_sqrtf(x);
	const float s_dn = __uint_as_float(__float_as_uint(s) - 1u);
	const float s_up = __uint_as_float(__float_as_uint(s) + 1u);
	const float r_dn = __builtin_fmaf(-s_dn, s, x);
	const float r_up = __builtin_fmaf(-s_up, s, x);
	float r = (r_dn <= 0.0f) ? s_dn : s;
	r = (r_up > 0.0f) ? s_up : r;
	return r;
}
// smallest input the short form is exact for
constexpr float kLeanSqrtMin = 0x1p-96f;

// the divisions inside rt_atan2f on the device: the octant fold (t - 1) / (t + 1) has t in [0, 1] (NaN when 0 / 0 or
// inf / inf, where the result is overridden): numerator zero or of magnitude >= 2^-24, denominator in [1, 2]
__device__ __forceinline__ float lean_atan2(float y, float x)
{
	return lean_atan2_with(y, x, [](float n, float d) { return n / d; }, [](float n, float d) { return lean_div_fix(n, d); });
}
// acos with the short square root: 0.5 * (1 - |x|) is zero, negative, NaN or >= 2^-25
__device__ __forceinline__ float lean_acos_dev(float x) { return lean_acos_with(x, lean_sqrt(0.5f * (1.0f - fabsf(x)))); }
#else
// host pass of the same translation units (kernel bodies are parsed, never run, there): the plain operators
RT_FN float lean_rcp_refined(float d) { return 1.0f / d; }
RT_FN float lean_div_core(float n, float d, float) { return n / d; }
RT_FN float lean_div(float n, float d) { return n / d; }
RT_FN float lean_div_fix(float n, float d) { return n / d; }
RT_FN float lean_inv(float d) { return 1.0f / d; }
RT_FN V3 lean_div3(V3 v, float d) { return v / d; }
RT_FN V3 lean_div3_fix(V3 v, float d) { return v / d; }
RT_FN float lean_sqrt(float x) { return sqrtf(x); }
constexpr float kLeanSqrtMin = 0x1p-96f;
RT_FN float lean_atan2(float y, float x) { return lean_atan2_portable(y, x); }
RT_FN float lean_acos_dev(float x) { return lean_acos(x); }
#endif

} // namespace rt
