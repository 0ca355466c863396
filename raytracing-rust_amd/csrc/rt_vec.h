// rt_vec.h -- Vec3 arithmetic of the HIP back end (host + device).
//
// Mirrors rt_core/src/vec.rs of the reference operation for operation (componentwise
// operators :10-106, dot :165-168 as (x*x + y*y) + z*z, cross :170-177, reflected :205-208,
// component_max :216-223 as x.max(y.max(z))).  Compiled with -ffp-contract=off: a fused
// multiply-add appears only where include/rt_detmath.h spells fmaf().
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "../../include/rt_detmath.h"

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

#define RT_FN __host__ __device__ __forceinline__

namespace rt {

struct V3 {
	float x, y, z;
};

RT_FN V3 v3(float x, float y, float z) { return V3{x, y, z}; }
RT_FN V3 v3s(float s) { return V3{s, s, s}; }
RT_FN V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RT_FN V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RT_FN V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
RT_FN V3 operator/(V3 a, V3 b) { return V3{a.x / b.x, a.y / b.y, a.z / b.z}; }
RT_FN V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
RT_FN V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
RT_FN V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
RT_FN V3 operator-(float s, V3 a) { return V3{s - a.x, s - a.y, s - a.z}; }
RT_FN V3 operator+(float s, V3 a) { return V3{s + a.x, s + a.y, s + a.z}; }
RT_FN V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }

RT_FN float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_FN V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
RT_FN float mag_sq(V3 a) { return dot(a, a); }
RT_FN float mag(V3 a) { return sqrtf(dot(a, a)); }
RT_FN V3 normalised(V3 a) { return a / mag(a); }
RT_FN V3 vabs(V3 a) { return V3{fabsf(a.x), fabsf(a.y), fabsf(a.z)}; }

// f32::min / f32::max ignore a NaN operand (aabb.rs:31-56 relies on it).  On the device fminf/fmaxf lower
// to v_min_f32 / v_max_f32 (IEEE mode), which do the same; everything they feed there is compared or
// divided by later, where the sign of a +0/-0 tie cannot matter.  On the host (BVH bounds, which
// rt_scene_get_nodes hands out) the tie is resolved the way rustc's f32::min/max resolve it on x86-64 --
// LLVM's minnum/maxnum lowering keeps the FIRST operand -- rather than the way libm happens to.
#if defined(__HIP_DEVICE_COMPILE__)
RT_FN float fmin_(float a, float b) { return fminf(a, b); }
RT_FN float fmax_(float a, float b) { return fmaxf(a, b); }
#else
RT_FN float fmin_(float a, float b) { return a != a ? b : (b < a ? b : a); }
RT_FN float fmax_(float a, float b) { return a != a ? b : (b > a ? b : a); }
#endif
RT_FN float component_max(V3 a) { return fmax_(a.x, fmax_(a.y, a.z)); }
RT_FN V3 min_by_component(V3 a, V3 b) { return V3{fmin_(a.x, b.x), fmin_(a.y, b.y), fmin_(a.z, b.z)}; }
RT_FN V3 max_by_component(V3 a, V3 b) { return V3{fmax_(a.x, b.x), fmax_(a.y, b.y), fmax_(a.z, b.z)}; }
RT_FN V3 reflected(V3 self, V3 normal) { return 2.0f * dot(self, normal) * normal - self; }
RT_FN bool contains_nan(V3 a) { return (a.x != a.x) || (a.y != a.y) || (a.z != a.z); }
RT_FN bool finite_f(float f) { return fabsf(f) <= 3.40282347e+38f; } // false for inf and NaN
// Vec3::is_finite is an OR over the components (vec.rs:245-247)
RT_FN bool is_finite_any(V3 a) { return finite_f(a.x) || finite_f(a.y) || finite_f(a.z); }
RT_FN bool is_zero(V3 a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }

// constants of rt_core/src/lib.rs:23-32 and std::f32
constexpr float kEpsilon = 3.0e-4f;               // rt_core::EPSILON
constexpr float kF32Epsilon = 1.1920928955078125e-07f; // f32::EPSILON
constexpr float kPi = RT_PI;
constexpr float kTau = RT_TAU;

// utility::gamma(n)  utility/mod.rs:83-86
RT_FN constexpr float gamma_n(int n)
{
	return ((float)n * 0.5f * kF32Epsilon) / (1.0f - (float)n * 0.5f * kF32Epsilon);
}

} // namespace rt
