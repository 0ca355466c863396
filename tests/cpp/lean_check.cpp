// lean_check.cpp -- host-side proof by enumeration that the branch-free elementary functions of
// raytracing-rust_amd/csrc/rt_lean.h return the bits of include/rt_detmath.h on the domains the render path feeds them.
// Built by tests/test_lean_math.py with `hipcc --cuda-host-only` (the header includes the HIP runtime header for its
// qualifiers; no device code is built or run).  Prints one line per check: "<name> <values tested> <mismatches>".
#include <cstdio>
#include <cstring>
#include <cstdint>
#include <thread>
#include <vector>
#include <atomic>
#include "../../raytracing-rust_amd/csrc/rt_lean.h"

static inline float f_of(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline uint32_t u_of(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline bool same(float a, float b) { return u_of(a) == u_of(b) || (a != a && b != b); }

template <class Fn> static uint64_t par_range(uint32_t lo, uint32_t hi_inclusive, uint32_t stride, Fn fn)
{
	const unsigned T = 8;
	std::atomic<uint64_t> bad{0};
	std::vector<std::thread> th;
	const uint64_t n = ((uint64_t)hi_inclusive - lo) / stride + 1;
	for (unsigned t = 0; t < T; ++t)
		th.emplace_back([&, t] {
			uint64_t b = 0;
			for (uint64_t i = t; i < n; i += T)
				b += fn((uint32_t)(lo + i * stride)) ? 0 : 1;
			bad += b;
		});
	for (auto &x : th) x.join();
	return bad.load();
}

int main(int argc, char **argv)
{
	const uint32_t stride = argc > 1 ? (uint32_t)atoi(argv[1]) : 1u; // 1 = exhaustive
	// sin + cos: every float in [0, 2^22] and its negative
	{
		const uint32_t hi = u_of(4194304.0f);
		const uint64_t bad = par_range(0u, hi, stride, [](uint32_t u) {
			bool ok = true;
			for (int sgn = 0; sgn < 2; ++sgn) {
				const float x = f_of(u | (sgn ? 0x80000000u : 0u));
				float s, c;
				rt::lean_sincos(x, s, c);
				ok = ok && same(s, rt_sinf(x)) && same(c, rt_cosf(x));
			}
			return ok;
		});
		std::printf("sincos %llu %llu\n", (unsigned long long)(2ull * (((uint64_t)hi) / stride + 1)), (unsigned long long)bad);
	}
	// acos: every float (all 2^32 bit patterns when stride == 1: NaN, infinities, |x| > 1 included)
	{
		const uint64_t bad = par_range(0u, 0xFFFFFFFFu, stride, [](uint32_t u) {
			const float x = f_of(u);
			return same(rt::lean_acos(x), rt_acosf(x));
		});
		std::printf("acos %llu %llu\n", (unsigned long long)(0x100000000ull / stride), (unsigned long long)bad);
	}
	// atan2: a dense grid of exponents and mantissas in both arguments, all sign combinations, plus every special value
	{
		std::vector<float> vals;
		const uint32_t mant[] = {0u, 1u, 0x2AAAAAu, 0x3504F3u, 0x400000u, 0x54F5C3u, 0x7FFFFEu, 0x7FFFFFu};
		for (int e = 1; e <= 254; e += (stride > 1 ? 9 : 3))
			for (uint32_t m : mant)
				vals.push_back(f_of(((uint32_t)e << 23) | m));
		for (uint32_t d : {0u, 1u, 0x7FFFFFu, 0x7F800000u, 0x7FC00000u, 0x3F800000u, 0x3ED413CDu, 0x3ED413CCu, 0x3ED413CEu})
			vals.push_back(f_of(d));
		const uint32_t n = (uint32_t)vals.size();
		const uint64_t bad = par_range(0u, n * n - 1u, 1u, [&](uint32_t k) {
			const float a = vals[k / n], b = vals[k % n];
			bool ok = true;
			for (int s = 0; s < 4; ++s) {
				const float y = (s & 1) ? -a : a, x = (s & 2) ? -b : b;
				ok = ok && same(rt::lean_atan2_portable(y, x), rt_atan2f(y, x));
			}
			return ok;
		});
		std::printf("atan2_grid %llu %llu\n", (unsigned long long)(4ull * n * n), (unsigned long long)bad);
		// ... and ratios swept finely through the octant boundary and the whole of [0, 1]: y = t * x for every float t in [2^-30, 1]
		const uint64_t bad2 = par_range(u_of(0x1p-30f), u_of(1.0f), stride * 7u, [](uint32_t u) {
			const float t = f_of(u);
			bool ok = true;
			for (float x : {1.0f, 3.0f, 0.37f, 1.0e10f}) {
				const float y = t * x;
				ok = ok && same(rt::lean_atan2_portable(y, x), rt_atan2f(y, x)) && same(rt::lean_atan2_portable(x, -y), rt_atan2f(x, -y));
			}
			return ok;
		});
		std::printf("atan2_ratio %llu %llu\n", (unsigned long long)(8ull * ((u_of(1.0f) - u_of(0x1p-30f)) / (stride * 7u) + 1)), (unsigned long long)bad2);
	}
	return 0;
}
