// valu_issue.hip -- what one VALU instruction costs on gfx950 when every SIMD holds W waves issuing nothing else: cycles per
// wave-instruction per SIMD for the instruction kinds the render kernels are made of (is a 32-bit integer multiply a
// full-rate instruction?  is v_mad_u64_u32 one multiply or two?).  hipcc --offload-arch=gfx950 -O3; run: valu_issue
// Round 4: cycles are READ, not inferred.  Every wave stamps s_memtime (shader-clock ticks) and s_memrealtime (100 MHz) around
// its loop; cycles per wave-instruction per SIMD = median over waves of d(s_memtime) / (instructions per wave x waves per SIMD),
// and the clock the chip really held is d(s_memtime) / d(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS give-back (6)).
// Round 3's version turned wall milliseconds into cycles with an assumed 2.4 GHz and reported 2.75 cycles for a plain
// instruction: the chip ran that power-dense loop well below 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND> __global__ __launch_bounds__(256) void loop_kernel(unsigned *out, unsigned n_iter, unsigned seed, unsigned long long *stamps)
{
	const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
	unsigned a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9E3779B9u, c = b + 77u, d = c ^ a;
	float fa = (float)(a & 1023u) * 0.001f + 1.0f, fb = 1.0001f, fc = 0.5f, fd = 0.25f;
	unsigned long long qa = a, qb = b;
	for (unsigned i = 0; i < n_iter; ++i) {
		// four independent chains, 64 instructions of the kind per chain per iteration
		if (KIND == 0) { REP64(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd) : "v"(0.999f), "v"(1.0e-3f));) }
		if (KIND == 1) { REP64(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed | 1u));) }
		if (KIND == 2) { REP64(asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed | 0xF0000001u));) }
		if (KIND == 3) { REP64(asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1" : "+v"(qa), "+v"(qb) : "v"(a), "v"(seed | 1u) : "vcc");) }
		if (KIND == 4) { REP64(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));) }
		if (KIND == 5) { REP64(asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));) }
		if (KIND == 6) { REP64(asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed | 1u));) }
		if (KIND == 7) { REP64(asm volatile("v_pk_fma_f32 %0, %0, %2, %2\n v_pk_fma_f32 %1, %1, %2, %2\n v_pk_fma_f32 %0, %0, %2, %2\n v_pk_fma_f32 %1, %1, %2, %2" : "+v"(qa), "+v"(qb) : "v"(qa));) }
		if (KIND == 8) { REP64(asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0\n v_div_fixup_f32 %2, %2, %1, %3\n v_div_fmas_f32 %3, %3, %1, %2\n v_alignbit_b32 %4, %4, %4, 7" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd), "+v"(a) : : "vcc");) }
		if (KIND == 9) { REP64(asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));) }
		if (KIND == 10) { REP64(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_mov_b32 %1, %2\n v_cndmask_b32 %2, %2, %3, vcc\n v_mov_b32 %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : );) }
		// 4 VALU + 4 SALU of the same wave, interleaved: does a wave's scalar work take vector issue slots from the SIMD?
		if (KIND == 11) { REP64(asm volatile("v_fma_f32 %0, %0, %4, %5\n s_add_u32 %6, %6, 1\n v_fma_f32 %1, %1, %4, %5\n s_xor_b32 %6, %6, 5\n v_fma_f32 %2, %2, %4, %5\n s_add_u32 %6, %6, 3\n v_fma_f32 %3, %3, %4, %5\n s_lshl_b32 %6, %6, 1" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd) : "v"(0.999f), "v"(1.0e-3f), "s"(seed) : "scc");) }
		// 4 VALU each DEPENDENT on the previous one (one chain): what a wave pays when the compiler gives it no independent work
		if (KIND == 12) { REP64(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2" : "+v"(fa) : "v"(0.999f), "v"(1.0e-3f));) }
		// 3 VALU + one v_readfirstlane / v_cmp + s_and: VALU results consumed by the scalar unit (what a wave vote does)
		if (KIND == 13) { REP64(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_cmp_gt_f32 vcc, %0, %1\n v_fma_f32 %2, %2, %4, %5\n v_cndmask_b32 %3, %3, %2, vcc" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd) : "v"(0.999f), "v"(1.0e-3f) : "vcc");) }
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
	if ((threadIdx.x & 63u) == 0u && stamps) {
		const unsigned w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
		stamps[2u * w] = t1 - t0;
		stamps[2u * w + 1u] = r1 - r0;
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ __float_as_uint(fa + fb + fc + fd) ^ (unsigned)(qa ^ qb);
}

#include <algorithm>
static unsigned long long *d_stamps = nullptr;
template <int KIND> static double run(const char *name, int waves_per_simd, unsigned *d_out, double)
{
	const int n_cu = 256, blocks = n_cu * waves_per_simd; // 256 threads = 4 waves = one per SIMD
	const unsigned n_iter = 20000;
	const int n_waves = blocks * 4;
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	hipLaunchKernelGGL(loop_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 2000u, 1u, (unsigned long long *)nullptr); // warm-up: clocks settle
	(void)hipEventRecord(e0);
	hipLaunchKernelGGL(loop_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, n_iter, 3u, d_stamps);
	(void)hipEventRecord(e1);
	const hipError_t se = hipEventSynchronize(e1);
	float ms = 0;
	(void)hipEventElapsedTime(&ms, e0, e1);
	if (se != hipSuccess || hipGetLastError() != hipSuccess)
		std::printf("%-34s waves/SIMD %d: HIP error %s\n", name, waves_per_simd, hipGetErrorString(se));
	std::vector<unsigned long long> st(2 * (size_t)n_waves);
	(void)hipMemcpy(st.data(), d_stamps, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
	std::vector<double> cyc(n_waves), ghz(n_waves);
	for (int w = 0; w < n_waves; ++w) {
		cyc[w] = (double)st[2 * w];
		ghz[w] = st[2 * w + 1] ? (double)st[2 * w] / (double)st[2 * w + 1] * 0.1 : 0.0; // ticks per 10 ns -> GHz
	}
	std::sort(cyc.begin(), cyc.end());
	std::sort(ghz.begin(), ghz.end());
	const double instr_per_wave = (double)n_iter * 64 * 4;
	const double per_simd = cyc[n_waves / 2] / (instr_per_wave * waves_per_simd); // measured cycles of SIMD time per wave-instruction
	const double wall_cycles_at_clock = ms * 1e-3 * ghz[n_waves / 2] * 1e9 / (instr_per_wave * waves_per_simd);
	std::printf("%-38s waves/SIMD %d: %8.3f ms  s_memtime: %6.2f cycles per wave-instruction per SIMD (p10 %.2f p90 %.2f)  clock held %.3f GHz "
	            "(p10 %.3f p90 %.3f)  [wall x that clock: %.2f; wall x 2.4 GHz, round 3's reading: %.2f]\n",
	            name, waves_per_simd, ms, per_simd, cyc[n_waves / 10] / (instr_per_wave * waves_per_simd), cyc[n_waves * 9 / 10] / (instr_per_wave * waves_per_simd),
	            ghz[n_waves / 2], ghz[n_waves / 10], ghz[n_waves * 9 / 10], wall_cycles_at_clock, ms * 1e-3 * 2.4e9 / (instr_per_wave * waves_per_simd));
	return per_simd;
}

int main(int argc, char **argv)
{
	setvbuf(stdout, nullptr, _IONBF, 0);
	const double clock_ghz = 2.4;
	unsigned *d_out = nullptr;
	if (hipMalloc(&d_out, 256 * 8 * 256 * sizeof(unsigned)) != hipSuccess) { std::printf("hipMalloc failed\n"); return 1; }
	if (hipMalloc(&d_stamps, 256 * 8 * 4 * 2 * sizeof(unsigned long long)) != hipSuccess) { std::printf("hipMalloc failed\n"); return 1; }
	std::vector<int> ws = {1, 2, 4, 8}; // or the waves per SIMD named on the command line (round 4: 5 6, for the six-wave config-2 kernel)
	if (argc > 1) {
		ws.clear();
		for (int i = 1; i < argc; ++i)
			ws.push_back(std::atoi(argv[i]));
	}
	for (int w : ws) {
		run<0>("v_fma_f32", w, d_out, clock_ghz);
		run<12>("v_fma_f32, ONE dependent chain", w, d_out, clock_ghz);
		run<5>("v_xor_b32", w, d_out, clock_ghz);
		run<10>("v_cndmask_b32 / v_mov_b32", w, d_out, clock_ghz);
		run<11>("v_fma_f32 + s_alu 1:1 (VALU only counted)", w, d_out, clock_ghz);
		run<13>("v_fma / v_cmp vcc / v_cndmask vcc", w, d_out, clock_ghz);
		run<1>("v_mul_lo_u32", w, d_out, clock_ghz);
		run<2>("v_mul_hi_u32", w, d_out, clock_ghz);
		run<6>("v_mul_u32_u24", w, d_out, clock_ghz);
		run<3>("v_mad_u64_u32", w, d_out, clock_ghz);
		run<7>("v_pk_fma_f32", w, d_out, clock_ghz);
		run<4>("v_rcp_f32", w, d_out, clock_ghz);
		run<9>("v_sqrt_f32", w, d_out, clock_ghz);
		run<8>("v_div_scale/fixup/fmas/alignbit mix", w, d_out, clock_ghz);
	}
	(void)hipFree(d_out);
	(void)hipFree(d_stamps);
	return 0;
}
