// valu_issue.hip -- what one VALU instruction costs on gfx950 when every SIMD holds W waves issuing nothing else: cycles per
// wave-instruction per SIMD for the instruction kinds the render kernels are made of (is a 32-bit integer multiply a
// full-rate instruction?  is v_mad_u64_u32 one multiply or two?).  hipcc --offload-arch=gfx950 -O3; run: valu_issue [waves_per_simd]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND> __global__ __launch_bounds__(256) void loop_kernel(unsigned *out, unsigned n_iter, unsigned seed)
{
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
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ __float_as_uint(fa + fb + fc + fd) ^ (unsigned)(qa ^ qb);
}

template <int KIND> static double run(const char *name, int waves_per_simd, unsigned *d_out, double clock_ghz)
{
	const int n_cu = 256, blocks = n_cu * waves_per_simd; // 256 threads = 4 waves = one per SIMD
	const unsigned n_iter = 2000;
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	hipLaunchKernelGGL(loop_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 10u, 1u);
	(void)hipEventRecord(e0);
	hipLaunchKernelGGL(loop_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, n_iter, 3u);
	(void)hipEventRecord(e1);
	const hipError_t se = hipEventSynchronize(e1);
	float ms = 0;
	(void)hipEventElapsedTime(&ms, e0, e1);
	if (se != hipSuccess || hipGetLastError() != hipSuccess)
		std::printf("%-34s waves/SIMD %d: HIP error %s\n", name, waves_per_simd, hipGetErrorString(se));
	const double instr_per_wave = (double)n_iter * 64 * 4;
	const double cycles = ms * 1e-3 * clock_ghz * 1e9;
	const double per_simd = cycles / (instr_per_wave * waves_per_simd); // cycles of SIMD time per wave-instruction
	std::printf("%-34s waves/SIMD %d: %8.3f ms  %6.2f cycles per wave-instruction per SIMD (at %.2f GHz)\n", name, waves_per_simd, ms, per_simd, clock_ghz);
	return per_simd;
}

int main(int argc, char **argv)
{
	setvbuf(stdout, nullptr, _IONBF, 0);
	const double clock_ghz = 2.4;
	unsigned *d_out = nullptr;
	if (hipMalloc(&d_out, 256 * 8 * 256 * sizeof(unsigned)) != hipSuccess) { std::printf("hipMalloc failed\n"); return 1; }
	for (int w : {1, 2, 4}) {
		run<0>("v_fma_f32", w, d_out, clock_ghz);
		run<5>("v_xor_b32", w, d_out, clock_ghz);
		run<10>("v_cndmask_b32 / v_mov_b32", w, d_out, clock_ghz);
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
	return 0;
}
