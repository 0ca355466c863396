// Microbenchmark: dependent random record fetches, the memory pattern of a BVH walk on a big tree.
// Every lane chases its own chain through `n_rec` records of `stride` bytes and reads `R` dwordx4 of each.
//   hipcc --offload-arch=gfx950 -O3 -o random_fetch random_fetch.hip && ./random_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int R>
__global__ __launch_bounds__(256) void chase(const float4 *__restrict__ buf, uint32_t n_rec, uint32_t stride16, int iters, float *out)
{
	uint32_t idx = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u % n_rec;
	float acc = 0.0f;
	for (int i = 0; i < iters; ++i) {
		const float4 *q = buf + (size_t)idx * stride16;
		float4 v[R];
#pragma unroll
		for (int k = 0; k < R; ++k)
			v[k] = q[k];
		float s = 0.0f;
#pragma unroll
		for (int k = 0; k < R; ++k)
			s += v[k].x + v[k].y + v[k].z + v[k].w;
		acc += s;
		idx = (idx * 1664525u + 1013904223u + __float_as_uint(s)) % n_rec;
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int R> static void run(const float4 *buf, uint32_t n_rec, uint32_t stride, int blocks, float *out, const char *what)
{
	const int iters = 2000;
	hipEvent_t a, b;
	hipEventCreate(&a); hipEventCreate(&b);
	chase<R><<<blocks, 256>>>(buf, n_rec, stride / 16, 100, out);
	hipEventRecord(a);
	chase<R><<<blocks, 256>>>(buf, n_rec, stride / 16, iters, out);
	hipEventRecord(b);
	hipEventSynchronize(b);
	float ms = 0;
	hipEventElapsedTime(&ms, a, b);
	const double fetches = (double)blocks * 256 * iters;
	printf("%-34s R=%d (%3d B of %3d B records, %4.0f MB): %7.2f ms  %7.2f G lane-fetches/s  %7.2f TB/s requested\n", what, R, R * 16, stride,
	       (double)n_rec * stride / 1e6, ms, fetches / ms / 1e6, fetches * R * 16 / ms / 1e9);
}

int main(int argc, char **argv)
{
	// Round 4: the access shape the big-tree walk really has since the compact wide node -- THREE 16-byte pieces of a 64-byte
	// record (48 of 64 B) per node step, TWO of a 32-byte record per leaf box -- at the kernels' 4 waves per SIMD, over sets that are
	// L2-resident per XCD (2 MB), split between L2 and the Infinity Cache (8, 32, 128 MB: the 1 M-triangle tree is 30 MB of nodes +
	// 32 MB of leaf boxes + 48 MB of primitives) and beyond the Infinity Cache (1 GB: the 10 M-triangle tree is 0.3 + 0.3 + 0.5 GB).
	// The round-1 figures (64 of 64 B at 8 MB and 134 MB) are kept for comparison.
	const size_t bytes = (size_t)1 << 30;
	float4 *buf;
	float *out;
	if (hipMalloc(&buf, bytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
	hipMemset(buf, 0, bytes);
	hipMalloc(&out, 4096 * 256 * sizeof(float));
	for (int waves_per_simd : {4, 8}) {
		const int blocks = 256 * waves_per_simd; // 256 CUs x (4 waves per block = 1 per SIMD)
		char what[64];
		for (size_t mb : {2, 8, 32, 128, 1024}) {
			snprintf(what, sizeof what, "%d waves/SIMD", waves_per_simd);
			run<3>(buf, (uint32_t)((mb << 20) / 64), 64, blocks, out, what);  // the compact wide node
			run<2>(buf, (uint32_t)((mb << 20) / 32), 32, blocks, out, what);  // a leaf box
			run<4>(buf, (uint32_t)((mb << 20) / 64), 64, blocks, out, what);  // (round 1's shape)
			run<3>(buf, (uint32_t)((mb << 20) / 48), 48, blocks, out, what);  // a primitive record: 48 of 48 B, straddling lines
		}
	}
	return 0;
}
