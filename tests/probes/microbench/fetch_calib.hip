// Calibration of the fabric-read counters (FETCH_SIZE, TCC_EA0_RDREQ*, TCC_MISS) for the access pattern of a
// BVH walk: dependent random 64-byte record fetches from a set far larger than L2 (and optionally than the
// Infinity Cache), with a KNOWN number of fetches.  MI355X_MICROARCH.md calibrates FETCH_SIZE for wide
// streaming reads only ("calibrate on a known byte count in your own access pattern").
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -- ./fetch_calib <set MiB> <record bytes: 64|128> <stream: 0|1>
// stream = 1: a coalesced float4 streaming read of the whole set instead (the guide's own case, as a control).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int R>
__global__ __launch_bounds__(256) void chase(const float4 *__restrict__ buf, uint32_t n_rec, int iters, float *out)
{
	uint32_t idx = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u % n_rec;
	float acc = 0.0f;
	for (int i = 0; i < iters; ++i) {
		const float4 *q = buf + (size_t)idx * R;
		float s = 0.0f;
#pragma unroll
		for (int k = 0; k < R; ++k) {
			const float4 v = q[k];
			s += v.x + v.y + v.z + v.w;
		}
		acc += s;
		idx = (idx * 1664525u + 1013904223u + __float_as_uint(s)) % n_rec;
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void stream_read(const float4 *__restrict__ buf, size_t n, float *out)
{
	float acc = 0.0f;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		const float4 v = buf[i];
		acc += v.x + v.y + v.z + v.w;
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main(int argc, char **argv)
{
	const size_t mib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1024;
	const int rec = argc > 2 ? atoi(argv[2]) : 64;
	const int stream = argc > 3 ? atoi(argv[3]) : 0;
	const size_t bytes = mib << 20;
	float4 *buf;
	float *out;
	if (hipMalloc(&buf, bytes) != hipSuccess) return 1;
	hipMemset(buf, 0, bytes);
	const int blocks = 2048, iters = 512;
	hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
	hipDeviceSynchronize();
	hipEvent_t a, b;
	hipEventCreate(&a); hipEventCreate(&b);
	hipEventRecord(a);
	if (stream)
		stream_read<<<blocks, 256>>>(buf, bytes / 16, out);
	else if (rec == 128)
		chase<8><<<blocks, 256>>>(buf, (uint32_t)(bytes / 128), iters, out);
	else
		chase<4><<<blocks, 256>>>(buf, (uint32_t)(bytes / 64), iters, out);
	hipEventRecord(b);
	hipEventSynchronize(b);
	float ms = 0;
	hipEventElapsedTime(&ms, a, b);
	const double fetches = (double)blocks * 256 * iters;
	if (stream)
		printf("stream: %zu MiB read once = %.0f bytes, %.3f ms, %.2f TB/s\n", mib, (double)bytes, ms, bytes / ms / 1e9);
	else
		printf("chase: set %zu MiB, %d-byte records, %.0f lane fetches = %.0f requested bytes, %.3f ms, %.2f G fetches/s\n", mib, rec, fetches,
		       fetches * rec, ms, fetches / ms / 1e6);
	return 0;
}
