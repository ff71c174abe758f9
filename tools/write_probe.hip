// Does a 1.5 % write stream slow a 1 GiB streaming read?  (profiling aid for DESIGN.md 3.1)
//   mode 0: reads only   1: trickle writes (1 dword per 64 B read, as the fused reduce does)
//   mode 2: same bytes, written in one burst at the end of each block   3: writes only
// build: hipcc --offload-arch=gfx950 -O3 tools/write_probe.hip -o tools/write_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) probe(const uint2 *__restrict__ in, unsigned int *__restrict__ out,
	size_t chunk_u2 /* uint2 per block */)
{
	const uint2 *p = in + (size_t) blockIdx.x * chunk_u2 + threadIdx.x;
	unsigned int *q = out + (size_t) blockIdx.x * (chunk_u2 / 32) + threadIdx.x;
	const int iters = (int) (chunk_u2 / 256);
	unsigned int acc = 0;
	unsigned int keep[64];
	int nkeep = 0;
	for (int i = 0; i < iters; i += 8) {
		uint2 v[8];
		if (MODE != 3) {
#pragma unroll
			for (int k = 0; k < 8; k++)
				v[k] = p[(size_t) (i + k) * 256];
#pragma unroll
			for (int k = 0; k < 8; k++)
				acc ^= v[k].x + v[k].y;
		}
		else
			acc += i;
		if ((i & 31) == 24) {
			if (MODE == 1 || MODE == 3)
				q[(size_t) (i / 32) * 256] = acc;
			else if (MODE == 2) {
#pragma unroll
				for (int k = 0; k < 64; k++)
					if (k == nkeep)
						keep[k] = acc;
				nkeep++;
			}
		}
	}
	if (MODE == 2) {
#pragma unroll
		for (int k = 0; k < 64; k++)
			if (k < nkeep)
				q[(size_t) k * 256] = keep[k];
	}
	if (MODE == 0 && acc == 0x12345678u)
		q[0] = acc;
}

template <int MODE>
static float run(const uint2 *in, unsigned int *out, int blocks, size_t chunk_u2)
{
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	float best = 1e9f;
	for (int rep = 0; rep < 12; rep++) {
		CK(hipEventRecord(e0));
		hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, chunk_u2);
		CK(hipEventRecord(e1));
		CK(hipEventSynchronize(e1));
		float ms;
		CK(hipEventElapsedTime(&ms, e0, e1));
		if (rep >= 2 && ms < best)
			best = ms;
	}
	return best;
}

int main()
{
	const size_t bytes = 1ull << 30;
	uint2 *in;
	unsigned int *out;
	CK(hipMalloc(&in, bytes));
	CK(hipMalloc(&out, bytes / 16));
	CK(hipMemset(in, 1, bytes));
	for (int blocks : { 1024, 2048 }) {
		const size_t chunk_u2 = bytes / 8 / blocks;
		printf("blocks %d: reads only %.4f | trickle writes %.4f | burst writes %.4f | writes only %.4f ms\n", blocks,
			run<0>(in, out, blocks, chunk_u2), run<1>(in, out, blocks, chunk_u2), run<2>(in, out, blocks, chunk_u2),
			run<3>(in, out, blocks, chunk_u2));
	}
	return 0;
}
