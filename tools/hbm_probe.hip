// HBM streaming-read / copy probe for the roofline denominator (report
// against the 8 TB/s spec AND the bandwidth a plain streaming kernel reaches on this box).
// build: hipcc --offload-arch=gfx950 -O3 tools/hbm_probe.hip -o tools/hbm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <typename V>
__global__ void read_kernel(const V *__restrict__ in, size_t n, unsigned *sink)
{
	size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
	size_t stride = (size_t) gridDim.x * blockDim.x;
	unsigned acc = 0;
	for (; i + 3 * stride < n; i += 4 * stride) {
		V a = in[i], b = in[i + stride], c = in[i + 2 * stride], d = in[i + 3 * stride];
		const unsigned *pa = (const unsigned *) &a, *pb = (const unsigned *) &b,
					   *pc = (const unsigned *) &c, *pd = (const unsigned *) &d;
		for (int k = 0; k < (int) (sizeof(V) / 4); k++)
			acc ^= pa[k] ^ pb[k] ^ pc[k] ^ pd[k];
	}
	for (; i < n; i += stride) {
		V a = in[i];
		const unsigned *pa = (const unsigned *) &a;
		for (int k = 0; k < (int) (sizeof(V) / 4); k++)
			acc ^= pa[k];
	}
	if (acc == 0x12345678)
		*sink = acc;
}

// rows of `width_bytes`, each block reads a 2 KB wide column strip top to bottom, 8 rows per
// iteration: the access pattern of the fused reduce kernel
__global__ void strip_kernel(const uint2 *__restrict__ in, size_t row_u2, int rows_per_block,
	int strips, unsigned *sink)
{
	int strip = blockIdx.x % strips;
	int seg = blockIdx.x / strips;
	const uint2 *p = in + (size_t) seg * rows_per_block * row_u2 + (size_t) strip * 256 + threadIdx.x;
	unsigned acc = 0;
	for (int r = 0; r < rows_per_block; r += 8) {
		uint2 v[8];
#pragma unroll
		for (int i = 0; i < 8; i++)
			v[i] = p[(size_t) (r + i) * row_u2];
#pragma unroll
		for (int i = 0; i < 8; i++)
			acc ^= v[i].x ^ v[i].y;
	}
	if (acc == 0x12345678)
		*sink = acc;
}

template <typename V>
__global__ void copy_kernel(const V *__restrict__ in, V *__restrict__ out, size_t n)
{
	size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
	size_t stride = (size_t) gridDim.x * blockDim.x;
	for (; i < n; i += stride)
		out[i] = in[i];
}

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename F>
static double time_ms(F f, int reps)
{
	hipEvent_t a, b;
	CHECK(hipEventCreate(&a));
	CHECK(hipEventCreate(&b));
	f();
	CHECK(hipDeviceSynchronize());
	CHECK(hipEventRecord(a));
	for (int i = 0; i < reps; i++)
		f();
	CHECK(hipEventRecord(b));
	CHECK(hipEventSynchronize(b));
	float ms;
	CHECK(hipEventElapsedTime(&ms, a, b));
	return ms / reps;
}

int main()
{
	const size_t bytes = (size_t) 1 << 30;
	void *in, *out;
	unsigned *sink;
	CHECK(hipMalloc(&in, bytes));
	CHECK(hipMalloc(&out, bytes));
	CHECK(hipMalloc(&sink, 4));
	CHECK(hipMemset(in, 1, bytes));
	for (int blocks : { 1024, 2048, 4096, 8192 }) {
		double ms16 = time_ms([&] { read_kernel<uint4><<<blocks, 256>>>((const uint4 *) in, bytes / 16, sink); }, 20);
		double ms8 = time_ms([&] { read_kernel<uint2><<<blocks, 256>>>((const uint2 *) in, bytes / 8, sink); }, 20);
		printf("read  1 GiB grid %5d: 16B/lane %.4f ms %.0f GB/s | 8B/lane %.4f ms %.0f GB/s\n", blocks,
			ms16, bytes / ms16 / 1e6, ms8, bytes / ms8 / 1e6);
	}
	for (int rows : { 512, 1024, 2048 }) {
		int strips = 32; // 65536 B row / 2048 B
		int segs = 16384 / rows;
		double ms = time_ms([&] { strip_kernel<<<strips * segs, 256>>>((const uint2 *) in, 65536 / 8, rows, strips, sink); }, 20);
		printf("strip 1 GiB (2KB x %4d rows per block, %d blocks): %.4f ms %.0f GB/s\n", rows,
			strips * segs, ms, bytes / ms / 1e6);
	}
	double msc = time_ms([&] { copy_kernel<uint4><<<4096, 256>>>((const uint4 *) in, (uint4 *) out, bytes / 16); }, 20);
	printf("copy  1 GiB (read+write 2 GiB): %.4f ms %.0f GB/s\n", msc, 2.0 * bytes / msc / 1e6);
	double msm = time_ms([&] { CHECK(hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, 0)); }, 20);
	printf("hipMemcpy D2D 1 GiB (2 GiB traffic): %.4f ms %.0f GB/s\n", msm, 2.0 * bytes / msm / 1e6);
	return 0;
}
