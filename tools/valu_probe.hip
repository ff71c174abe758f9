// VALU issue-rate probe: cycles per wave64 instruction for the integer MAC candidates.
// build: hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o tools/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short short2v __attribute__((ext_vector_type(2)));
typedef int int4v __attribute__((ext_vector_type(4)));

template <int OP>
__global__ void k(int *out, int n, unsigned seed)
{
	int a[16];
	unsigned x = threadIdx.x * 2654435761u + seed, y = x * 40503u + 7;
	for (int i = 0; i < 16; i++)
		a[i] = i + threadIdx.x;
	for (int it = 0; it < n; it++) {
#pragma unroll
		for (int i = 0; i < 16; i++) {
			if (OP == 0)
				a[i] = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, x), __builtin_bit_cast(short2v, y), a[i], false);
			else if (OP == 1)
				a[i] = __builtin_amdgcn_sdot4((int) x, (int) y, a[i], false);
			else if (OP == 2)
				a[i] = __builtin_amdgcn_udot4(x, y, a[i], false);
			else if (OP == 3)
				a[i] = (int) __umul24(x, y) + a[i];
			else if (OP == 4)
				a[i] = a[i] * (int) x + (int) y;
			else if (OP == 5)
				a[i] = __builtin_amdgcn_perm(x, a[i], y);
			else if (OP == 6)
				a[i] = a[i] + (int) x;
		}
		x += 3;
	}
	int s = 0;
	for (int i = 0; i < 16; i++)
		s ^= a[i];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// f64 ops: 8 independent chains per lane
template <int OP>
__global__ void kd(double *out, int n, double seed)
{
	double a[8];
	double x = threadIdx.x * 1.25 + seed, y = 0.999 + seed * 1e-9;
	float xf = (float) x;
	for (int i = 0; i < 8; i++)
		a[i] = i + threadIdx.x;
	for (int it = 0; it < n; it++) {
#pragma unroll
		for (int i = 0; i < 8; i++) {
			if (OP == 0)
				a[i] = __fma_rn(a[i], y, x);
			else if (OP == 1)
				a[i] = __dadd_rn(a[i], x);
			else if (OP == 2)
				a[i] = __dmul_rn(a[i], y);
			else if (OP == 3)
				a[i] += (double) (xf + (float) i); // cvt_f64_f32 + add_f32 + add_f64
		}
		x += 3.0;
		xf += 1.0f;
	}
	double s = 0;
	for (int i = 0; i < 8; i++)
		s += a[i];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void rund(const char *name)
{
	double *out;
	hipMalloc(&out, 256 * 1024 * 8 * sizeof(double));
	const int n = 4096, blocks = 256 * 8;
	hipEvent_t a, b;
	hipEventCreate(&a);
	hipEventCreate(&b);
	kd<OP><<<blocks, 256>>>(out, n, 1);
	hipDeviceSynchronize();
	hipEventRecord(a);
	kd<OP><<<blocks, 256>>>(out, n, 2);
	hipEventRecord(b);
	hipEventSynchronize(b);
	float ms;
	hipEventElapsedTime(&ms, a, b);
	double wi = (double) blocks * 4 / 1024 * n * 8;
	printf("%-14s %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, ms,
		ms * 1e6 / wi, ms * 1e6 / wi * 2.4);
	hipFree(out);
}

template <int OP>
void run(const char *name)
{
	int *out;
	hipMalloc(&out, 256 * 1024 * 8 * sizeof(int));
	const int n = 4096, blocks = 256 * 8;
	hipEvent_t a, b;
	hipEventCreate(&a);
	hipEventCreate(&b);
	k<OP><<<blocks, 256>>>(out, n, 1);
	hipDeviceSynchronize();
	hipEventRecord(a);
	k<OP><<<blocks, 256>>>(out, n, 2);
	hipEventRecord(b);
	hipEventSynchronize(b);
	float ms;
	hipEventElapsedTime(&ms, a, b);
	// wave-instructions per SIMD: blocks*4 waves / 1024 SIMDs * n * 16
	double wi = (double) blocks * 4 / 1024 * n * 16;
	printf("%-14s %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, ms,
		ms * 1e6 / wi, ms * 1e6 / wi * 2.4);
	hipFree(out);
}

int main()
{
	run<0>("sdot2 i16");
	run<1>("sdot4 i8");
	run<2>("udot4 u8");
	run<3>("mad_u32_u24");
	run<4>("mad i32 full");
	run<5>("perm_b32");
	run<6>("add_u32");
	rund<0>("fma_f64");
	rund<1>("add_f64");
	rund<2>("mul_f64");
	rund<3>("cvt+addf32+add_f64");
	return 0;
}
