// Is the three-operation constant division (div_const, colour_device.h) the IEEE quotient
// everywhere the sharpen kernel uses it?  Exhaustive comparison on the device:
//   1. LabS -> XYZ: step_Lab2XYZ_c against step_Lab2XYZ (IEEE `/`) for all 2^15 x 2^16 (L, a)
//      pairs (X, Y) and, with b = a, (L, b) pairs (Z), including LabS2Lab's own division;
//   2. sRGB -> Lab: the bucket / fraction of cbrt_lerp for all 2^24 uchar RGB triples.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Iinclude -Ilibvips_amd/csrc tools/div_probe.hip -o tools/div_probe
#include "colour_device.h"

#include <cstdio>
#include <cmath>
#include <vector>

using namespace vh;

__global__ void probe_lab2xyz(unsigned long long *bad)
{
	const unsigned int idx = blockIdx.x * blockDim.x + threadIdx.x; // 2^31 threads
	const int L = (int) (idx >> 16);
	const int A = (int) (idx & 0xffff) - 32768;
	Px ref, fast;
	ref.a = (float) __ddiv_rn((double) L, 32767.0 / 100.0);
	ref.b = (float) __ddiv_rn((double) A, 32768.0 / 128.0);
	ref.c = ref.b;
	fast.a = (float) DIV_CONST_F((double) L, 32767.0 / 100.0);
	fast.b = __fmul_rn((float) A, 0.00390625f);
	fast.c = fast.b;
	const Px x = step_Lab2XYZ(ref), y = step_Lab2XYZ_c(fast);
	if (__float_as_uint(x.a) != __float_as_uint(y.a) || __float_as_uint(x.b) != __float_as_uint(y.b) ||
		__float_as_uint(x.c) != __float_as_uint(y.c) || __float_as_uint(ref.a) != __float_as_uint(fast.a) ||
		__float_as_uint(ref.b) != __float_as_uint(fast.b))
		atomicAdd(bad, 1ULL);
}

template <int WHICH>
static __device__ bool cbrt_bucket_same(float v)
{
	const double num = (double) __fmul_rn(100000.0f, v);
	const float fast = (float) (WHICH == 0 ? DIV_CONST_F(num, 95.0470) : WHICH == 1 ? DIV_CONST_F(num, 100.0)
																				: DIV_CONST_F(num, 108.8827));
	const float ref = (float) (WHICH == 0 ? __ddiv_rn(num, 95.0470) : WHICH == 1 ? __ddiv_rn(num, 100.0)
																			 : __ddiv_rn(num, 108.8827));
	return __float_as_uint(fast) == __float_as_uint(ref);
}

__global__ void probe_xyz2lab(const float *v2Y, unsigned long long *bad)
{
	const unsigned int idx = blockIdx.x * blockDim.x + threadIdx.x; // 2^24 threads
	Px v;
	v.a = v2Y[idx & 255];
	v.b = v2Y[(idx >> 8) & 255];
	v.c = v2Y[(idx >> 16) & 255];
	v = step_scRGB2XYZ(v);
	if (!cbrt_bucket_same<0>(v.a) || !cbrt_bucket_same<1>(v.b) || !cbrt_bucket_same<2>(v.c))
		atomicAdd(bad, 1ULL);
}

int main()
{
	unsigned long long *bad, host[2] = { 0, 0 };
	if (hipMalloc(&bad, 16) != hipSuccess || hipMemset(bad, 0, 16) != hipSuccess)
		return 1;
	// vips_col_make_tables_RGB_8, LabQ2sRGB.c:151-159
	std::vector<float> v2Y(256);
	for (int i = 0; i < 256; i++) {
		const float f = i / 255.0;
		v2Y[i] = f <= 0.04045 ? f / 12.92 : pow((f + 0.055) / (1 + 0.055), 2.4);
	}
	float *d_v2Y;
	if (hipMalloc(&d_v2Y, 1024) != hipSuccess || hipMemcpy(d_v2Y, v2Y.data(), 1024, hipMemcpyHostToDevice) != hipSuccess)
		return 1;
	hipLaunchKernelGGL(probe_lab2xyz, dim3(1u << 23), dim3(256), 0, 0, bad);
	hipLaunchKernelGGL(probe_xyz2lab, dim3(1u << 16), dim3(256), 0, 0, d_v2Y, bad + 1);
	if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(host, bad, 16, hipMemcpyDeviceToHost) != hipSuccess)
		return 1;
	printf("LabS -> XYZ, constant divisions against IEEE division: %llu of 2147483648 (L, a|b) pairs differ\n", host[0]);
	printf("sRGB -> Lab, cbrt table position: %llu of 16777216 RGB triples differ\n", host[1]);
	return host[0] || host[1] ? 2 : 0;
}
