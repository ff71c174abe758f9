// The ushort streaming resample kernels: __global__ wrappers and launches of resample16_body.h;
// host side resample16_host.h (both shared with tests/emul).
#include "resample16_body.h"

namespace vh {

__global__ void __launch_bounds__(R16_NT)
reducev16(R16VArgs a)
{
	__shared__ unsigned int r16_slot[4];
	reducev16_block(a, r16_slot);
}

__global__ void __launch_bounds__(R16_NT)
reducev8(R16VArgs a)
{
	__shared__ unsigned int r16_slot[4];
	reducev8_block(a, r16_slot);
}

__global__ void __launch_bounds__(R16_NT)
shrinkv16(R16VArgs a)
{
	shrinkv16_body(a, (int) blockIdx.x, (int) blockIdx.y, (int) gridDim.y);
}

template <int B>
__global__ void __launch_bounds__(R16_NT)
reduceh16(R16HArgs a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned int r16_lds[];
	reduceh16_body<B>(a, (int) blockIdx.x, (int) blockIdx.y, (int) gridDim.y, r16_lds);
}

template <int B>
__global__ void __launch_bounds__(R16_NT)
shrinkh16(R16HArgs a)
{
	shrinkh16_body<B>(a, (int) blockIdx.x, (int) blockIdx.y, (int) gridDim.y);
}

template <int B>
__global__ void __launch_bounds__(R16_NT)
shrinkbox16(R16HArgs a)
{
	shrinkbox16_body<B>(a, (int) blockIdx.x, (int) blockIdx.y, (int) gridDim.y);
}

template <int B, int L>
__global__ void __launch_bounds__(R16_NT)
shrinkbox16c(R16HArgs a)
{
	shrinkbox16c_body<B, L>(a, (int) blockIdx.x, (int) blockIdx.y, (int) gridDim.y);
}

} // namespace vh

#include "resample16_host.h"

namespace vh {

static int r16_launch_v(int which, const R16VArgs &a, int gx, int gy)
{
	if (which == 0)
		hipLaunchKernelGGL(reducev16, dim3(gx), dim3(R16_NT), 0, stream(), a);
	else if (which == 2)
		hipLaunchKernelGGL(reducev8, dim3(gx), dim3(R16_NT), 0, stream(), a);
	else
		hipLaunchKernelGGL(shrinkv16, dim3(gx, gy, 1), dim3(R16_NT), 0, stream(), a);
	return hipGetLastError() != hipSuccess ? -1 : 0;
}

static int r16_launch_h(int which, int bands, const R16HArgs &a, int gx, int gy, size_t lds)
{
	const dim3 grid(gx, gy, 1), block(R16_NT, 1, 1);
#define R16_H(B) \
	if (bands == B) { \
		if (which == 0) \
			hipLaunchKernelGGL(reduceh16<B>, grid, block, lds, stream(), a); \
		else if (which == 2) \
			hipLaunchKernelGGL(shrinkbox16<B>, grid, block, 0, stream(), a); \
		else \
			hipLaunchKernelGGL(shrinkh16<B>, grid, block, 0, stream(), a); \
		return hipGetLastError() != hipSuccess ? -1 : 0; \
	}
	R16_H(1) R16_H(2) R16_H(3) R16_H(4)
#undef R16_H
	return -1;
}

static int r16_launch_boxc(int bands, int lanes_per_box, const R16HArgs &a, int gx, int gy)
{
	const dim3 grid(gx, gy, 1), block(R16_NT, 1, 1);
#define R16_C(B, L) \
	if (bands == B && lanes_per_box == L) { \
		hipLaunchKernelGGL((shrinkbox16c<B, L>), grid, block, 0, stream(), a); \
		return hipGetLastError() != hipSuccess ? -1 : 0; \
	}
	R16_C(1, 1) R16_C(1, 2) R16_C(1, 4) R16_C(2, 1) R16_C(2, 2) R16_C(2, 4) R16_C(4, 1) R16_C(4, 2) R16_C(4, 4)
#undef R16_C
	return -1;
}

} // namespace vh
