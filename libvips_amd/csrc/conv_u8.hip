// Integer convolution of uchar images on packed bytes: the __global__ wrappers and launches of
// conv_u8_body.h (see there); host side conv_u8_host.h (both shared with tests/emul).
#include "conv_u8_body.h"

namespace vh {

// H: the mask's half-width when it is at most 4 (coefficient dwords without a tap are compiled out), else -1
template <int B, int ND, int H, bool RG>
__global__ void __launch_bounds__(256)
conv_u8_sep(Cu8Args a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned int cu8_lds[];
	conv_u8_sep_block<B, ND, H, RG>(a, cu8_lds);
}

template <int B, int MH, int H>
__global__ void __launch_bounds__(256)
conv_u8_2d(Cu8Args a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned int cu8_lds[];
	conv_u8_2d_block<B, MH, H>(a, cu8_lds);
}

} // namespace vh

#include "conv_u8_host.h"

namespace vh {

template <typename K>
static int cu8_go(K kernel, const Cu8Args &a, int grid, size_t lds)
{
	if (lds > 64 * 1024)
		VH_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
	hipLaunchKernelGGL(kernel, dim3(grid), dim3(CU8_NT), lds, stream(), a);
	VH_CHECK(hipGetLastError());
	return 0;
}

#define CU8_SEP(B, ND, H) \
	if (bands == B && nd == ND && h == H) \
		return regs || ND == 3 ? cu8_go(conv_u8_sep<B, ND, H, true>, a, grid, lds) : cu8_go(conv_u8_sep<B, ND, H, false>, a, grid, lds);
#define CU8_SEP_B(B) \
	CU8_SEP(B, 3, 1) CU8_SEP(B, 3, 2) CU8_SEP(B, 3, -1) CU8_SEP(B, 5, -1) CU8_SEP(B, 7, -1) CU8_SEP(B, 9, -1)
#define CU8_2D(B, MH, H) \
	if (bands == B && mh == MH && h == H) \
		return cu8_go(conv_u8_2d<B, MH, H>, a, grid, lds);
#define CU8_2D_B(B) \
	CU8_2D(B, 3, 1) CU8_2D(B, 3, 2) CU8_2D(B, 3, -1) CU8_2D(B, 5, 1) CU8_2D(B, 5, 2) CU8_2D(B, 5, -1) \
	CU8_2D(B, 7, 1) CU8_2D(B, 7, 2) CU8_2D(B, 7, -1)

static int cu8_launch_sep(int bands, int nd, bool regs, const Cu8Args &a, int grid, size_t lds)
{
	const int h = nd == 3 && a.half <= 2 ? a.half : -1;
	CU8_SEP_B(1) CU8_SEP_B(3) CU8_SEP_B(4)
	return 1;
}

static int cu8_launch_2d(int bands, int mh, const Cu8Args &a, int grid, size_t lds)
{
	const int h = a.half < 1 ? 1 : a.half > 2 ? -1 : a.half; // (a 1-wide mask: as 3 wide with zero taps)
	CU8_2D_B(1) CU8_2D_B(3) CU8_2D_B(4)
	return 1;
}

} // namespace vh
